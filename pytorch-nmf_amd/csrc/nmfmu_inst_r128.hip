// Instantiations of the four-wave fused MU kernel for padded rank 128 (one translation unit per rank so they build in parallel).
#include "nmfmu_fused.h"

namespace nmfmu {
int launch_fused_r128(int beta_kind, int prec, int mode, const FusedArgs& a, int grid, hipStream_t s) {
  return launch_fused_dispatch<128, true>(beta_kind, prec, mode, a, grid, s);
}
}  // namespace nmfmu
