// Instantiations of the fused MU kernel for padded rank 64 (one translation unit per rank so they build in parallel).
#include "nmfmu_fused.h"

namespace nmfmu {
int launch_fused_r64(int beta_kind, int x3, int mode, int stage, int g, const FusedArgs& a, int grid, hipStream_t s) {
  return launch_fused_dispatch<64, true>(beta_kind, x3, mode, stage, g, a, grid, s);
}
}  // namespace nmfmu
