// Instantiation of the software-pipelined one-wave-per-SIMD fused MU kernel (padded rank 256, beta = 1, fp16), nmfmu_sp.h.
#include "nmfmu_sp.h"

namespace nmfmu {

bool sp_available(int r_pad, int opt, int mode) { return r_pad == 256 && opt == kOpF16 && mode == kModeMU; }

int launch_sp(int r_pad, int opt, const FusedArgs& a, int grid, hipStream_t s) {
  if (r_pad == 256 && opt == kOpF16) return launch_sp_one<256, kOpF16>(a, grid, s);
  return -2;
}

}  // namespace nmfmu
