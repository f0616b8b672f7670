// Instantiation of the software-pipelined one-wave-per-SIMD fused MU kernel (padded rank 256, beta = 1, fp16), nmfmu_sp.h.
#include "nmfmu_sp.h"

namespace nmfmu {

bool sp_available(int r_pad, int opt, int mode) { return r_pad == 256 && opt == kOpF16 && mode == kModeMU; }

int launch_sp(int r_pad, int opt, const FusedArgs& a, int grid, hipStream_t s) {
  if (r_pad == 256 && opt == kOpF16) return launch_sp_one<256, kOpF16>(a, grid, s);
  return -2;
}

int launch_sp2_a(int beta_kind, const FusedArgs& a, int grid, hipStream_t s);   // nmfmu_inst_sp2a.hip: kIS, kSqrt
int launch_sp2_b(int beta_kind, const FusedArgs& a, int grid, hipStream_t s);   // nmfmu_inst_sp2b.hip: kSqrt3, kGen

int launch_sp2(int r_pad, int beta_kind, const FusedArgs& a, int grid, hipStream_t s) {
  if (r_pad != 128) return -2;
  if (beta_kind == kIS || beta_kind == kSqrt) return launch_sp2_a(beta_kind, a, grid, s);
  if (beta_kind == kSqrt3 || beta_kind == kGen) return launch_sp2_b(beta_kind, a, grid, s);
  return -2;
}

}  // namespace nmfmu
