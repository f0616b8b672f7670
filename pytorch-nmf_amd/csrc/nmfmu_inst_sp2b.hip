// Instantiations of the software-pipelined two-accumulator kernel (nmfmu_sp2.h): beta = 1.5 and the generic beta.
#include "nmfmu_sp2.h"

namespace nmfmu {
int launch_sp2_b(int beta_kind, const FusedArgs& a, int grid, hipStream_t s) {
  if (beta_kind == kSqrt3) return launch_sp2_one<128, kOpF16, kSqrt3>(a, grid, s);
  if (beta_kind == kGen) return launch_sp2_one<128, kOpF16, kGen>(a, grid, s);
  return -2;
}
}  // namespace nmfmu
