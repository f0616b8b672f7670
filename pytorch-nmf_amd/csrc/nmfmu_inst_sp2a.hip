// Instantiations of the software-pipelined two-accumulator kernel (nmfmu_sp2.h): beta = 0 and beta = 0.5 (one unit per pair of
// betas: the hand-placed streams are large and build in parallel).
#include "nmfmu_sp2.h"

namespace nmfmu {
int launch_sp2_a(int beta_kind, const FusedArgs& a, int grid, hipStream_t s) {
  if (beta_kind == kIS) return launch_sp2_one<128, kOpF16, kIS>(a, grid, s);
  if (beta_kind == kSqrt) return launch_sp2_one<128, kOpF16, kSqrt>(a, grid, s);
  return -2;
}
}  // namespace nmfmu
