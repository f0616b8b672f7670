// NMFD GEMM instances with the implicit Toeplitz operand staged as a window of table entries (nmfmu_gemm.h: WS).  The same
// operand combinations as launch_gemm's one-shift-axis branches; a translation unit of its own keeps the build parallel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nmfmu.h"
#include "nmfmu_gemm.h"

namespace nmfmu {

int launch_gemm_ws(int x3, int epi, int beta_kind, int ops, int f16, const GemmArgs& a, hipStream_t s) {
  if (f16) {
    if (x3) return -2;
#define GF16(E, B, O) \
  if (epi == E && (E == kEpiF32 || beta_kind == B) && ops == O) \
    return launch_gemm_one<false, E, B, O, GemmSmall, kOpF16, false, true>(a, s);
    GF16(kEpiRatio, kKL, kOpsBHu) GF16(kEpiRatio, kKL, kOpsAHu) GF16(kEpiLoss, kKL, kOpsBHu) GF16(kEpiF32, kEuc, kOpsBHuT)
#undef GF16
    return -2;
  }
#define G1(X, E, B, O) \
  if (x3 == (X ? 1 : 0) && epi == E && beta_kind == B && ops == O) \
    return launch_gemm_one<X, E, B, O, GemmSmall, kOpBf16, false, true>(a, s);
#define GB(X, E, O) G1(X, E, kKL, O) G1(X, E, kEuc, O) G1(X, E, kIS, O) G1(X, E, kGen, O)
  GB(false, kEpiRatio, kOpsBHu) GB(true, kEpiRatio, kOpsBHu) GB(false, kEpiRatio, kOpsAHu) GB(true, kEpiRatio, kOpsAHu)
  GB(false, kEpiLoss, kOpsBHu) GB(true, kEpiLoss, kOpsBHu)
  if (epi == kEpiF32 && ops == kOpsBHuT)
    return x3 ? launch_gemm_one<true, kEpiF32, kEuc, kOpsBHuT, GemmSmall, kOpBf16, false, true>(a, s)
              : launch_gemm_one<false, kEpiF32, kEuc, kOpsBHuT, GemmSmall, kOpBf16, false, true>(a, s);
#undef GB
#undef G1
  return -2;
}

}  // namespace nmfmu
