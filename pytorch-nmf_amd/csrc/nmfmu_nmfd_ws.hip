// NMFD GEMM instances of round 5 (nmfmu_gemm.h): the implicit Toeplitz operand staged as a window of table entries (WS), and
// the eight-wave tile whose two wave groups split the contraction of every k-tile (GemmSmallK2).  The same operand
// combinations as launch_gemm's one-shift-axis branches; a translation unit of its own keeps the build parallel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nmfmu.h"
#include "nmfmu_gemm.h"

namespace nmfmu {

template <class SH, bool WS, int NST>
static int launch_r5(int x3, int epi, int beta_kind, int ops, int f16, const GemmArgs& a, hipStream_t s) {
  // (split bf16 on the eight-wave tile: 210 registers, one workgroup per CU -- the same two waves per SIMD as two four-wave
  // workgroups -- and three of its stages would not leave room for two workgroups: neither is instantiated)
  if constexpr (SH::KW > 1 || NST > 2) {
    if (x3) return -2;
  }
  constexpr bool kNoX3 = SH::KW > 1 || NST > 2;
  if (f16) {
    if (x3) return -2;
#define GF16(E, B, O) \
  if (epi == E && (E == kEpiF32 || beta_kind == B) && ops == O) return launch_gemm_one<false, E, B, O, SH, kOpF16, false, WS, NST>(a, s);
    GF16(kEpiRatio, kKL, kOpsBHu) GF16(kEpiRatio, kKL, kOpsAHu) GF16(kEpiLoss, kKL, kOpsBHu) GF16(kEpiF32, kEuc, kOpsBHuT)
#undef GF16
    return -2;
  }
#define G1(X, E, B, O) \
  if constexpr (!(X && kNoX3)) \
    if (x3 == (X ? 1 : 0) && epi == E && beta_kind == B && ops == O) return launch_gemm_one<X, E, B, O, SH, kOpBf16, false, WS, NST>(a, s);
#define GB(X, E, O) G1(X, E, kKL, O) G1(X, E, kEuc, O) G1(X, E, kIS, O) G1(X, E, kGen, O)
  GB(false, kEpiRatio, kOpsBHu) GB(true, kEpiRatio, kOpsBHu) GB(false, kEpiRatio, kOpsAHu) GB(true, kEpiRatio, kOpsAHu)
  GB(false, kEpiLoss, kOpsBHu) GB(true, kEpiLoss, kOpsBHu)
  if (epi == kEpiF32 && ops == kOpsBHuT) {
    if constexpr (!kNoX3) {
      if (x3) return launch_gemm_one<true, kEpiF32, kEuc, kOpsBHuT, SH, kOpBf16, false, WS, NST>(a, s);
    }
    return launch_gemm_one<false, kEpiF32, kEuc, kOpsBHuT, SH, kOpBf16, false, WS, NST>(a, s);
  }
#undef GB
#undef G1
  return -2;
}

// ws: window staging (the caller has checked gemm_window_stageable); eight: the eight-wave tile; nst: staging buffers (2 | 3).
// -2: the combination is not instantiated (the caller falls back)
int launch_gemm_r5(int x3, int epi, int beta_kind, int ops, int f16, const GemmArgs& a, hipStream_t s, bool ws, bool eight, int nst) {
  if (a.koff || (ops != kOpsBHu && ops != kOpsBHuT && ops != kOpsAHu)) return -2;
#define R5(SH, W, N) return launch_r5<SH, W, N>(x3, epi, beta_kind, ops, f16, a, s)
  if (ws && !eight && nst == 2) R5(GemmSmall, true, 2);
  if (ws && eight && nst == 2) R5(GemmSmallK2, true, 2);
  if (ws && !eight && nst == 3) R5(GemmSmall, true, 3);
  if (ws && eight && nst == 3) R5(GemmSmallK2, true, 3);
  if (!ws && eight && nst == 2) R5(GemmSmallK2, false, 2);
  if (!ws && eight && nst == 3) R5(GemmSmallK2, false, 3);
#undef R5
  return -2;
}

}  // namespace nmfmu
