// Instantiations of the ping-pong (eight-wave, two-halves-out-of-phase) fused MU kernel, nmfmu_pp.h.
#include "nmfmu_pp.h"

namespace nmfmu {

bool pp_available(int r_pad, int opt, int mode) {
  return (r_pad == 32 || r_pad == 64 || r_pad == 128) && (opt == kOpBf16 || opt == kOpF16) &&
         (mode == kModeMU || mode == kModeLoss);
}

template <int R_PAD>
static int launch_pp_r(int opt, int mode, int var, const FusedArgs& a, int grid, hipStream_t s) {
#define NMFMU_PP_CASE(O, M, V) \
  if (opt == O && mode == M && var == V) return launch_pp_one<R_PAD, O, M, V>(a, grid, s);
  NMFMU_PP_CASE(kOpBf16, kModeMU, 0)
  NMFMU_PP_CASE(kOpF16, kModeMU, 0)
  NMFMU_PP_CASE(kOpBf16, kModeLoss, 0)
  NMFMU_PP_CASE(kOpF16, kModeLoss, 0)
  if constexpr (R_PAD == 128) {   // experiment variants of the headline instance only (NMFMU_PP_VAR, see nmfmu_pp.h)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 1)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 4)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 512)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 128)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 16384)
    NMFMU_PP_CASE(kOpF16, kModeMU, 16384)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 16512)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 152)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 160)
    NMFMU_PP_CASE(kOpBf16, kModeMU, 184)
    NMFMU_PP_CASE(kOpF16, kModeMU, 128)
  }
#undef NMFMU_PP_CASE
  if (var != 0) return launch_pp_r<R_PAD>(opt, mode, 0, a, grid, s);   // variant not built for this shape
  return -2;
}

int launch_pp(int r_pad, int opt, int mode, int var, const FusedArgs& a, int grid, hipStream_t s) {
  switch (r_pad) {
    case 32: return launch_pp_r<32>(opt, mode, var, a, grid, s);
    case 64: return launch_pp_r<64>(opt, mode, var, a, grid, s);
    case 128: return launch_pp_r<128>(opt, mode, var, a, grid, s);
  }
  return -2;
}

}  // namespace nmfmu
