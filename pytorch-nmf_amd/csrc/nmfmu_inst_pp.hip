// Instantiations of the ping-pong (eight-wave, two-halves-out-of-phase) fused MU kernel, nmfmu_pp.h.
#include "nmfmu_pp.h"

namespace nmfmu {

bool pp_available(int r_pad, int opt, int mode) {
  return (r_pad == 32 || r_pad == 64 || r_pad == 128) && (opt == kOpBf16 || opt == kOpF16) &&
         (mode == kModeMU || mode == kModeLoss);
}

template <int R_PAD>
static int launch_pp_r(int opt, int mode, const FusedArgs& a, int grid, hipStream_t s, bool xr, bool lacc) {
  if (lacc) {   // riding loss (nmfmu_mu_step_with_loss): fp16 operands, MU half-step, either target width
    if (opt != kOpF16 || mode != kModeMU || !a.loss_part) return -2;
    return xr ? launch_pp_one<R_PAD, kOpF16, kModeMU, true, true>(a, grid, s) : launch_pp_one<R_PAD, kOpF16, kModeMU, false, true>(a, grid, s);
  }
  if (xr) {   // the 3-byte target (NMFMU_PREC_F16R): fp16 operands only
    if (opt == kOpF16 && mode == kModeMU) return launch_pp_one<R_PAD, kOpF16, kModeMU, true>(a, grid, s);
    if (opt == kOpF16 && mode == kModeLoss) return launch_pp_one<R_PAD, kOpF16, kModeLoss, true>(a, grid, s);
    return -2;
  }
  if (opt == kOpBf16 && mode == kModeMU) return launch_pp_one<R_PAD, kOpBf16, kModeMU>(a, grid, s);
  if (opt == kOpF16 && mode == kModeMU) return launch_pp_one<R_PAD, kOpF16, kModeMU>(a, grid, s);
  if (opt == kOpBf16 && mode == kModeLoss) return launch_pp_one<R_PAD, kOpBf16, kModeLoss>(a, grid, s);
  if (opt == kOpF16 && mode == kModeLoss) return launch_pp_one<R_PAD, kOpF16, kModeLoss>(a, grid, s);
  return -2;
}

int launch_pp(int r_pad, int opt, int mode, const FusedArgs& a, int grid, hipStream_t s, bool xr, bool lacc) {
  switch (r_pad) {
    case 32: return launch_pp_r<32>(opt, mode, a, grid, s, xr, lacc);
    case 64: return launch_pp_r<64>(opt, mode, a, grid, s, xr, lacc);
    case 128: return launch_pp_r<128>(opt, mode, a, grid, s, xr, lacc);
  }
  return -2;
}

}  // namespace nmfmu
