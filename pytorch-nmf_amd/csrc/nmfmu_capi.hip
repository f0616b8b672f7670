// C ABI (include/nmfmu.h) over the HIP kernels.  Thin: argument checks, grid sizing, dispatch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

#include "../../include/nmfmu.h"
#include "nmfmu_aux.h"
#include "nmfmu_fused.h"
#include "nmfmu_pp.h"

using namespace nmfmu;

namespace {

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

static bool stage_is_reg(const nmfmu_step* st) { return st->stage == NMFMU_STAGE_REG; }

// Experiment hooks (read once): NMFMU_PP=0 routes beta = 1 / bf16 half-steps back to the four-wave kernels of
// nmfmu_fused.h; NMFMU_PP_VAR selects a build-time variant of the ping-pong kernel (nmfmu_pp.h, VAR bits).
static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}
static int pp_enabled() {
  static const int v = env_int("NMFMU_PP", 1);
  return v;
}
static void* g_pp_debug = nullptr;   // nmfmu_debug_set_buffer
static int pp_var() {
  static const int v = env_int("NMFMU_PP_VAR", 0);
  return v;
}
// which half-steps the ping-pong kernel serves: beta == 1, one operand plane (bf16 or fp16), padded rank <= 128
static bool pp_eligible(int r_pad, int precision, float beta) {
  if (nmfmu_beta_kind(beta) != NMFMU_BETA_KL || r_pad > 128) return false;
  if (precision == NMFMU_PREC_F16) return true;
  return precision == NMFMU_PREC_BF16 && pp_enabled();
}

int fused_dispatch(const nmfmu_step* st, int mode, float* loss_part, int M, int K, hipStream_t s,
                   const float* fuse_kl_den = nullptr, const float* fuse_kl_part = nullptr, int fuse_kl_nparts = 0) {
  if (!st || !st->owner.p1_hi || !st->panel.p1_hi) return NMFMU_ERR_ARG;
  if (!st->xp && mode == kModeMU) return NMFMU_ERR_ARG;   // (the denominator-only pass and the loss may run without a target)
  if (st->owner.rows_pad % kRowPad || st->panel.rows_pad % kRowPad) return NMFMU_ERR_ARG;
  if (st->block_rows != 128 && st->block_rows != 256) return NMFMU_ERR_ARG;
  const int G = st->block_rows / 128;
  if (st->r_pad != pad_rank(st->rank) || st->nsplit < 1) return NMFMU_ERR_ARG;
  const int x3 = st->precision == NMFMU_PREC_BF16X3 ? 1 : 0;
  if (x3 && (!st->owner.p1_lo || !st->panel.p1_lo)) return NMFMU_ERR_ARG;
  const int kind = nmfmu_beta_kind(st->beta);
  FusedArgs a;
  a.xp = st->xp;
  a.p1_hi = static_cast<const uint16_t*>(st->panel.p1_hi);
  a.p1_lo = static_cast<const uint16_t*>(st->panel.p1_lo);
  a.p2_hi = static_cast<const uint16_t*>(st->panel.p2_hi);
  a.p2_lo = static_cast<const uint16_t*>(st->panel.p2_lo);
  a.a1_hi = static_cast<const uint16_t*>(st->owner.p1_hi);
  a.a1_lo = static_cast<const uint16_t*>(st->owner.p1_lo);
  a.slab_num = st->slab_num;
  a.slab_den = st->slab_den;
  a.loss_part = loss_part;
  a.M = M;
  a.K = K;
  a.M_pad = st->owner.rows_pad;
  a.ktiles = st->panel.rows_pad / kBK;
  a.nsplit = st->nsplit;
  a.tiles_per_split = (a.ktiles + st->nsplit - 1) / st->nsplit;
  a.beta = st->beta;
  a.fuse_apply = 0;
  a.kl_part = nullptr, a.kl_nparts = 0;
  if (fuse_kl_den || fuse_kl_part) {  // beta == 1, nsplit == 1: apply in the epilogue
    a.fuse_apply = 1;
    a.rank = st->rank;
    a.f = st->owner.f;
    a.kl_den = fuse_kl_den;
    a.kl_part = fuse_kl_part, a.kl_nparts = fuse_kl_nparts;
    a.o1_hi = static_cast<uint16_t*>(st->owner.p1_hi), a.o1_lo = static_cast<uint16_t*>(st->owner.p1_lo);
    a.o2_hi = static_cast<uint16_t*>(st->owner.p2_hi), a.o2_lo = static_cast<uint16_t*>(st->owner.p2_lo);
    a.colsum_part = st->owner.colsum_part;
    a.l1 = st->l1, a.l2 = st->l2, a.gamma = st->gamma;
  }
  if (mode == kModeMU || mode == kModeDen) {
    if (!a.slab_num || !a.p2_hi || (x3 && !a.p2_lo)) return NMFMU_ERR_ARG;
    if (mode == kModeMU && kind != kKL && !a.slab_den) return NMFMU_ERR_ARG;
    if (mode == kModeDen && (kind != kGen || G != 1 || stage_is_reg(st))) return NMFMU_ERR_UNSUPPORTED;
  } else if (!loss_part) {
    return NMFMU_ERR_ARG;
  }
  const int grid = (st->owner.rows_pad / st->block_rows) * st->nsplit;
  const int stage = st->stage == NMFMU_STAGE_REG ? 0 : 1;
  if (G == 2 && stage == 1 && st->xp && (mode == kModeMU || mode == kModeLoss) &&
      pp_eligible(st->r_pad, st->precision, st->beta)) {
    const int opt = st->precision == NMFMU_PREC_F16 ? kOpF16 : kOpBf16;
    const int var = mode == kModeMU ? pp_var() : 0;
    if (!(var & 256)) a.tiles_per_split = (a.tiles_per_split + 1) & ~1;   // register-X tile loop is unrolled by two
    if (var & 128) {
      if (!g_pp_debug) return NMFMU_ERR_ARG;
      a.loss_part = static_cast<float*>(g_pp_debug);   // (unused by the MU mode otherwise)
    }
    return launch_pp(st->r_pad, opt, mode, var, a, grid, s);
  }
  if (st->precision == NMFMU_PREC_F16) return NMFMU_ERR_UNSUPPORTED;   // fp16 operands exist in the ping-pong kernel only
  if (a.kl_part) return NMFMU_ERR_UNSUPPORTED;                          // so do partial-sum denominators
  switch (st->r_pad) {
    case 32: return launch_fused_r32(kind, x3, mode, stage, G, a, grid, s);
    case 64: return launch_fused_r64(kind, x3, mode, stage, G, a, grid, s);
    case 128: return launch_fused_r128(kind, x3, mode, stage, G, a, grid, s);
    case 256: return launch_fused_r256(kind, x3, mode, stage, G, a, grid, s);
  }
  return NMFMU_ERR_UNSUPPORTED;
}

struct Timer {
  std::vector<hipEvent_t> ev;
};

}  // namespace

extern "C" {

int nmfmu_abi_version(void) { return NMFMU_ABI_VERSION; }
int nmfmu_pad_rows(int rows) { return rows <= 0 ? NMFMU_ERR_ARG : pad_rows(rows); }
int nmfmu_pad_rank(int rank) {
  const int r = pad_rank(rank);
  return r < 0 ? NMFMU_ERR_UNSUPPORTED : r;
}
int nmfmu_beta_kind(float beta) {
  if (beta == 1.f) return NMFMU_BETA_KL;
  if (beta == 2.f) return NMFMU_BETA_EUC;
  if (beta == 0.f) return NMFMU_BETA_IS;
  return NMFMU_BETA_GEN;
}
int nmfmu_supported(int r_pad, int precision) {
  if (r_pad != 32 && r_pad != 64 && r_pad != 128 && r_pad != 256) return 0;
  if (precision == NMFMU_PREC_BF16) return 1;
  if (precision == NMFMU_PREC_BF16X3) return r_pad <= 128;  // 4 image planes x 2 stages must fit 160 KiB of LDS
  if (precision == NMFMU_PREC_F16) return r_pad <= 128;     // ping-pong kernel only (beta == 1)
  return 0;
}

int nmfmu_block_rows(int r_pad, int precision, float beta) {
  // beta == 1 with one operand plane and padded rank <= 128: the eight-wave ping-pong kernel on 256-row tiles.
  // Everything else: 128-row tiles (two workgroups per CU) measure faster than the four-wave 256-row variant
  // (0.154 vs 0.179 ms per half-step at 4096x65536 r128); those stay built and selectable where has_g2() holds.
  if (pp_eligible(r_pad, precision, beta)) return 256;
#if NMFMU_SP
  if (r_pad == 128 && precision == NMFMU_PREC_BF16 && nmfmu_beta_kind(beta) == NMFMU_BETA_KL) return 256;
#endif
  return 128;
}

int nmfmu_choose_nsplit(int owner_rows_pad, int panel_rows_pad, int block_rows, int num_cu) {
  if (owner_rows_pad <= 0 || panel_rows_pad <= 0 || (block_rows != 128 && block_rows != 256)) return NMFMU_ERR_ARG;
  const int mblocks = owner_rows_pad / block_rows;
  const int ktiles = panel_rows_pad / kBK;
  static const int forced = env_int("NMFMU_FORCE_NSPLIT", 0);   // experiment hook
  if (forced > 0) return std::min(forced, std::max(1, ktiles));
  // 128-row tiles run two workgroups per CU (both wave slots of every SIMD); 256-row tiles run one.
  const int target = (block_rows == 128 ? 2 : 1) * std::max(num_cu, 1);
  int ns = (target + mblocks - 1) / mblocks;
  if (ns > 8) ns = (ns + 7) / 8 * 8;           // same-chunk workgroups then share an XCD (block b runs on XCD b % 8)
  ns = std::min(ns, std::max(1, ktiles / 4));  // at least 4 tiles per workgroup to amortise prologue/epilogue
  return std::max(ns, 1);
}

int nmfmu_step_block_rows(int owner_rows_pad, int panel_rows_pad, int r_pad, int precision, float beta, int num_cu) {
  if (pp_eligible(r_pad, precision, beta)) return 256;   // both half-steps (the apply is fused there as well)
  // Four-wave kernels: a half-step whose owner axis alone fills the chip with 128-row workgroups (no contraction
  // split) keeps the 128-row tile: the MU apply then runs in the epilogue, which measures faster at two workgroups
  // per CU (0.172 vs 0.178 ms at configs[1]'s W half-step).  Split half-steps take nmfmu_block_rows()'s tile.
  const int ns128 = nmfmu_choose_nsplit(owner_rows_pad, panel_rows_pad, 128, num_cu);
  if (ns128 < 0) return ns128;
  return ns128 == 1 ? 128 : nmfmu_block_rows(r_pad, precision, beta);
}

size_t nmfmu_xp_bytes(int owner_rows_pad, int panel_rows_pad, int precision) {
  return (size_t)owner_rows_pad * (size_t)panel_rows_pad * (precision == NMFMU_PREC_BF16X3 ? 4 : 2);
}
size_t nmfmu_image_bytes(int rows_pad, int r_pad) { return (size_t)rows_pad * (size_t)r_pad * 2; }
size_t nmfmu_slab_bytes(int owner_rows_pad, int r_pad, int nsplit) {
  return (size_t)nsplit * (size_t)owner_rows_pad * (size_t)r_pad * 4;
}
size_t nmfmu_colsum_part_bytes(int rows_pad, int r_pad) { return (size_t)(rows_pad / 16) * (size_t)r_pad * 4; }

int nmfmu_pack_x(const float* v, int64_t ld, int rows, int cols, int transpose, int precision, int block_rows, void* xp,
                 int owner_rows_pad, int panel_rows_pad, uint32_t* flags, void* stream) {
  if (!v || !xp || rows <= 0 || cols <= 0 || (block_rows != 128 && block_rows != 256)) return NMFMU_ERR_ARG;
  const int m = transpose ? cols : rows, k = transpose ? rows : cols;
  if (owner_rows_pad != pad_rows(m) || panel_rows_pad != pad_rows(k)) return NMFMU_ERR_ARG;
  const int fmt = precision == NMFMU_PREC_BF16X3 ? 1 : (precision == NMFMU_PREC_F16 ? 2 : 0);
  return launch_pack_x(v, ld, rows, cols, transpose != 0, fmt, xp, owner_rows_pad, panel_rows_pad, flags,
                       block_rows / 128, S(stream));
}

static int pack_factor_common(const nmfmu_factor* fac, int rank, int r_pad, int precision, const float* scale,
                              void* stream) {
  if (!fac || !fac->f || !fac->p1_hi || !fac->p2_hi || !fac->colsum || !fac->colsum_part) return NMFMU_ERR_ARG;
  if (r_pad != pad_rank(rank) || fac->rows_pad != pad_rows(fac->rows)) return NMFMU_ERR_ARG;
  const bool x3 = precision == NMFMU_PREC_BF16X3;
  if (x3 && (!fac->p1_lo || !fac->p2_lo)) return NMFMU_ERR_ARG;
  ApplyArgs a{};
  a.f = fac->f;
  a.p1_hi = fac->p1_hi, a.p1_lo = fac->p1_lo, a.p2_hi = fac->p2_hi, a.p2_lo = fac->p2_lo;
  a.colsum_part = fac->colsum_part, a.colsum = fac->colsum;
  a.rows = fac->rows, a.rank = rank, a.rows_pad = fac->rows_pad;
  a.gamma = 1.f;
  a.scale = scale;
  a.f16 = precision == NMFMU_PREC_F16;
  return launch_apply(r_pad, a, x3, /*pack_only=*/true, S(stream));
}

int nmfmu_pack_factor(const nmfmu_factor* fac, int rank, int r_pad, int precision, void* stream) {
  return pack_factor_common(fac, rank, r_pad, precision, nullptr, stream);
}

int nmfmu_pack_factor_scaled(const nmfmu_factor* fac, int rank, int r_pad, int precision, const float* scale,
                             void* stream) {
  if (!scale) return NMFMU_ERR_ARG;
  return pack_factor_common(fac, rank, r_pad, precision, scale, stream);
}

int nmfmu_mu_partial(const nmfmu_step* st, void* stream) {
  if (!st) return NMFMU_ERR_ARG;
  if (!nmfmu_supported(st->r_pad, st->precision)) return NMFMU_ERR_UNSUPPORTED;
  return fused_dispatch(st, kModeMU, nullptr, st->owner.rows, st->panel.rows, S(stream));
}

int nmfmu_den_partial(const nmfmu_step* st, void* stream) {
  if (!st) return NMFMU_ERR_ARG;
  if (!nmfmu_supported(st->r_pad, st->precision)) return NMFMU_ERR_UNSUPPORTED;
  return fused_dispatch(st, kModeDen, nullptr, st->owner.rows, st->panel.rows, S(stream));
}

int nmfmu_mu_step(const nmfmu_step* st, const float* kl_den, int phase, void* stream) {
  if (!st || phase < 0 || phase > 2) return NMFMU_ERR_ARG;
  if (!nmfmu_supported(st->r_pad, st->precision)) return NMFMU_ERR_UNSUPPORTED;
  const bool kl = nmfmu_beta_kind(st->beta) == NMFMU_BETA_KL;
  if (kl && !kl_den) return NMFMU_ERR_ARG;
  const bool fuse = kl && st->nsplit == 1 && st->owner.f && st->owner.p2_hi && st->owner.colsum && st->owner.colsum_part;
  int e = 0;
  if (phase != 2) {  // the fused kernel (with nmf.py:78-92 in its epilogue when the workgroup owns whole rows)
    e = fuse ? fused_dispatch(st, kModeMU, nullptr, st->owner.rows, st->panel.rows, S(stream), kl_den)
             : nmfmu_mu_partial(st, stream);
    if (e) return e;
  }
  if (phase != 1) {  // what is left: the column-sum finalize, or the whole apply
    e = fuse ? launch_colsum_finalize(st->owner.colsum_part, st->owner.rows_pad / st->block_rows, st->r_pad,
                                      st->owner.colsum, S(stream))
             : nmfmu_mu_apply(st, nullptr, nullptr, 0, kl ? kl_den : nullptr, stream);
  }
  return e;
}

static int apply_common(const nmfmu_step* st, const float* num, const float* den, int nslab, const float* kl_den,
                        int trainer, float ortho, float* grad, void* stream, const float* kl_part, int kl_nparts,
                        int skip_finalize);

int nmfmu_parts_supported(const nmfmu_step* st) {
  if (!st) return 0;
  return st->block_rows == 256 && st->stage == NMFMU_STAGE_DMA && pp_eligible(st->r_pad, st->precision, st->beta) ? 1 : 0;
}

int nmfmu_colsum_nparts(const nmfmu_step* st) {
  if (!st || (st->block_rows != 128 && st->block_rows != 256)) return NMFMU_ERR_ARG;
  // fused apply (nsplit == 1): one partial per workgroup tile; otherwise one per apply-kernel stripe
  return st->nsplit == 1 ? st->owner.rows_pad / st->block_rows : st->owner.rows_pad / apply_stripe_rows(st->owner.rows_pad);
}

int nmfmu_pack_nparts(int rows_pad) { return rows_pad <= 0 ? NMFMU_ERR_ARG : rows_pad / apply_stripe_rows(rows_pad); }

int nmfmu_colsum_finalize(const nmfmu_factor* fac, int nparts, int r_pad, void* stream) {
  if (!fac || !fac->colsum || !fac->colsum_part || nparts <= 0) return NMFMU_ERR_ARG;
  return launch_colsum_finalize(fac->colsum_part, nparts, r_pad, fac->colsum, S(stream));
}

int nmfmu_mu_step_parts(const nmfmu_step* st, const float* kl_part, int kl_nparts, int phase, void* stream) {
  if (!st || phase < 0 || phase > 2 || !kl_part || kl_nparts <= 0) return NMFMU_ERR_ARG;
  if (!nmfmu_parts_supported(st)) return NMFMU_ERR_UNSUPPORTED;
  const bool fuse = st->nsplit == 1 && st->owner.f && st->owner.p2_hi && st->owner.colsum_part;
  int e = 0;
  if (phase != 2) {
    e = fuse ? fused_dispatch(st, kModeMU, nullptr, st->owner.rows, st->panel.rows, S(stream), nullptr, kl_part, kl_nparts)
             : nmfmu_mu_partial(st, stream);
    if (e) return e;
  }
  if (phase != 1 && !fuse)
    e = apply_common(st, nullptr, nullptr, 0, nullptr, 0, 0.f, nullptr, stream, kl_part, kl_nparts, /*skip_finalize=*/1);
  return e;
}

int nmfmu_slab_reduce(const nmfmu_step* st, float* num_out, float* den_out, void* stream) {
  if (!st || !num_out || !st->slab_num) return NMFMU_ERR_ARG;
  const int64_t plane = (int64_t)st->owner.rows_pad * st->r_pad;
  int e = launch_slab_reduce(st->slab_num, st->nsplit, plane, num_out, S(stream));
  if (e) return e;
  if (den_out) {
    if (!st->slab_den) return NMFMU_ERR_ARG;
    e = launch_slab_reduce(st->slab_den, st->nsplit, plane, den_out, S(stream));
  }
  return e;
}

static int apply_common(const nmfmu_step* st, const float* num, const float* den, int nslab, const float* kl_den,
                        int trainer, float ortho, float* grad, void* stream, const float* kl_part,
                        int kl_nparts, int skip_finalize) {
  if (!st || !st->owner.f) return NMFMU_ERR_ARG;
  const bool kl = nmfmu_beta_kind(st->beta) == NMFMU_BETA_KL;
  ApplyArgs a{};
  a.f = st->owner.f;
  a.num = num ? num : st->slab_num;
  a.den = num ? den : st->slab_den;
  a.nslab = num ? nslab : st->nsplit;
  a.kl_den = kl ? kl_den : nullptr;
  a.kl_part = kl ? kl_part : nullptr, a.kl_nparts = kl_nparts, a.skip_finalize = skip_finalize;
  if (!a.num || a.nslab < 1) return NMFMU_ERR_ARG;
  if (kl ? (!a.kl_den && !a.kl_part) : !a.den) return NMFMU_ERR_ARG;
  a.p1_hi = st->owner.p1_hi, a.p1_lo = st->owner.p1_lo, a.p2_hi = st->owner.p2_hi, a.p2_lo = st->owner.p2_lo;
  a.colsum_part = st->owner.colsum_part, a.colsum = st->owner.colsum;
  a.rows = st->owner.rows, a.rank = st->rank, a.rows_pad = st->owner.rows_pad;
  a.l1 = st->l1, a.l2 = st->l2, a.gamma = st->gamma;
  a.trainer = trainer, a.ortho = ortho, a.grad = grad;
  a.f16 = st->precision == NMFMU_PREC_F16;
  if (!a.p1_hi || !a.p2_hi || !a.colsum || !a.colsum_part) return NMFMU_ERR_ARG;
  return launch_apply(st->r_pad, a, st->precision == NMFMU_PREC_BF16X3, /*pack_only=*/false, S(stream));
}

int nmfmu_mu_apply(const nmfmu_step* st, const float* num, const float* den, int nslab, const float* kl_den,
                   void* stream) {
  return apply_common(st, num, den, nslab, kl_den, 0, 0.f, nullptr, stream, nullptr, 0, 0);
}

int nmfmu_trainer_apply(const nmfmu_step* st, const float* num, const float* den, int nslab, const float* kl_den,
                        float ortho, float* grad, void* stream) {
  if (!(ortho >= 0.f)) return NMFMU_ERR_ARG;
  return apply_common(st, num, den, nslab, kl_den, 1, ortho, grad, stream, nullptr, 0, 0);
}

int nmfmu_loss_part_count(int owner_rows_pad, int block_rows, int nsplit) {
  if (block_rows != 128 && block_rows != 256) return NMFMU_ERR_ARG;
  return (owner_rows_pad / block_rows) * nsplit;
}

int nmfmu_loss(const nmfmu_step* st, float* loss_part, double* out, void* stream) {
  if (!st || !out) return NMFMU_ERR_ARG;
  if (!nmfmu_supported(st->r_pad, st->precision)) return NMFMU_ERR_UNSUPPORTED;
  int e = fused_dispatch(st, kModeLoss, loss_part, st->owner.rows, st->panel.rows, S(stream));
  if (e) return e;
  return launch_sum_finalize_f32(loss_part, (st->owner.rows_pad / st->block_rows) * st->nsplit, out, S(stream));
}

int nmfmu_beta_div(const float* x, const float* y, int64_t n, float beta, double* part, double* out, void* stream) {
  if (!x || !y || !part || !out || n < 0) return NMFMU_ERR_ARG;
  return launch_beta_div(x, y, n, beta, nmfmu_beta_kind(beta), part, out, S(stream));
}

int nmfmu_mu_terms(const float* s, const float* v, int64_t n, float beta, float* gn, float* gp, void* stream) {
  if (!s || !v || !gn || !gp || n <= 0) return NMFMU_ERR_ARG;
  return launch_mu_terms(s, v, n, beta, nmfmu_beta_kind(beta), gn, gp, S(stream));
}

int nmfmu_trainer_update(float* f, int rows, int cols, const float* neg, const float* pos, float l1, float l2, float ortho,
                         float gamma, float* grad, void* stream) {
  if (!f || !neg || !pos || rows <= 0 || cols <= 0 || !(ortho >= 0.f)) return NMFMU_ERR_ARG;
  return launch_trainer_update(f, rows, cols, neg, pos, l1, l2, ortho, gamma, grad, S(stream));
}

int nmfmu_norms(const float* x, int64_t n, double* part, double* out, void* stream) {
  if (!x || !part || !out || n <= 0) return NMFMU_ERR_ARG;
  return launch_norms(x, n, part, out, S(stream));
}

int nmfmu_reconstruct(const float* owner, int m, const float* panel, int k, int rank, float* out, int64_t ld,
                      void* stream) {
  if (!owner || !panel || !out || m <= 0 || k <= 0 || rank <= 0 || ld < k) return NMFMU_ERR_ARG;
  return launch_reconstruct(owner, m, panel, k, rank, out, ld, S(stream));
}

int nmfmu_timer_create(int n_events, void** timer) {
  if (n_events <= 0 || !timer) return NMFMU_ERR_ARG;
  Timer* t = new (std::nothrow) Timer;
  if (!t) return NMFMU_ERR_ARG;
  t->ev.resize(n_events);
  for (auto& e : t->ev) {
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return (int)r;
  }
  *timer = t;
  return NMFMU_OK;
}
int nmfmu_timer_record(void* timer, int idx, void* stream) {
  Timer* t = static_cast<Timer*>(timer);
  if (!t || idx < 0 || idx >= (int)t->ev.size()) return NMFMU_ERR_ARG;
  return (int)hipEventRecord(t->ev[idx], S(stream));
}
int nmfmu_timer_elapsed_ms(void* timer, int idx_from, int idx_to, float* ms) {
  Timer* t = static_cast<Timer*>(timer);
  if (!t || !ms || idx_from < 0 || idx_to < 0 || idx_from >= (int)t->ev.size() || idx_to >= (int)t->ev.size())
    return NMFMU_ERR_ARG;
  hipError_t r = hipEventSynchronize(t->ev[idx_to]);
  if (r != hipSuccess) return (int)r;
  return (int)hipEventElapsedTime(ms, t->ev[idx_from], t->ev[idx_to]);
}
int nmfmu_timer_destroy(void* timer) {
  Timer* t = static_cast<Timer*>(timer);
  if (!t) return NMFMU_ERR_ARG;
  for (auto& e : t->ev) hipEventDestroy(e);
  delete t;
  return NMFMU_OK;
}

int nmfmu_debug_set_buffer(void* buf) {
  g_pp_debug = buf;
  return NMFMU_OK;
}

int nmfmu_probe_mfma(const uint16_t* a, const uint16_t* b, float* d, void* stream) {
  if (!a || !b || !d) return NMFMU_ERR_ARG;
  return launch_probe_mfma(a, b, d, S(stream));
}
int nmfmu_probe_lds_dma(const uint32_t* src, uint32_t* dst, int n_dwords, void* stream) {
  if (!src || !dst || n_dwords <= 0 || n_dwords % 1024 || n_dwords > 16384) return NMFMU_ERR_ARG;
  return launch_probe_lds_dma(src, dst, n_dwords, S(stream));
}

}  // extern "C"
