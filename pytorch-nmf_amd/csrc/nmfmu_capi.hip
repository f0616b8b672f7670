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
#include "nmfmu_sp.h"
#include "nmfmu_sp2.h"

#ifndef NMFMU_SP
#define NMFMU_SP 1   // (A/B switch of round 6; 0 = padded rank 256, beta == 1, fp16 stays on the four-wave kernel)
#endif
#ifndef NMFMU_SP2
#define NMFMU_SP2 1  // (A/B switch of round 6; 0 = padded rank 128, beta in {0, 0.5, 1.5, generic}, fp16 stays on the four-wave kernel)
#endif
#ifndef NMFMU_FUSE_APPLY_TWO_ACC
#define NMFMU_FUSE_APPLY_TWO_ACC 1   // (A/B switch of round 3; 0 = beta != 1 always goes through slabs + the apply kernel)
#endif

using namespace nmfmu;

namespace {

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool is_f16(int precision) {   // fp16 operand images
  return precision == NMFMU_PREC_F16 || precision == NMFMU_PREC_F16X || precision == NMFMU_PREC_F16R;
}
inline bool x_is_f32(int precision) { return precision == NMFMU_PREC_BF16X3 || precision == NMFMU_PREC_F16X; }

// Diagnostic builds only (make EXTRA=-DNMFMU_DEBUG_HOOKS): NMFMU_FORCE_NSPLIT overrides the contraction split and
// nmfmu_debug_set_buffer registers the clock-stamp buffer of the ping-pong kernel (tools/pp_timeline.py).  The product
// library reads no environment variable and keeps no global state.
#ifdef NMFMU_DEBUG_HOOKS
static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}
static void* g_pp_debug = nullptr;
#endif
#ifndef NMFMU_PP_F16R
#define NMFMU_PP_F16R 1   // 0: the 3-byte target stays on the four-wave kernel at beta == 1 as well (A/B builds)
#endif
// which half-steps the ping-pong kernel serves: beta == 1, one operand plane (bf16 or fp16; fp16 with the 3-byte target),
// padded rank <= 128
static bool pp_eligible(int r_pad, int precision, float beta) {
  return nmfmu_beta_kind(beta) == NMFMU_BETA_KL && r_pad <= 128 &&
         (precision == NMFMU_PREC_F16 || precision == NMFMU_PREC_BF16 || (NMFMU_PP_F16R && precision == NMFMU_PREC_F16R));
}
// which half-steps the software-pipelined one-wave-per-SIMD kernel serves (nmfmu_sp.h): beta == 1, fp16 operands and target,
// padded rank 256 -- the kernel of configs[4]'s shard
static bool sp_eligible(int r_pad, int precision, float beta) {
  return NMFMU_SP && nmfmu_beta_kind(beta) == NMFMU_BETA_KL && r_pad == 256 && precision == NMFMU_PREC_F16;
}
// ... and its two-accumulator sibling (nmfmu_sp2.h): beta != 1 and != 2, fp16 operands and target, padded rank 128 -- the
// kernel of configs[2]'s beta < 1 legs
static bool sp2_eligible(int r_pad, int precision, float beta) {
  const int k = nmfmu_beta_kind(beta);
  return NMFMU_SP2 && (k == NMFMU_BETA_IS || k == NMFMU_BETA_GEN) && r_pad == 128 && precision == NMFMU_PREC_F16;
}
// beta -> kernel branch (nmfmu_fused.h: BetaKind): the public kinds plus the two rsqrt special cases of the generic one
static int kernel_beta_kind(float beta) {
  if (beta == 0.5f) return kSqrt;
  if (beta == 1.5f) return kSqrt3;
  return nmfmu_beta_kind(beta);
}

struct GramRef {   // Gram matrix of the panel (nmfmu_gram_panel): fp32 matrix + 16-bit images with per-row scales
  const float* gram;
  const void* hi;
  const void* lo;
  const float* scale;
};

int fused_dispatch(const nmfmu_step* st, int mode, float* loss_part, int M, int K, hipStream_t s,
                   const float* fuse_kl_den = nullptr, bool fuse_apply = false, const GramRef* gm = nullptr) {
  if (!st || !st->owner.p1_hi || !st->panel.p1_hi) return NMFMU_ERR_ARG;
  if (!st->xp && (mode == kModeMU || mode == kModeXB)) return NMFMU_ERR_ARG;   // (the denominator-only pass and the loss may run without a target)
  if (st->owner.rows_pad % kRowPad || st->panel.rows_pad % kRowPad) return NMFMU_ERR_ARG;
  if (st->block_rows != 128 && st->block_rows != 256) return NMFMU_ERR_ARG;
  if (st->stage != NMFMU_STAGE_DMA && st->stage != NMFMU_STAGE_DMA_SPLIT && st->stage != NMFMU_STAGE_DMA_NOP2)
    return NMFMU_ERR_UNSUPPORTED;   // (register staging: no longer built)
  // split panel (PLCA: p1 = the Z-scaled factor, p2 = the unscaled one): the instantiation that stages BOTH images
  if (st->stage == NMFMU_STAGE_DMA_SPLIT && mode == kModeMU && !(st->precision == NMFMU_PREC_BF16X3)) {
    if (nmfmu_beta_kind(st->beta) != NMFMU_BETA_KL || st->block_rows != 128) return NMFMU_ERR_UNSUPPORTED;
    mode = kModeMU2;
  }
  if (st->r_pad != pad_rank(st->rank) || st->nsplit < 1) return NMFMU_ERR_ARG;
  const bool x3 = st->precision == NMFMU_PREC_BF16X3;
  if (x3 && (!st->owner.p1_lo || !st->panel.p1_lo)) return NMFMU_ERR_ARG;
  const int kind = nmfmu_beta_kind(st->beta);
  if (st->precision == NMFMU_PREC_F16R && kind == NMFMU_BETA_EUC) return NMFMU_ERR_UNSUPPORTED;   // the target is an operand there: F16X
  FusedArgs a{};
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
  a.status = st->status;
  a.cs_owner = st->owner.colsum;   // fp16 operands, beta < 1: typical S -> scale of Gn / Gp (nmfmu_fused.h)
  a.cs_panel = st->panel.colsum;
  if (fuse_apply) {  // nsplit == 1: apply in the epilogue
    a.fuse_apply = 1;
    a.rank = st->rank;
    a.f = st->owner.f;
    a.kl_den = fuse_kl_den;
    a.o1_hi = static_cast<uint16_t*>(st->owner.p1_hi), a.o1_lo = static_cast<uint16_t*>(st->owner.p1_lo);
    a.o2_hi = static_cast<uint16_t*>(st->owner.p2_hi), a.o2_lo = static_cast<uint16_t*>(st->owner.p2_lo);
    a.colsum_part = st->owner.colsum_part;
    a.l1 = st->l1, a.l2 = st->l2, a.gamma = st->gamma;
  }
  if (mode == kModeMU2) {
    if (!a.slab_num || !a.p2_hi || fuse_apply) return NMFMU_ERR_ARG;
  } else if (mode == kModeXB) {   // beta == 2 without reconstruction: numerator slabs only (or the fused apply with the Gram images)
    if (kind != kEuc || x3 || st->block_rows != 128) return NMFMU_ERR_UNSUPPORTED;
    if (!a.p2_hi || (!fuse_apply && !a.slab_num)) return NMFMU_ERR_ARG;
    if (fuse_apply && (!gm || st->r_pad > 128)) return NMFMU_ERR_ARG;
    if (gm) {   // denominator in the kernel: fused apply, or ONE denominator slab, its rank tiles shared out over the row block's first workgroups
      if (!gm->hi || !gm->lo || !gm->scale || (!fuse_apply && !a.slab_den)) return NMFMU_ERR_ARG;
      a.gram_hi = static_cast<const uint16_t*>(gm->hi), a.gram_lo = static_cast<const uint16_t*>(gm->lo);
      a.gram_scale = gm->scale;
    } else {
      a.slab_den = nullptr;   // numerator slabs only (nmfmu_xb_partial)
    }
  } else if (mode == kModeMU || mode == kModeDen) {
    if (!a.slab_num || !a.p2_hi || (x3 && !a.p2_lo)) return NMFMU_ERR_ARG;
    if (mode == kModeMU && kind != kKL && !a.slab_den) return NMFMU_ERR_ARG;
    if (mode == kModeDen && (kind != kGen || st->block_rows != 128)) return NMFMU_ERR_UNSUPPORTED;
  } else if (!loss_part) {
    return NMFMU_ERR_ARG;
  }
  const int grid = (st->owner.rows_pad / st->block_rows) * st->nsplit;
  if (st->block_rows == 256) {   // the eight-wave ping-pong kernel (beta == 1, one operand plane, padded rank <= 128)
    if (!st->xp || mode == kModeDen || !pp_eligible(st->r_pad, st->precision, st->beta)) return NMFMU_ERR_UNSUPPORTED;
    a.tiles_per_split = (a.tiles_per_split + 1) & ~1;   // its tile loop is unrolled by two
    a.debug = mode == kModeMU ? st->stamps : nullptr;
#ifdef NMFMU_DEBUG_HOOKS
    if (mode == kModeMU && g_pp_debug) a.debug = g_pp_debug;
#endif
    return launch_pp(st->r_pad, is_f16(st->precision) ? kOpF16 : kOpBf16, mode, a, grid, s, st->precision == NMFMU_PREC_F16R,
                     /*riding loss*/ mode == kModeMU && loss_part != nullptr);
  }
  if (mode == kModeMU && sp_eligible(st->r_pad, st->precision, st->beta)) {
    a.tiles_per_split = (a.tiles_per_split + 3) & ~3;   // its tile loop runs in groups of four (ring slot = tile & 3)
    a.debug = st->stamps;
#ifdef NMFMU_DEBUG_HOOKS
    if (g_pp_debug) a.debug = g_pp_debug;
#endif
    if (st->stage == NMFMU_STAGE_DMA_NOP2) a.o2_hi = nullptr;   // nobody reads the owner's transposed image: the epilogue skips it
    return launch_sp(st->r_pad, kOpF16, a, grid, s);
  }
  if (mode == kModeMU && sp2_eligible(st->r_pad, st->precision, st->beta)) {
    a.tiles_per_split = (a.tiles_per_split + 3) & ~3;
    return launch_sp2(st->r_pad, kernel_beta_kind(st->beta), a, grid, s);
  }
  const int kk = kernel_beta_kind(st->beta);
#ifdef NMFMU_DEBUG_HOOKS
  a.debug = mode == kModeXB ? g_pp_debug : nullptr;   // per-workgroup phase stamps of the streaming kernel (tools/xb_timeline.py)
#endif
  switch (st->r_pad) {
    case 32: return launch_fused_r32(kk, st->precision, mode, a, grid, s);
    case 64: return launch_fused_r64(kk, st->precision, mode, a, grid, s);
    case 128: return launch_fused_r128(kk, st->precision, mode, a, grid, s);
    case 256: return launch_fused_r256(kk, st->precision, mode, a, grid, s);
  }
  return NMFMU_ERR_UNSUPPORTED;
}

struct Timer {
  std::vector<hipEvent_t> ev;
};

}  // namespace

extern "C" {

int nmfmu_abi_version(void) { return NMFMU_ABI_VERSION; }
int nmfmu_abi_check(int compiled_against) { return compiled_against == NMFMU_ABI_VERSION ? NMFMU_OK : NMFMU_ERR_ARG; }
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
  if (precision == NMFMU_PREC_F16) return 1;                // ping-pong kernel (beta == 1, r_pad <= 128), else four-wave
  if (precision == NMFMU_PREC_F16X) return 1;               // four-wave kernel: fp16 operands, fp32 target
  if (precision == NMFMU_PREC_F16R) return 1;               // as F16 with a 3-byte target (beta != 2): ping-pong / four-wave kernel
  return 0;
}

int nmfmu_block_rows(int r_pad, int precision, float beta) {
  // beta == 1 with one operand plane and padded rank <= 128: the eight-wave ping-pong kernel on 256-row tiles.
  // Everything else: the four-wave kernel on 128-row tiles (two workgroups per CU where the registers allow).
  return pp_eligible(r_pad, precision, beta) ? 256 : 128;
}

int nmfmu_choose_nsplit(int owner_rows_pad, int panel_rows_pad, int block_rows, int num_cu) {
  if (owner_rows_pad <= 0 || panel_rows_pad <= 0 || (block_rows != 128 && block_rows != 256)) return NMFMU_ERR_ARG;
  const int mblocks = owner_rows_pad / block_rows;
  const int ktiles = panel_rows_pad / kBK;
#ifdef NMFMU_DEBUG_HOOKS
  static const int forced = env_int("NMFMU_FORCE_NSPLIT", 0);
  if (forced > 0) return std::min(forced, std::max(1, ktiles));
#endif
  // 128-row tiles run two workgroups per CU (both wave slots of every SIMD); 256-row tiles run one.
  const int target = (block_rows == 128 ? 2 : 1) * std::max(num_cu, 1);
  int ns = (target + mblocks - 1) / mblocks;
  if (ns > 8) ns = (ns + 7) / 8 * 8;           // same-chunk workgroups then share an XCD (block b runs on XCD b % 8)
  ns = std::min(ns, std::max(1, ktiles / 4));  // at least 4 tiles per workgroup to amortise prologue/epilogue
  return std::max(ns, 1);
}

int nmfmu_kernel_family(int r_pad, int precision, float beta) {
  if (pp_eligible(r_pad, precision, beta)) return NMFMU_KERNEL_PP;
  if (sp_eligible(r_pad, precision, beta) || sp2_eligible(r_pad, precision, beta)) return NMFMU_KERNEL_SP;
  return NMFMU_KERNEL_FUSED;
}

int nmfmu_choose_nsplit_for(int owner_rows_pad, int panel_rows_pad, int r_pad, int precision, float beta, int block_rows, int num_cu) {
  if (owner_rows_pad <= 0 || panel_rows_pad <= 0 || (block_rows != 128 && block_rows != 256)) return NMFMU_ERR_ARG;
  if (block_rows == 128 && (sp_eligible(r_pad, precision, beta) || sp2_eligible(r_pad, precision, beta))) {
    // one 512-register workgroup per CU: as many splits as fill the chip ONCE (whole rounds when the row blocks alone
    // exceed it), each a multiple of four tiles
    const int mblocks = owner_rows_pad / 128, ktiles = panel_rows_pad / kBK;
    int ns = (std::max(num_cu, 1) + mblocks - 1) / mblocks;
    ns = std::min(ns, std::max(1, ktiles / 8));
    return std::max(ns, 1);
  }
  return nmfmu_choose_nsplit(owner_rows_pad, panel_rows_pad, block_rows, num_cu);
}

int nmfmu_step_block_rows(int owner_rows_pad, int panel_rows_pad, int r_pad, int precision, float beta, int num_cu) {
  if (pp_eligible(r_pad, precision, beta)) return 256;   // both half-steps (the apply is fused there as well)
  if (owner_rows_pad <= 0 || panel_rows_pad <= 0) return NMFMU_ERR_ARG;
  (void)num_cu;
  return 128;                                              // four-wave kernel
}

size_t nmfmu_xp_bytes(int owner_rows_pad, int panel_rows_pad, int precision) {
  return (size_t)owner_rows_pad * (size_t)panel_rows_pad * (x_is_f32(precision) ? 4 : (precision == NMFMU_PREC_F16R ? 3 : 2));
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
  const int fmt = x_is_f32(precision) ? 1 : (precision == NMFMU_PREC_F16 ? 2 : (precision == NMFMU_PREC_F16R ? 3 : 0));
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
  a.skip_colsum = scale != nullptr;   // the scaled packing serves PLCA's EM, which reads no column sums of the images' factor
  a.f16 = is_f16(precision);
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
  // (a split panel -- images of two different matrices -- belongs to nmfmu_mu_partial: PLCA's EM reads the slabs; a complete
  // MU half-step of ONE panel matrix has nothing to split, and the fused-apply epilogue stages p1 only)
  if (st->stage == NMFMU_STAGE_DMA_SPLIT) return NMFMU_ERR_ARG;
  const bool kl = nmfmu_beta_kind(st->beta) == NMFMU_BETA_KL;
  if (kl && !kl_den) return NMFMU_ERR_ARG;
  // apply in the fused kernel's epilogue when the workgroup owns whole rows: beta == 1 (closed-form denominators) on
  // both kernels, beta != 1 (two accumulator sets) on the four-wave kernel in the single-plane precisions up to rank pad 128
  const bool fuse = st->nsplit == 1 && st->owner.f && st->owner.p2_hi && st->owner.colsum && st->owner.colsum_part &&
                    (kl || (NMFMU_FUSE_APPLY_TWO_ACC && st->precision != NMFMU_PREC_BF16X3 && st->r_pad <= 128));
  int e = 0;
  if (phase != 2) {  // the fused kernel (with nmf.py:78-92 in its epilogue when the workgroup owns whole rows)
    e = fuse ? fused_dispatch(st, kModeMU, nullptr, st->owner.rows, st->panel.rows, S(stream), kl_den, true)
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
                        int trainer, float ortho, float* grad, void* stream, int den_nslab = 0, int skip_colsum = 0) {
  if (!st || !st->owner.f) return NMFMU_ERR_ARG;
  const bool kl = nmfmu_beta_kind(st->beta) == NMFMU_BETA_KL;
  ApplyArgs a{};
  a.f = st->owner.f;
  a.num = num ? num : st->slab_num;
  a.den = num ? den : st->slab_den;
  a.nslab = num ? nslab : st->nsplit;
  a.kl_den = kl ? kl_den : nullptr;
  if (!a.num || a.nslab < 1) return NMFMU_ERR_ARG;
  a.den_nslab = den_nslab;
  a.skip_colsum = skip_colsum;
  if (kl ? !a.kl_den : !a.den) return NMFMU_ERR_ARG;
  a.p1_hi = st->owner.p1_hi, a.p1_lo = st->owner.p1_lo, a.p2_hi = st->owner.p2_hi, a.p2_lo = st->owner.p2_lo;
  a.colsum_part = st->owner.colsum_part, a.colsum = st->owner.colsum;
  a.rows = st->owner.rows, a.rank = st->rank, a.rows_pad = st->owner.rows_pad;
  a.l1 = st->l1, a.l2 = st->l2, a.gamma = st->gamma;
  a.trainer = trainer, a.ortho = ortho, a.grad = grad;
  a.f16 = is_f16(st->precision);
  a.status = st->status;
  if (!a.p1_hi || !a.p2_hi || !a.colsum || !a.colsum_part) return NMFMU_ERR_ARG;
  if (st->stage == NMFMU_STAGE_DMA_NOP2 && sp_eligible(st->r_pad, st->precision, st->beta)) a.p2_hi = a.p2_lo = nullptr;   // (see nmfmu.h)
  return launch_apply(st->r_pad, a, st->precision == NMFMU_PREC_BF16X3, /*pack_only=*/false, S(stream));
}

int nmfmu_mu_apply(const nmfmu_step* st, const float* num, const float* den, int nslab, const float* kl_den,
                   void* stream) {
  return apply_common(st, num, den, nslab, kl_den, 0, 0.f, nullptr, stream);
}

int nmfmu_trainer_apply(const nmfmu_step* st, const float* num, const float* den, int nslab, const float* kl_den,
                        float ortho, float* grad, void* stream) {
  if (!(ortho >= 0.f)) return NMFMU_ERR_ARG;
  return apply_common(st, num, den, nslab, kl_den, 1, ortho, grad, stream);
}

int nmfmu_xb_supported(int r_pad, int precision, float beta) {
  if (nmfmu_beta_kind(beta) != NMFMU_BETA_EUC) return 0;
  if (r_pad != 32 && r_pad != 64 && r_pad != 128 && r_pad != 256) return 0;
  if (precision == NMFMU_PREC_F16X) return r_pad <= 128;   // (three fp32 X buffers + rank-256 accumulators: no registers)
  return precision == NMFMU_PREC_BF16 || precision == NMFMU_PREC_F16;
}

int nmfmu_xb_partial(const nmfmu_step* st, void* stream) {
  if (!st) return NMFMU_ERR_ARG;
  if (!nmfmu_xb_supported(st->r_pad, st->precision, st->beta)) return NMFMU_ERR_UNSUPPORTED;
  return fused_dispatch(st, kModeXB, nullptr, st->owner.rows, st->panel.rows, S(stream));
}

int nmfmu_xb_step(const nmfmu_step* st, const void* g_hi, const void* g_lo, const float* g_scale, int phase, void* stream) {
  if (!st || !g_hi || !g_lo || !g_scale || phase < 0 || phase > 2) return NMFMU_ERR_ARG;
  if (!nmfmu_xb_supported(st->r_pad, st->precision, st->beta)) return NMFMU_ERR_UNSUPPORTED;
  // the workgroup owns whole rows (unsplit contraction): numerator, denominator (owner fragments x Gram image) and
  // nmf.py:78-92 in the kernel's epilogue; otherwise nsplit numerator slabs + ONE denominator slab (written by the first
  // min(nsplit, r_pad / 32) workgroups of every row block, one 32-rank tile each, the same MFMA product) + the apply kernel
  const bool fuse = st->nsplit == 1 && st->owner.f && st->owner.p2_hi && st->owner.colsum && st->owner.colsum_part &&
                    st->r_pad <= 128;
  if (!fuse && !st->slab_den) return NMFMU_ERR_ARG;
  const GramRef gm{nullptr, g_hi, g_lo, g_scale};
  int e = 0;
  if (phase != 2) {
    e = fused_dispatch(st, kModeXB, nullptr, st->owner.rows, st->panel.rows, S(stream), nullptr, fuse, &gm);
    if (e) return e;
  }
  // (no column-sum finalize on this path: beta == 2 reads neither the closed-form denominators nor the fp16 scale)
  if (phase != 1 && !fuse)
    e = apply_common(st, nullptr, nullptr, 0, nullptr, 0, 0.f, nullptr, stream, /*den_nslab=*/1, /*skip_colsum=*/1);
  return e;
}

int nmfmu_mu_step_allreduce(const nmfmu_step* st, nmfmu_comm* comm, float* xbuf, void* stream) {
  // The column-sharded H half-step of SURVEY 8(e) as ONE host call: partial sums -> slab reduction into the packed buffer
  // [numerator M_pad x R_PAD | panel column sums R_PAD (beta == 1) or denominator M_pad x R_PAD] -> ONE RCCL all-reduce
  // -> apply -- all enqueued back to back on the caller's stream, so the host never returns to its interpreter between
  // the kernel and the collective (the torch.distributed route costs a Python / c10d round trip and two stream hops there).
  if (!st || !comm || !xbuf) return NMFMU_ERR_ARG;
  const bool kl = nmfmu_beta_kind(st->beta) == NMFMU_BETA_KL;
  const size_t plane = (size_t)st->owner.rows_pad * st->r_pad;
  const size_t tail = kl ? (size_t)st->r_pad : plane;
  if (kl && !st->panel.colsum) return NMFMU_ERR_ARG;
  int e = nmfmu_mu_partial(st, stream);
  if (e) return e;
  e = nmfmu_slab_reduce(st, xbuf, kl ? nullptr : xbuf + plane, stream);
  if (e) return e;
  if (kl) {
    hipError_t he = hipMemcpyAsync(xbuf + plane, st->panel.colsum, st->r_pad * sizeof(float), hipMemcpyDeviceToDevice, S(stream));
    if (he != hipSuccess) return (int)he;
  }
  e = nmfmu_comm_allreduce_sum_f32(comm, xbuf, plane + tail, stream);
  if (e) return e;
  return kl ? nmfmu_mu_apply(st, xbuf, nullptr, 1, xbuf + plane, stream)
            : nmfmu_mu_apply(st, xbuf, xbuf + plane, 1, nullptr, stream);
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

int nmfmu_loss_checkpoint(const nmfmu_step* st, float* loss_part, double* out2, const float* fa, float* fa_snap, int64_t na,
                          const float* fb, float* fb_snap, int64_t nb, void* stream) {
  if (!st || !loss_part || !out2 || !fa || !fa_snap || !fb || !fb_snap || na < 0 || nb < 0 || (na & 3) || (nb & 3)) return NMFMU_ERR_ARG;
  if (!nmfmu_supported(st->r_pad, st->precision)) return NMFMU_ERR_UNSUPPORTED;
  int e = fused_dispatch(st, kModeLoss, loss_part, st->owner.rows, st->panel.rows, S(stream));
  if (e) return e;
  return launch_checkpoint(loss_part, (st->owner.rows_pad / st->block_rows) * st->nsplit, st->status, out2, fa, fa_snap, na, fb,
                           fb_snap, nb, S(stream));
}

int nmfmu_riding_loss_supported(const nmfmu_step* st) {
  // the half-steps whose kernel can carry the loss term: beta == 1 on the ping-pong kernel with fp16 operands (either target
  // width); unsplit contraction (fused apply) or split (partial sums + apply kernel) alike
  if (!st || !st->xp || st->block_rows != 256 || !pp_eligible(st->r_pad, st->precision, st->beta) || !is_f16(st->precision)) return 0;
  if (st->stage == NMFMU_STAGE_DMA_SPLIT || st->nsplit < 1 || !st->slab_num) return 0;
  return st->owner.f && st->owner.colsum ? 1 : 0;
}
int nmfmu_riding_loss_part_count(const nmfmu_step* st) {
  return nmfmu_riding_loss_supported(st) ? 16 * (st->owner.rows_pad / st->block_rows) * st->nsplit : NMFMU_ERR_UNSUPPORTED;
}
int nmfmu_target_sums_nparts(void) { return 1024; }
int nmfmu_target_sums(const float* v, int64_t ld, int rows, int cols, double* part, double* out4, void* stream) {
  if (!v || !part || !out4 || rows <= 0 || cols <= 0 || ld < cols) return NMFMU_ERR_ARG;
  return launch_target_sums(v, ld, rows, cols, part, nmfmu_target_sums_nparts(), out4, S(stream));
}
int nmfmu_mu_step_with_loss(const nmfmu_step* st, const float* kl_den, float* xlogs_part, const double* target_sums,
                            double* out2, void* stream) {
  if (!st || !kl_den || !xlogs_part || !target_sums || !out2) return NMFMU_ERR_ARG;
  if (!nmfmu_riding_loss_supported(st)) return NMFMU_ERR_UNSUPPORTED;
  // (the same two launches as nmfmu_mu_step(st, kl_den, 0, stream), the first one on the kernel instance that also accumulates)
  const bool fuse = st->nsplit == 1 && st->owner.p2_hi && st->owner.colsum_part;
  int e = fuse ? fused_dispatch(st, kModeMU, xlogs_part, st->owner.rows, st->panel.rows, S(stream), kl_den, true)
               : fused_dispatch(st, kModeMU, xlogs_part, st->owner.rows, st->panel.rows, S(stream));
  if (e) return e;
  e = fuse ? launch_colsum_finalize(st->owner.colsum_part, st->owner.rows_pad / st->block_rows, st->r_pad, st->owner.colsum, S(stream))
           : nmfmu_mu_apply(st, nullptr, nullptr, 0, kl_den, stream);
  if (e) return e;
  return launch_riding_finish(xlogs_part, nmfmu_riding_loss_part_count(st) / 2, target_sums,
                              (double)st->owner.rows_pad * (double)st->panel.rows_pad, st->status, out2, S(stream));
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

int nmfmu_ubench_mfma_hbm(const void* operands, size_t operand_bytes, int f16, const void* stream_src, int kib_per_tile,
                          int waves, int tiles, int grid, float* out, void* stream) {
  if (!operands || operand_bytes < 65536 || !out || (waves != 4 && waves != 8) || tiles <= 0 || grid <= 0) return NMFMU_ERR_ARG;
  if (kib_per_tile != 0 && !stream_src) return NMFMU_ERR_ARG;
  const int rc = launch_ubench_mfma_hbm(operands, operand_bytes, f16, stream_src, kib_per_tile, waves, tiles, grid, out, S(stream));
  return rc == -2 ? NMFMU_ERR_UNSUPPORTED : rc;
}

int nmfmu_ubench_mfma_hbm2(const void* operands, size_t operand_bytes, int f16, const void* stream_src, int kib_per_tile,
                           int waves, int tiles, int wrap_tiles, int grid, float* out, uint64_t* stamps, void* stream) {
  if (!operands || operand_bytes < 65536 || !out || (waves != 4 && waves != 8) || tiles <= 0 || grid <= 0 || wrap_tiles < 0)
    return NMFMU_ERR_ARG;
  if (kib_per_tile != 0 && !stream_src) return NMFMU_ERR_ARG;
  const int rc = launch_ubench_mfma_hbm(operands, operand_bytes, f16, stream_src, kib_per_tile, waves, tiles, grid, out, S(stream),
                                        wrap_tiles, reinterpret_cast<unsigned long long*>(stamps));
  return rc == -2 ? NMFMU_ERR_UNSUPPORTED : rc;
}

int nmfmu_debug_set_buffer(void* buf) {
#ifdef NMFMU_DEBUG_HOOKS
  g_pp_debug = buf;
  return NMFMU_OK;
#else
  (void)buf;
  return NMFMU_ERR_UNSUPPORTED;   // diagnostic builds only
#endif
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
