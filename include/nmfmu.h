/* nmfmu.h -- C ABI of the MI355X-native NMF multiplicative-update engine.
 *
 * This is the drop-in boundary for the one hot path of yoyololicon/pytorch-NMF
 * (torchnmf 0.3.5) that this project accelerates: the dense beta-divergence MU
 * iteration of `torchnmf.nmf.NMF.fit` / `NMFD.fit`.  The reference has no FFI
 * of its own (it is pure Python over ATen), so each entry point below names the
 * reference code it replaces (file:line under the reference checkout).  A
 * maintainer's binding (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer unless noted;
 *  - the library never allocates device memory: the caller owns every buffer
 *    and sizes it with the nmfmu_*_bytes() queries;
 *  - every call enqueues work on `stream` (a hipStream_t passed as void*) and
 *    returns without synchronising; no hidden host<->device sync;
 *  - return value: 0 = NMFMU_OK, > 0 = a hipError_t, < 0 = NMFMU_ERR_*;
 *  - re-entrant; one host thread per device.
 *
 * Orientation.  The reference factorises V (N x C) ~ H (N x R) @ W (C x R)^T
 * (nmf.py:659-662).  A half-step updates one factor (the "owner", M rows) using
 * the other (the "panel", K rows) and X = V (H half-step) or V^T (W half-step),
 * stored in a kernel-specific tiled layout (see csrc/nmfmu_layout.h).
 */
#ifndef NMFMU_H_
#define NMFMU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMFMU_ABI_VERSION 9 /* 2: nmfmu_gemm_desc grew the implicit-operand fields; trainer / convnd / tables entries;
                               3: nmfmu_gemm_desc.tile_rows, NMFMU_EPI_FOLD, NMFMU_PREC_F16;
                               4: NMFMU_PREC_F16 for every beta and padded rank 256 (four-wave kernel); nmfmu_mu_step_parts /
                                  nmfmu_parts_supported / nmfmu_gemm_tile256_supported removed (measured neutral / not faster);
                                  NMFMU_STAGE_REG and the 256 x 256 GEMM tile no longer built; nmfmu_step.status;
                               5: nmfmu_gemm_desc.rag_c0 / rag_channels (ragged channels inside the GEMM grid), nmfmu_gemm_ragged_supported,
                                  nmfmu_conv_fold_parts_apply_h_tables / nmfmu_fold_hsum_parts_tables;
                               6: NMFMU_PREC_F16X (fp16 operands, fp32 target); nmfmu_gram_panel / nmfmu_xb_* (beta == 2 without
                                  the reconstruction); NMFMU_ERR_ALLOC;
                               7: nmfmu_gemm_desc.win_* / NMFMU_OPS_A_WIN (H numerator of NMFD / NMF2D / NMF3D without the unfolded Y),
                                  nmfmu_conv_pack_wk, nmfmu_conv_apply_h_rows; nmfmu_gemm_desc.t_koff, nmfmu_convnd_tables / _table_bytes / _koff (implicit
                                  operands with several shift axes);
                               8: nmfmu_abi_check (load-time guard for bindings that are not the bundled Python host),
                                  nmfmu_ubench_mfma_hbm (in-run ceiling of the MU step for bench.py); the NMFD GEMMs stage an implicit
                                  operand as a sliding window of table entries where its tiles hold no padding (nmfmu_gemm_desc.stage_mode,
                                  nmfmu_gemm_window_staged); nmfmu_conv_apply_pack_w_wk / nmfmu_conv_apply_h_rows_sums / nmfmu_conv_h_rows_parts;
                               9: nmfmu_kernel_family / nmfmu_choose_nsplit_for (round 6: beta == 1 at padded rank 256 with fp16 operands runs the
                                  software-pipelined one-wave-per-SIMD kernel, ONE workgroup per CU -- the split must know the kernel);
                                  nmfmu_step.stamps (in-kernel clock stamps in the product build); nmfmu_ubench_mfma_hbm2;
                                  NMFMU_PREC_F16R (3-byte target) */

#define NMFMU_OK 0
#define NMFMU_ERR_UNSUPPORTED (-2) /* rank / precision / beta combination not built */
#define NMFMU_ERR_ARG (-3)         /* inconsistent sizes or null pointer */
#define NMFMU_ERR_ALLOC (-4)       /* a host allocation inside the library failed (communicator / timer handles) */

/* precision of the MFMA operands (accumulation is always fp32, factors are kept in fp32) */
#define NMFMU_PREC_BF16 0   /* X stored bf16; operands bf16                                   */
#define NMFMU_PREC_BF16X3 1 /* X stored fp32; operands split hi+lo, 3 MFMAs per product        */
#define NMFMU_PREC_F16 2    /* X stored fp16; operands fp16 (11 significant bits, same MFMA rate as bf16); values are
                               clamped to 65504 when packed, conversions saturate.  The single-plane mode that meets the
                               reference within 1e-4 at the BASELINE shapes.  beta == 1 at padded rank <= 128 runs on the
                               ping-pong kernel (nmfmu_pp.h), everything else on the four-wave kernel (nmfmu_fused.h) */
#define NMFMU_PREC_F16X 3   /* X stored fp32 (nmfmu_xp_bytes: 4 bytes per element), operands fp16: the target is never rounded
                               (nmf.py:65 is where X enters; it stays in fp32 VALU arithmetic, for beta == 2 it is an fp16
                               hi + lo operand pair), so the mode is parity-grade on targets fp16 does not hold exactly, at 1x
                               MFMA work and twice the X stream.  Four-wave kernel, every beta, padded rank <= 256 */
#define NMFMU_PREC_F16R 4   /* (ABI 9) as F16X at THREE bytes per element of X: the target rounded (nearest even) to the top 24 bits
                               of its fp32 word -- 16 significant bits (relative error <= 2^-16), fp32's whole range -- stored as
                               the f16 layout's 16-bit words (bits 31..16) plus one byte per element (bits 15..8); the kernels
                               rebuild the fp32 with one v_perm_b32 per element.  What 'auto' takes for a target fp16 does not
                               hold exactly.  beta == 1 at padded rank <= 128: ping-pong kernel; every other beta but 2: the
                               four-wave kernel, padded rank <= 256 */

/* beta branches of nmf.py:61-74 / metrics.py:78-96 */
#define NMFMU_BETA_KL 0  /* beta == 1 */
#define NMFMU_BETA_EUC 1 /* beta == 2 */
#define NMFMU_BETA_IS 2  /* beta == 0 */
#define NMFMU_BETA_GEN 3 /* anything else */

/* how the panel tile reaches LDS */
#define NMFMU_STAGE_REG 0 /* global -> VGPR -> ds_write: no longer built, NMFMU_ERR_UNSUPPORTED */
#define NMFMU_STAGE_DMA 1 /* global_load_lds (LDS-DMA)  */
#define NMFMU_STAGE_DMA_NOP2 3 /* (ABI 9) as NMFMU_STAGE_DMA, and the caller promises that NOTHING reads the transposed images (p2_*) of
                                  this step's factors: where the step's kernel does not read the panel's p2 itself -- beta == 1 at
                                  padded rank 256 with fp16 operands (the software-pipelined kernel: ONE image + transposing LDS
                                  reads) -- the fused apply / nmfmu_mu_apply no longer refresh the owner's p2 (64 of the 384 KiB a
                                  workgroup's epilogue moves).  Everywhere else it behaves as NMFMU_STAGE_DMA */
#define NMFMU_STAGE_DMA_SPLIT 2 /* as NMFMU_STAGE_DMA, and panel.p1_* / panel.p2_* are images of DIFFERENT matrices (PLCA: the
                                   Z-scaled factor for the reconstruction, the unscaled one for the second GEMM).  Since ABI 6
                                   the single-plane four-wave kernels stage ONE panel image and gather the second GEMM's
                                   operands from it (ds_read_b64_tr_b16); a split panel must say so.  beta == 1, nmfmu_mu_partial */

/* One factor (W or H) as the engine sees it. */
typedef struct nmfmu_factor {
  float* f;           /* fp32 master, [rows][rank] row-major contiguous: the nn.Parameter storage (nmf.py:216-237) */
  void* p1_hi;        /* bf16 row-major image  [rows_pad][r_pad]          (nmfmu_image_bytes)                     */
  void* p1_lo;        /* low plane, BF16X3 only (else NULL)                                                        */
  void* p2_hi;        /* bf16 transposed tiles [rows_pad/64][r_pad][64]                                            */
  void* p2_lo;
  float* colsum;      /* [r_pad]  sum over rows of f: the beta==1 denominators of nmf.py:122-131                   */
  float* colsum_part; /* nmfmu_colsum_part_bytes(): scratch for the deterministic two-stage column sum            */
  int32_t rows;
  int32_t rows_pad;   /* nmfmu_pad_rows(rows) */
} nmfmu_factor;

/* One MU half-step: everything `_double_backward_update` (nmf.py:52-92) needs. */
typedef struct nmfmu_step {
  const void* xp;      /* X in fragment order: owner.rows_pad x panel.rows_pad (nmfmu_pack_x)                      */
  nmfmu_factor owner;  /* the factor being updated */
  nmfmu_factor panel;  /* the factor held fixed    */
  float* slab_num;     /* [nsplit][owner.rows_pad][r_pad] partial numerators                                       */
  float* slab_den;     /* same, partial denominators (beta != 1; may be NULL for beta == 1)                        */
  int32_t rank;
  int32_t r_pad;       /* nmfmu_pad_rank(rank) */
  int32_t nsplit;      /* contraction-axis split (workgroups per owner block), nmfmu_choose_nsplit                 */
  int32_t precision;   /* NMFMU_PREC_*  */
  int32_t stage;       /* NMFMU_STAGE_* */
  int32_t block_rows;  /* owner rows per workgroup tile: nmfmu_block_rows(); xp must have been packed with it        */
  float beta;
  float gamma;         /* nmf.py:341-346 */
  float l1, l2;        /* nmf.py:348-349 */
  uint32_t* status;    /* NULL or one device word (ABI 4): bit 0 is OR-ed in when an update had to clamp a factor value
                          at 65504 for its fp16 image (NMFMU_PREC_F16) -- the fit has left the mode's range           */
  void* stamps;        /* NULL, or (ABI 9) a device buffer of >= 64 + 5 * workgroups uint64: the ping-pong and the
                          software-pipelined kernel record their clocks there -- workgroup 0, wave 0 (and wave 4 of the
                          ping-pong pair): shader cycles (s_memtime), the constant 100 MHz clock (s_memrealtime) and the tile
                          count at kernel entry / loop start / loop end / exit; every workgroup the 100 MHz clock at the same four
                          points and where it ran.  Nothing is stamped inside the tile loop.  bench.py derives cycles per tile,
                          the in-kernel clock and the matrix pipe's busy fraction from it (roofline.in_kernel); layout in
                          tools/pp_timeline.py                                                                          */
} nmfmu_step;

/* ---- static queries (host only, no device work) ------------------------------------------------------------ */
int nmfmu_abi_version(void);
/* Load-time guard for bindings other than the bundled Python host: pass the NMFMU_ABI_VERSION the binding was
 * COMPILED against; NMFMU_OK iff it is the library's, NMFMU_ERR_ARG otherwise.  Struct layouts are append-only, but the
 * MEANING of a field may tighten between versions -- e.g. since ABI 6 a split panel (panel.p1_* and panel.p2_* images of
 * different matrices) must say NMFMU_STAGE_DMA_SPLIT: with NMFMU_STAGE_DMA the single-plane kernels stage p1 only and
 * derive the second GEMM's operand from it, so an ABI-5 caller's split panel would give wrong numerators without any
 * error.  A binding that calls this once after dlopen cannot run into that silently. */
int nmfmu_abi_check(int compiled_against);
int nmfmu_pad_rows(int rows);             /* rows rounded up to a multiple of 256                                  */
int nmfmu_pad_rank(int rank);             /* 32 / 64 / 128 / 256, or NMFMU_ERR_UNSUPPORTED                         */
int nmfmu_beta_kind(float beta);          /* NMFMU_BETA_*                                                          */
int nmfmu_supported(int r_pad, int precision);
int nmfmu_block_rows(int r_pad, int precision, float beta); /* 128 or 256: owner rows per workgroup tile          */
int nmfmu_choose_nsplit(int owner_rows_pad, int panel_rows_pad, int block_rows, int num_cu);
/* which fused kernel a half-step of this (padded rank, precision, beta) runs on, and the split that fills the chip with ITS
 * workgroups per CU (the ping-pong and the software-pipelined kernel: one; the four-wave kernel: two where its registers allow) */
#define NMFMU_KERNEL_FUSED 0 /* nmfmu::fused_kernel, four waves, 128-row tiles (nmfmu_fused.h)                         */
#define NMFMU_KERNEL_PP 1    /* nmfmu::pp_kernel, eight waves in two half-phases, 256-row tiles (nmfmu_pp.h)            */
#define NMFMU_KERNEL_SP 2    /* nmfmu::sp_kernel / sp2_kernel, four waves, software-pipelined across tiles, 128-row tiles
                                (nmfmu_sp.h: beta == 1 at padded rank 256; nmfmu_sp2.h: beta != 1, 2 at padded rank 128; fp16) */
int nmfmu_kernel_family(int r_pad, int precision, float beta);
int nmfmu_choose_nsplit_for(int owner_rows_pad, int panel_rows_pad, int r_pad, int precision, float beta, int block_rows, int num_cu);
/* tile height for ONE half-step of this shape (128 where the owner axis alone fills the chip, else nmfmu_block_rows) */
int nmfmu_step_block_rows(int owner_rows_pad, int panel_rows_pad, int r_pad, int precision, float beta, int num_cu);
size_t nmfmu_xp_bytes(int owner_rows_pad, int panel_rows_pad, int precision);
size_t nmfmu_image_bytes(int rows_pad, int r_pad);            /* one plane of p1 or of p2                          */
size_t nmfmu_slab_bytes(int owner_rows_pad, int r_pad, int nsplit);
size_t nmfmu_colsum_part_bytes(int rows_pad, int r_pad);

/* ---- one-time packing ----------------------------------------------------------------------------------------
 * nmfmu_pack_x: V (fp32, rows x cols, row stride ld elements) -> fragment-order X.
 *   transpose = 0: owner axis = V rows  (X = V,   H half-step and the loss)
 *   transpose = 1: owner axis = V cols  (X = V^T, W half-step)
 * Also performs the target validation of nmf.py:329-336 in the same pass:
 *   flags[0] |= 1 if any element fails (v >= 0)   (negative or NaN)
 *   flags[1]  = min over elements of the fp32 bit pattern (caller presets 0x7f800000); == 0 iff V.min() == 0
 */
int nmfmu_pack_x(const float* v, int64_t ld, int rows, int cols, int transpose, int precision, int block_rows, void* xp,
                 int owner_rows_pad, int panel_rows_pad, uint32_t* flags, void* stream);

/* nmfmu_pack_factor: build the bf16 images and the column sums from the fp32 master (after the user or
 * load_state_dict changed W/H; the apply step keeps them current afterwards). */
int nmfmu_pack_factor(const nmfmu_factor* fac, int rank, int r_pad, int precision, void* stream);

/* nmfmu_pack_factor_scaled: images of f[row][r] * scale[r] (the master stays as it is; since ABI 6 fac->colsum is NOT
 * refreshed -- only the partial sums in colsum_part are written -- PLCA's EM reads none).  PLCA feeds the first GEMM of
 * the fused kernel with the Z-scaled panel and the second with the unscaled one (scale = ones). */
int nmfmu_pack_factor_scaled(const nmfmu_factor* fac, int rank, int r_pad, int precision, const float* scale,
                             void* stream);

/* ---- the MU half-step ----------------------------------------------------------------------------------------
 * nmfmu_mu_partial: reconstruct + both backward passes of nmf.py:376-378 / 389-391, fused:
 *   slab_num[s] = sum over the s-th contraction chunk of  Gn(X, owner panel^T) @ panel      (nmf.py:77)
 *   slab_den[s] = likewise with Gp                                                    (nmf.py:82, beta != 1)
 * st->stage: NMFMU_STAGE_DMA when panel.p1_* and panel.p2_* are the two images of ONE matrix (every NMF half-step; the
 * single-plane kernels then read p1 only), NMFMU_STAGE_DMA_SPLIT when they are images of different matrices (PLCA's
 * Z-scaled / unscaled pair; beta == 1, this entry only).  Passing a split panel with NMFMU_STAGE_DMA is not detectable
 * here and yields the numerators of p1's matrix -- see nmfmu_abi_check.
 */
int nmfmu_mu_partial(const nmfmu_step* st, void* stream);

/* nmfmu_den_partial: the positive term alone for a generic beta (not 0, 1, 2), WITHOUT a target:
 *   slab_num[s] = sum over the s-th contraction chunk of (owner panel^T + eps)^(beta-1) @ panel      (st->xp may be NULL)
 * This is the dense pass the reference makes for sparse targets (nmf.py:628-636).  nmfmu_loss likewise accepts
 * st->xp == NULL and then evaluates beta_div against an all-zero target, i.e. sum (S + eps)^beta / beta. */
int nmfmu_den_partial(const nmfmu_step* st, void* stream);

/* nmfmu_mu_step: one complete single-device half-step = nmfmu_mu_partial + nmfmu_mu_apply (st->stage as there; a split
 * panel is not meaningful for a complete MU half-step and NMFMU_STAGE_DMA_SPLIT is rejected).  When the contraction is
 * not split (nsplit == 1) and beta == 1 the apply runs inside the fused kernel's epilogue (no slab round trip).
 * kl_den: column sums of the panel (beta == 1), else ignored.  phase: 0 = everything, 1 = only the fused kernel,
 * 2 = only what follows it (lets a caller bracket the dominant kernel with events). */
int nmfmu_mu_step(const nmfmu_step* st, const float* kl_den, int phase, void* stream);

/* beta == 2 WITHOUT the reconstruction (round 4).  The reference's beta == 2 branch (nmf.py:61-63) puts no eps inside the
 * two grad_outputs, so the update's numerator is X @ panel exactly and its denominator owner @ (panel^T panel): one
 * streaming MFMA GEMM over X (4 N C R flops per iteration instead of 12 N C R, HBM-bound) plus a rank x rank Gram matrix.
 *   nmfmu_gram_panel : G = panel^T panel from the panel's transposed 16-bit image (MFMA, fp32 accumulate, deterministic
 *                      two-stage sum).  Outputs: gram [r_pad][r_pad] fp32 (zero in the padding); g_hi / g_lo 16-bit
 *                      images of row r scaled by 2^-e[r] (inside fp16's range), g_scale[r] = 2^e[r].  ws: nmfmu_gram_ws_bytes.
 *   nmfmu_xb_partial : numerator slabs only (st->slab_num; st->slab_den is not touched) -- for callers that reduce across
 *                      devices before the apply.
 *   nmfmu_xb_step    : the complete half-step; phase as in nmfmu_mu_step.  The denominator is one more small MFMA product
 *                      (owner fragments x Gram image rows) inside the same kernel.  Unsplit contraction and padded rank
 *                      <= 128: everything in one kernel (nmf.py:78-92 in its epilogue); otherwise st->nsplit numerator slabs
 *                      + ONE denominator slab in st->slab_den (owner_rows_pad x r_pad floats suffice) + the apply kernel.
 * Single-plane precisions only (NMFMU_PREC_BF16 / F16 / F16X; with F16X the fp32 target enters the GEMM as an fp16
 * hi + lo pair). */
int nmfmu_xb_supported(int r_pad, int precision, float beta);
size_t nmfmu_gram_ws_bytes(int r_pad);
int nmfmu_gram_panel(const nmfmu_factor* panel, int r_pad, int precision, void* ws, float* gram, void* g_hi, void* g_lo,
                     float* g_scale, void* stream);
int nmfmu_xb_partial(const nmfmu_step* st, void* stream);
int nmfmu_xb_step(const nmfmu_step* st, const void* g_hi, const void* g_lo, const float* g_scale, int phase, void* stream);

/* Column-sum bookkeeping (nmf.py:122-131's H.sum / W.sum, deterministic two-stage sums): the number of partial sums
 * [nparts][r_pad] a half-step (nmfmu_colsum_nparts) or nmfmu_pack_factor (nmfmu_pack_nparts) leaves in colsum_part, and
 * the second stage on its own. */
int nmfmu_colsum_nparts(const nmfmu_step* st);
int nmfmu_pack_nparts(int rows_pad);
int nmfmu_colsum_finalize(const nmfmu_factor* fac, int nparts, int r_pad, void* stream);

/* nmfmu_slab_reduce: num_out = sum_s slab_num[s] (and den_out likewise unless NULL).  Used by the column-sharded
 * multi-GPU path so that one all-reduce carries [owner.rows_pad x r_pad] floats. */
int nmfmu_slab_reduce(const nmfmu_step* st, float* num_out, float* den_out, void* stream);

/* nmfmu_mu_apply: nmf.py:78-92.  neg = relu(sum of `nslab` numerator slabs) + eps;
 *   beta == 1: pos = kl_den[r]  (closed form nmf.py:122-131, no relu/eps)   else pos = relu(sum den slabs) + eps;
 *   pos += l1; pos += l2 * f;  f *= (neg / pos) ** gamma;
 * then refreshes owner's bf16 images and column sums.  num/den = NULL means "use st->slab_* with st->nsplit slabs".
 */
int nmfmu_mu_apply(const nmfmu_step* st, const float* num, const float* den, int nslab, const float* kl_den,
                   void* stream);

/* nmfmu_trainer_apply: the update of trainer.BetaMu.step (trainer.py:93-112) for one parameter, fed by the same
 * partial sums as nmfmu_mu_apply (the closure's WH.backward(output_neg) / WH.backward(output_pos) of trainer.py:93-97
 * are exactly the numerator / denominator contractions of nmfmu_mu_partial):
 *   neg = relu(sum num slabs);  pos = relu(sum den slabs)   [beta == 1: pos = kl_den[r], the ones-backward]
 *   grad = pos - neg                                          (p.grad of trainer.py:98; grad may be NULL)
 *   pos += l1;  pos += l2 * f;  pos += ortho * (sum_r f[row][r] - f);  pos += eps;  neg += eps
 *   f *= (neg / pos) ** gamma
 * l1 / l2 / gamma come from *st.  Refreshes owner's bf16 images and column sums like nmfmu_mu_apply. */
int nmfmu_trainer_apply(const nmfmu_step* st, const float* num, const float* den, int nslab, const float* kl_den,
                        float ortho, float* grad, void* stream);

/* trainer.BetaMu on a CHAIN of NMF layers (nn.Sequential, tests/test_trainer.py:10-32): the matrix products of the
 * forward / backward passes are nmfmu_reconstruct calls; these are the two pieces in between.
 *   nmfmu_mu_terms      : gn, gp = the seeds of trainer.py:75-91 from the prediction s and the target v (n elements;
 *                         beta == 1 gives gp = 1)
 *   nmfmu_trainer_update: trainer.py:93-112 on a plain row-major parameter f[rows][cols] with neg / pos of the same shape
 *                         (grad may be NULL) */
int nmfmu_mu_terms(const float* s, const float* v, int64_t n, float beta, float* gn, float* gp, void* stream);
int nmfmu_trainer_update(float* f, int rows, int cols, const float* neg, const float* pos, float l1, float l2, float ortho,
                         float gamma, float* grad, void* stream);

/* ---- loss -----------------------------------------------------------------------------------------------------
 * nmfmu_loss: beta_div(owner panel^T, X) of metrics.py:60-96 without materialising the reconstruction
 * (replaces nmf.py:360-361 and 400-401).  loss_part: nmfmu_loss_part_count() floats of scratch; *out (device
 * double) receives the divergence (NOT yet sqrt(2 x)).  rows/cols are the logical sizes of X.
 */
int nmfmu_loss_part_count(int owner_rows_pad, int block_rows, int nsplit);
int nmfmu_loss(const nmfmu_step* st, float* loss_part, double* out, void* stream);
/* (ABI 9) nmfmu_loss + everything else a loss checkpoint of the fit loop needs, in two launches: out2[0] = the same value
 * nmfmu_loss writes (same kernels, same summation order), out2[1] = 1.0 when bit 0 of st->status is set (a factor left fp16's
 * range; 0.0 without a status word), and the two factors' fp32 masters fa / fb (na / nb floats, multiples of 4, 16-byte
 * aligned) copied into fa_snap / fb_snap -- the snapshot a deferred stop decision rolls back to (nmf.py:393-407 judged one
 * checkpoint late). */
/* (ABI 9) "riding loss": the periodic KL loss of the fit loop (nmf.py:400-401) WITHOUT its own pass over the target.
 * metrics.py:22 is  target @ (log(target + eps) - log(input + eps)) - target.sum() + input.sum():
 *   nmfmu_target_sums        once per fit: out4 = { sum x ln(x + eps), sum x, max x, 1.0 if any x != fp16(x) else 0.0 } of the fp32
 *                            target in ONE pass (part: 4 * nmfmu_target_sums_nparts() doubles of scratch); the last two are what a
 *                            host needs to admit the fp16 modes (nmfmu_mu_step_with_loss reads the first two)
 *   nmfmu_mu_step_with_loss  the half-step that FOLLOWS a checkpoint (= nmfmu_mu_step(st, kl_den, 0, ..)), its kernel also
 *                            accumulating sum x log2(s) and sum s over its elements -- s = owner panel^T + eps is the reconstruction
 *                            of the factors before the update, i.e. the input the reference evaluates, from the same operand
 *                            images nmfmu_loss reads -- into xlogs_part (nmfmu_riding_loss_part_count() floats), then
 *                            out2 = { loss, fp16-range flag } (device doubles, as nmfmu_loss_checkpoint).  The caller snapshots
 *                            the factors at the checkpoint itself (they are the ones this loss belongs to)
 * nmfmu_riding_loss_supported: beta == 1 on the ping-pong kernel, fp16 operands (NMFMU_PREC_F16 / F16R), the buffers of
 * nmfmu_mu_step present.  Three VALU instructions per element on the half-steps that carry it. */
int nmfmu_riding_loss_supported(const nmfmu_step* st);
int nmfmu_riding_loss_part_count(const nmfmu_step* st);
int nmfmu_target_sums_nparts(void);
int nmfmu_target_sums(const float* v, int64_t ld, int rows, int cols, double* part, double* out4, void* stream);
int nmfmu_mu_step_with_loss(const nmfmu_step* st, const float* kl_den, float* xlogs_part, const double* target_sums,
                            double* out2, void* stream);
int nmfmu_loss_checkpoint(const nmfmu_step* st, float* loss_part, double* out2, const float* fa, float* fa_snap, int64_t na,
                          const float* fb, float* fb_snap, int64_t nb, void* stream);

/* nmfmu_beta_div: metrics.beta_div(x, y, beta) on two plain fp32 device arrays of n elements.
 * part: 1024 doubles of scratch. */
int nmfmu_beta_div(const float* x, const float* y, int64_t n, float beta, double* part, double* out, void* stream);

/* nmfmu_norms: out[0] = sum |x|, out[1] = sum x^2 over n fp32 elements (metrics.sparseness, metrics.py:99-115).
 * part: 1024 doubles of scratch. */
int nmfmu_norms(const float* x, int64_t n, double* part, double* out, void* stream);

/* nmfmu_reconstruct: out[m][k] = sum_r owner[m][r] panel[k][r]  (NMF.reconstruct, nmf.py:691-693) computed from the
 * fp32 masters with fp32 MFMA; out is row-major [owner.rows][panel.rows], ld elements per row. */
int nmfmu_reconstruct(const float* owner, int m, const float* panel, int k, int rank, float* out, int64_t ld,
                      void* stream);


/* ---- convolutive NMF (NMFD, nmf.py:700-779) --------------------------------------------------------------------
 * NMFD is dense NMF on unfolded operands with effective rank R*T: with W (C,R,T) viewed as Wm (C x R*T) and
 * Hu[(b,l)][(r,t)] = H[b][r][l-t], the reconstruction F.conv1d(H, W.flip(2), padding=T-1) (nmf.py:776-779) is
 * Wm Hu^T and both conv-backward passes (nmf.py:77, 82) are GEMMs too.  The entry points below are the pieces
 * NMFD.fit is assembled from (torchnmf_amd/nmfd_engine.py shows the order).  All matrices are zero padded to
 * multiples of 128 in every dimension; bf16 operands are (hi[, lo]) planes with the contraction index contiguous.
 */
#define NMFMU_EPI_RATIO 0 /* D = A B^T (+eps); Gn = f(D, x) [and Gp, beta != 1] stored as bf16 planes (nmf.py:61-74)   */
#define NMFMU_EPI_F32 1   /* D stored as fp32                                                                         */
#define NMFMU_EPI_LOSS 2  /* beta_div(D, x) partial per workgroup into out[(m_pad/128) * (n_pad/128)] (metrics.py)    */
#define NMFMU_EPI_FOLD 3  /* D = Y[(r,t)][(b,l)] (H numerator before the col2im sum, nmf.py:77/82 conv backward wrt the
                             input) is not stored; out receives the diagonal sums of every 128 x 128 tile
                             (nmfmu_fold_part_bytes) for nmfmu_conv_fold_parts_apply_h.  Planes only; t_batch .. t_lh
                             describe H; needs taps >= 128 and Lh + taps - 1 >= 128 (nmfmu_fold_parts_supported)       */

typedef struct nmfmu_gemm_desc {
  const void* a_hi; /* [m_pad][k_pad] bf16 */
  const void* a_lo; /* BF16X3 only */
  const void* b_hi; /* [n_pad][k_pad] bf16 */
  const void* b_lo;
  int32_t m_pad, n_pad, k_pad;
  int32_t precision; /* NMFMU_PREC_* */
  float beta;
  const float* x;   /* [m_pad][n_pad] fp32 target (RATIO, LOSS) */
  void* gn_hi;      /* RATIO outputs, [m_pad][n_pad] bf16 */
  void* gn_lo;
  void* gp_hi;      /* beta != 1 */
  void* gp_lo;
  float* out;       /* F32: [m_pad][n_pad]; LOSS: partials */
  int32_t m_valid, n_valid; /* LOSS: logical extent */
  /* Implicit conv-unfold operands (nmfmu_conv_tables): instead of materialising the Toeplitz matrix
   * Hu[(b,l)][(r,t)] = H[b][r][l-t] (T times larger than H), an operand may be fetched chunk by chunk from a
   * window table that is only 8x H.  ops selects which operand is implicit; its hi / lo pointers then address the table. */
  int32_t ops;              /* NMFMU_OPS_* */
  int32_t t_batch, t_rank, t_taps, t_lh; /* B, R, T, Lh of H (implicit operands and NMFMU_EPI_FOLD) */
  /* Workgroup tile: 0 or 128 = 128 x 128.  (ABI 3 also had a 256 x 256 tile; without a stream-K scheduler it never
   * paid at the BASELINE shapes and is no longer built: NMFMU_ERR_UNSUPPORTED.) */
  int32_t tile_rows;
  /* Leading dimension of x / gn / gp / out when the GEMM covers only the first n_pad columns of wider matrices
   * (0 = n_pad).  Together with a reduced m_pad this lets a caller leave a few ragged rows / columns to
   * nmfmu_conv_ragged_rows instead of paying a whole 128-wide tile row for them. */
  int32_t n_ld;
  /* Contraction length actually run (0 = k_pad; a multiple of 64 covering the logical extent): the zero tail of
   * 128-padded planes need not be multiplied. */
  int32_t k_len;
  /* NMFMU_EPI_F32: k_split (> 1) workgroups share the contraction of every tile; partial z goes to
   * out + z * m_pad * n_ld (the consumer adds them).  For long-k GEMMs with too few tiles to fill the chip.
   * NMFMU_EPI_FOLD: with tail_rows > 0 the LAST tail_rows tile rows are contraction-split k_split ways (the tail-round
   * split: when the tile count is N full rounds of the chip plus a few tiles, those few cost a fraction of a round
   * instead of a whole one); partial z of their diagonal sums goes to out + z * nmfmu_fold_part_bytes() / 4 and
   * nmfmu_conv_fold_parts_apply_h_tail adds them in a fixed order. */
  int32_t k_split;
  int32_t tail_rows;   /* (ABI 4, appended) */
  /* (ABI 5, appended) Ragged channels inside the grid of a reconstruction GEMM (NMFMU_EPI_RATIO with ops B_HU or A_HU):
   * channels [rag_c0, rag_channels), at most 16 -- those beyond the GEMM's own m_pad (B_HU) resp. n_pad (A_HU) rows of
   * the explicit operand, whose planes must hold rows rag_c0 .. rag_c0 + 15 (zero beyond the logical extent) -- ride
   * along as one extra 16 x 16 x 32 MFMA block per workgroup, with the same elementwise epilogue into the same planes
   * (row / column rag_c0.. of x / gn / gp at pitch n_ld).  Replaces a separate nmfmu_conv_ragged_rows launch (modes
   * 0 / 1) when nmfmu_gemm_ragged_supported().  rag_channels == 0: off. */
  int32_t rag_c0, rag_channels;
  /* (ABI 7, appended) Window operand, ops == NMFMU_OPS_A_WIN with NMFMU_EPI_F32: the H numerator of the convolutive
   * models without the unfolded Y matrix.  A is never stored: A[(b,j)][(t,c)] = P[(b, j + t)][c], where P = a_hi / a_lo
   * are the row-major ratio planes of the H half-step ([t_batch * prod(l)][win_pitch] elements, l = lh + taps - 1 per
   * shift axis) -- tile row (b, j) of k-tile (t, 64 channels) is row (b, j + t) of P, so the operand fetch is the plain
   * plane fetch with a per-lane row map and a scalar tap offset.  B = [n_pad][k_pad] planes of W ordered
   * k = (t * ceil(win_channels / 64) + ck) * 64 + c' (nmfmu_conv_pack_wk); k_len = prod(taps) * ceil(win_channels / 64) * 64;
   * m_pad >= t_batch * prod(lh) rows (b, j), j flattened over the shift axes; n_pad a multiple of 32 (the padded rank);
   * the planes P must be smaller than 2 GiB each (32-bit lane offsets).
   * out[(b,j)][r] = sum_{c,t} P[(b, j + t)][c] W[c][r][t]   (conv backward wrt H, nmf.py:776-779 / 857-860 / 937-940). */
  int32_t win_nd;          /* shift axes, 1 .. 3 */
  int32_t win_lh[3];       /* H extent per axis (outermost first; the first win_nd entries) */
  int32_t win_taps[3];     /* taps per axis */
  int32_t win_channels;    /* C */
  int32_t win_pitch;       /* elements per row of P (multiple of 64, >= ceil(C / 64) * 64) */
  /* Fold F (0 / 1: none): with a rank of at most 32 / F the 32-wide N tile would multiply mostly padding; instead F
   * consecutive taps of the LAST axis, t = F q + d, share one k position and go to F columns:
   *   out[(b, jo, j')][r F + d] = sum_{c, to, q} P[(b, jo + to, j' + F q)][c] W[c][r][to][F q + d],   j' in [0, lh_last + F - 1)
   * (1 / F of the MFMA work and operand traffic); the consumer adds num[..][j] = sum_d out[(.., j + d)][r F + d]
   * (nmfmu_conv_apply_h_rows).  taps_last % F == 0, rank * F <= n_pad, m_pad >= batch * prod(lh_outer) * (lh_last + F - 1),
   * k_len = prod(taps) / F * ceil(C / 64) * 64, B from nmfmu_conv_pack_wk with the same F. */
  int32_t win_fold;
  /* (ABI 7) Implicit operands (ops B_HU / B_HUT / A_HU) of an H with SEVERAL shift axes (NMF2D / NMF3D): win_nd > 1,
   * win_lh / win_taps as above, the operand's hi / lo = the tables of nmfmu_convnd_tables, and t_koff = DEVICE copy of
   * nmfmu_convnd_koff(ops, ..., k_pad) -- one int per 8-wide k-chunk.  Needs taps and V extent of the LAST axis to be
   * multiples of 8.  NULL with win_nd <= 1: the one-axis form above (t_taps, t_lh). */
  const int32_t* t_koff;
  /* (ABI 8, appended) How an implicit operand (ops B_HU / B_HUT / A_HU, one shift axis) reaches LDS: 0 = automatic -- as a
   * sliding WINDOW of the distinct table entries a k-tile touches (128 + 56 entries = 2.9 KiB instead of the 16 KiB of a
   * chunk-major tile, which holds most entries eight times: 19 instead of 32 LDS-DMA pieces per k-tile) wherever the
   * implicit operand's tiles hold no padding (nmfmu_gemm_window_staged), chunk-major elsewhere; 1 = chunk-major always
   * (the form of ABI <= 7; same results bit for bit -- the products and their order are unchanged). */
  int32_t stage_mode;
} nmfmu_gemm_desc;

#define NMFMU_OPS_PLANES 0   /* A and B are bf16 planes                                                            */
#define NMFMU_OPS_B_HU 1     /* B = Hu : rows (b,l), k = (r,t)  -- reversed-window table  (W half-step reconstruction) */
#define NMFMU_OPS_B_HUT 2    /* B = HuT: rows (r,t), k = (b,l)  -- forward-window table   (W numerator)              */
#define NMFMU_OPS_A_HU 3     /* A = Hu                                                     (H half-step reconstruction) */
#define NMFMU_OPS_A_WIN 4    /* A = shifted rows of the ratio planes, B = W as [r][(t,c)]  (H numerator; see win_* below)  */

/* nmfmu_conv_tables: the two window tables of H (B, R, Lh) for T taps, bf16 (hi[, lo]):
 *   rev[1 + (b R + r) JJ + jj] = { H[b][r][j], H[b][r][j-1], ..., H[b][r][j-7] }     j = jj - (T-1)
 *   fwd[1 + (b R + r) JJ + jj] = { H[b][r][j], H[b][r][j+1], ..., H[b][r][j+7] }
 * with JJ = Lh + 2T - 2 (= L + T - 1), entries outside [0, Lh) zero, chunk 0 all zero.  Requires T % 8 == 0 and
 * (Lh + T - 1) % 8 == 0; nmfmu_conv_table_bytes() gives the size of ONE plane of ONE table. */
size_t nmfmu_conv_table_bytes(int batch, int rank, int lh, int taps);
int nmfmu_conv_tables(const float* h, int batch, int rank, int lh, int taps, void* rev_hi, void* rev_lo, void* fwd_hi,
                      void* fwd_lo, void* stream);

/* Window tables for several shift axes, H (batch, rank, lh[0..ndim-1]), taps[0..ndim-1] (outermost axis first).  Let
 * jj_d = lh_d + 2 taps_d - 2 and number the zero-padded lines of H (b, r, p_0, .., p_{n-2}), p_d in [0, jj_d), line
 * (.., p_d, ..) = H[b][r][p_0 - (taps_0 - 1)] .. (all zero when an index falls outside H).  The tables are the one-axis
 * tables of these lines along the last axis, concatenated:
 *   rev[1 + (((b R + r) jj_0 + p_0) .. ) jj_last + p] = { line[j], line[j-1], .., line[j-7] },  j = p - (taps_last - 1)
 *   fwd[...]                                          = { line[j], line[j+1], .., line[j+7] }
 * chunk 0 all zero.  The chunk a GEMM lane fetches is (term of its tile row) + koff[k-chunk]:
 *   rows (b, l), k = (r, t)  (B_HU, A_HU):  row term = b R JJ + sum_d (l_d + taps_d - 1) S_d,
 *                                           koff[kc] = 1 + r JJ - sum_d t_d S_d              (t = the chunk's first tap)
 *   rows (r, t), k = (b, l)  (B_HUT):       row term = r JJ + sum_d (taps_d - 1 - t_d) S_d,
 *                                           koff[kc] = 1 + b R JJ + sum_d l_d S_d
 * with S_last = 1, S_d = S_{d+1} jj_{d+1}, JJ = prod jj_d; koff = INT32_MIN for k-chunks in the padding of k.
 * nmfmu_convnd_koff is host code and fills k_pad / 8 + 8 ints (the last 8 are spare: the kernel reads one k-tile ahead);
 * the caller copies the array to the device. */
size_t nmfmu_convnd_table_bytes(int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps);
int nmfmu_convnd_tables(const float* h, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps, int precision,
                        void* rev_hi, void* rev_lo, void* fwd_hi, void* fwd_lo, void* stream);   /* BF16 | BF16X3 (lo planes) | F16 */
int nmfmu_convnd_koff(int ops, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps, int k_pad,
                      int32_t* koff_host);

int nmfmu_gemm(const nmfmu_gemm_desc* d, int epilogue, void* stream);
/* 1 when nmfmu_gemm(d, epilogue) stages its implicit operand as a window of table entries (stage_mode 0 and a shape whose
 * implicit operand has no padding inside any tile: rows (b,l): (Lh + T - 1) % 128 == 0, T >= 64, n_pad resp. m_pad ==
 * batch * (Lh + T - 1), k_len == rank * T (a multiple of 64); rows (r,t): n_pad == rank * T (a multiple of 128), T >= 128,
 * (Lh + T - 1) % 64 == 0, k_len == batch * (Lh + T - 1)); 0 otherwise; < 0: the descriptor is invalid. */
int nmfmu_gemm_window_staged(const nmfmu_gemm_desc* d, int epilogue);
/* NMFMU_PREC_F16 in the GEMM engine: fp16 operand planes / window tables (nmfmu_conv_tables_f16,
 * nmfmu_conv_apply_pack_w_sums with precision F16) and fp16 ratio planes, one plane each, same MFMA rate as bf16 with
 * 11 significant bits; ratios saturate at 65504.  Built for the combinations of the beta == 1 NMFD iteration on implicit
 * operands: RATIO with B or A = Hu, LOSS with B = Hu, F32 with B = HuT, FOLD (else NMFMU_ERR_UNSUPPORTED). */
int nmfmu_gemm_f16_supported(float beta, int epilogue, int ops);
int nmfmu_conv_tables_f16(const float* h, int batch, int rank, int lh, int taps, void* rev, void* fwd, void* stream);

/* Ragged channels of the NMFD reconstruction (RATIO / LOSS epilogues with A or B = Wm): channels c0 .. channels-1 by
 * direct summation S[c][(b,l)] = sum_{r,t} w[c][r][t] h[b][r][l-t] from the fp32 masters, so that the GEMM only has to
 * cover the first c0 = floor(channels / 128) * 128 of them (m_pad resp. n_pad = c0, n_ld = the planes' width): a
 * spectrogram with 2^k + 1 bins otherwise pays a whole 128-row tile row for one channel.
 *   mode 0: ratio planes gn (gp) [c][ld]   (W half-step)      mode 1: [(b,l)][ld]   (H half-step)
 *   mode 2: beta_div partials, loss_part[nmfmu_conv_ragged_blocks() * (channels - c0)]
 * x has the layout of the outputs.  BF16 / BF16X3 / F16. */
int nmfmu_conv_ragged_supported(int rank, int taps);
int nmfmu_conv_ragged_blocks(int batch, int lh, int taps);
/* 1 if nmfmu_gemm can take `extra` (1..16) ragged channels into its own grid (nmfmu_gemm_desc.rag_*): eight workgroups
 * share out the 128 frames of an implicit-operand tile, so the explicit operand needs >= 8 whole 128-row tiles. */
int nmfmu_gemm_ragged_supported(int ops, int m_pad, int n_pad, int extra);
int nmfmu_conv_ragged_rows(const float* w, int channels, int rank, int taps, const float* h, int batch, int lh, int c0,
                           int precision, float beta, int mode, const float* x, int64_t ld, void* gn_hi, void* gn_lo,
                           void* gp_hi, void* gp_lo, float* loss_part, void* stream);

/* Strided 2-D gather of an fp32 tensor into a zero-padded row-major matrix (fp32 copy and/or bf16 hi[,lo] planes):
 *   dst[row][col] = src[(row / row_inner) * row_outer_stride + (row % row_inner) * row_inner_stride
 *                     + (col / col_inner) * col_outer_stride + (col % col_inner) * col_inner_stride]
 * Covers V (B,C,L) -> [c][(b,l)] and [(b,l)][c], and W (C,R*T) -> Wm / Wm^T.  Optional validation flags as in
 * nmfmu_pack_x (nmf.py:329-336). */
int nmfmu_pack2d(const float* src, int rows, int cols, int row_inner, int64_t row_outer_stride, int64_t row_inner_stride,
                 int col_inner, int64_t col_outer_stride, int64_t col_inner_stride, int rows_pad, int cols_pad,
                 float* dst_f32, void* dst_hi, void* dst_lo, uint32_t* flags, void* stream);

/* Toeplitz unfold of H (batch, rank, lh): hu [(b,l)][(r,t)] (bl_pad x rp_pad) and hut [(r,t)][(b,l)]. */
int nmfmu_conv_unfold(const float* h, int batch, int rank, int lh, int taps, void* hu_hi, void* hu_lo, void* hut_hi,
                      void* hut_lo, int bl_pad, int rp_pad, void* stream);

/* out[r] = sum over outer o and inner i of src[o][r][i]: the closed-form beta == 1 denominators (nmf.py:122-131).
 * part: rank * 128 floats of scratch (deterministic two-stage sum). */
int nmfmu_rank_sums(const float* src, int outer, int rank, int inner, float* part, float* out, void* stream);

/* nmf.py:78-92 for W (channels, rank, taps) in place; num/den fp32 [c_pad][rp_pad]; kl_den[rank] or den.
 * (This entry and nmfmu_conv_apply_pack_w below are the un-fused forms: kept for bindings written against earlier ABI
 * versions; the Python host calls nmfmu_conv_apply_pack_w_sums, whose optional operands make it a superset.) */
int nmfmu_conv_apply_w(float* w, int channels, int rank, int taps, const float* num, const float* den,
                       const float* kl_den, int rp_pad, float l1, float l2, float gamma, void* stream);

/* nmfmu_conv_apply_w (when update != 0) fused with the re-packing of W into both GEMM operand layouts:
 * Wm [c_pad][rp_pad] and WmT [rp_pad][c_pad] bf16 planes (hi[, lo]; padding zero).  update == 0 only packs.
 * PRECONDITION of update != 0: the same plane buffers (and tile-sum buffer, in the _sums form) have been through ONE
 * update == 0 call before -- channel tiles that are padding only are skipped by the update pass and rely on the zeros
 * the pack-only pass wrote there. */
int nmfmu_conv_apply_pack_w(float* w, int channels, int rank, int taps, const float* num, const float* den,
                            const float* kl_den, int c_pad, int rp_pad, float l1, float l2, float gamma, int update,
                            void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, void* stream);

/* Folds y[(r,t)][(b,l)] (fp32 [rp_pad][bl_pad]) along the taps, neg[b][r][j] = sum_t y[(r,t)][(b,j+t)], then
 * nmf.py:78-92 for H (batch, rank, lh) in place. */
int nmfmu_conv_fold_apply_h(float* h, int batch, int rank, int lh, int taps, const float* y_num, const float* y_den,
                            const float* kl_den, int bl_pad, float l1, float l2, float gamma, void* stream);
/* The same update from the per-tile diagonal sums of NMFMU_EPI_FOLD GEMMs (p_num / p_den) instead of Y: saves writing
 * and re-reading 4 * R*T * B*L bytes per GEMM.  Deterministic (fixed gather order). */
/* The same two kernels with the beta == 1 denominators (nmf.py:122-131) fused in instead of nmfmu_rank_sums launches:
 *   nmfmu_conv_apply_pack_w_sums     takes num as num_slabs split-K partials [slab][c_pad][rp_pad] (nmfmu_gemm_desc.k_split),
 *                                    sum_{b,j} H[b][r][j] as kl_den (finished) or as kl_hpart[rank][n_hparts]
 *                                    partials, and leaves wcol[c_pad/64][rp_pad/64][2] = sums of W per 64 x 64 tile and
 *                                    rank (taps >= 64: at most two ranks per tile);
 *   nmfmu_conv_fold_parts_apply_h_sums takes sum_{c,t} W[c][r][t] as kl_den or as kl_wcol (c_tiles = c_pad / 64) and
 *                                    leaves hsum_part[rank][nmfmu_fold_hsum_parts(batch, lh)] partial sums of the new H. */
int nmfmu_conv_apply_pack_w_sums(float* w, int channels, int rank, int taps, const float* num, const float* den,
                                 const float* kl_den, const float* kl_hpart, int n_hparts, float* wcol, int num_slabs,
                                 int c_pad, int rp_pad, float l1, float l2, float gamma, int update, int precision,
                                 void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, void* stream);
int nmfmu_fold_hsum_parts(int batch, int lh);
int nmfmu_conv_fold_parts_apply_h_sums(float* h, int batch, int rank, int lh, int taps, const float* p_num,
                                       const float* p_den, const float* kl_den, const float* kl_wcol, int c_tiles,
                                       int rp_pad, float* hsum_part, int bl_pad, float l1, float l2, float gamma,
                                       void* stream);
/* ... and for an EPI_FOLD GEMM that ran with the tail-round split (nmfmu_gemm_desc.tail_rows / k_split; m_pad = the GEMM's):
 * the superset of the two entries above (kl_wcol / hsum_part may be NULL). */
int nmfmu_conv_fold_parts_apply_h_tail(float* h, int batch, int rank, int lh, int taps, const float* p_num,
                                       const float* p_den, const float* kl_den, const float* kl_wcol, int c_tiles,
                                       int rp_pad, float* hsum_part, int bl_pad, float l1, float l2, float gamma, int m_pad,
                                       int tail_rows, int k_split, void* stream);
/* ... and rewriting the window tables of the new H in the same launch (ABI 5; replaces the nmfmu_conv_tables call that
 * would follow).  A table entry spans eight consecutive j, so blocks recompute a halo of their neighbours' elements and
 * need the OLD values while the neighbours overwrite theirs: h_old is a copy of h that this launch only reads, h_next a
 * second buffer that receives the new values beside h itself; the caller swaps the two every iteration (and refreshes
 * h_old from h whenever h was changed by anything else).  The tables must have been built by nmfmu_conv_tables[_f16]
 * once: entries whose windows lie outside [0, lh) are never touched.  hsum_part holds
 * nmfmu_fold_hsum_parts_tables(batch, lh) partial sums per rank.  rev_lo / fwd_lo: NMFMU_PREC_BF16X3 only. */
int nmfmu_fold_hsum_parts_tables(int batch, int lh);
int nmfmu_conv_fold_parts_apply_h_tables(float* h, const float* h_old, float* h_next, int batch, int rank, int lh, int taps,
                                         const float* p_num, const float* p_den, const float* kl_den, const float* kl_wcol,
                                         int c_tiles, int rp_pad, float* hsum_part, int bl_pad, float l1, float l2,
                                         float gamma, int m_pad, int tail_rows, int k_split, int precision, void* rev_hi,
                                         void* rev_lo, void* fwd_hi, void* fwd_lo, void* stream);
size_t nmfmu_fold_part_bytes(int m_pad, int n_pad);
int nmfmu_fold_parts_supported(int batch, int rank, int lh, int taps);
int nmfmu_conv_fold_parts_apply_h(float* h, int batch, int rank, int lh, int taps, const float* p_num, const float* p_den,
                                  const float* kl_den, int bl_pad, float l1, float l2, float gamma, void* stream);

/* NMF2D / NMF3D (nmf.py:782-942): the same two steps with ndim = 2 or 3 shift axes (lh[ndim], taps[ndim] outermost
 * first; ndim = 1 is NMFD).  Flattened, (b,l) has batch * prod(lh + taps - 1) rows and (r,t) rank * prod(taps) columns;
 * everything else of the NMFD sequence (pack2d, gemm, rank_sums, conv_apply_w with taps = prod(taps)) is shared. */
int nmfmu_convnd_unfold(const float* h, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps,
                        void* hu_hi, void* hu_lo, void* hut_hi, void* hut_lo, int bl_pad, int rp_pad, void* stream);
int nmfmu_convnd_fold_apply_h(float* h, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps,
                              const float* y_num, const float* y_den, const float* kl_den, int bl_pad, float l1,
                              float l2, float gamma, void* stream);

/* ---- sparse-COO targets (nmf.py:351-398, 602-638), beta in {1, 2} ---------------------------------------------
 * V is handed over as CSR over the owner axis of the half-step: rowptr[owner_rows + 1], colidx / vals[nnz] (int32
 * indices; rows of V for the H half-step, rows of V^T for the W half-step).  Factors are the plain fp32 masters.
 *   nmfmu_sp_partial : num[row][:] = sum over the row's entries of g(v, <owner[row], panel[col]>) * panel[col][:]
 *                      (g = v / (s + eps) for beta 1, v for beta 2), num is [owner_rows][r_pad], later passed to
 *                      nmfmu_mu_apply as a single slab together with kl_den (beta 1) or den (beta 2)
 *   nmfmu_gram       : gram = f^T f (rank x rank);  nmfmu_rowmat : den[row][:] = owner[row] @ gram  (beta 2)
 *   nmfmu_sp_loss_neg: *out = sum_nnz v log(s + eps) (beta 1) or v s (beta 2), the data-dependent term of the loss
 *                      the reference tracks on sparse targets; part: (owner_rows + 3) / 4 doubles of scratch
 * Other beta: NMFMU_ERR_UNSUPPORTED (their positive term is a dense N x C pass in the reference too). */
int nmfmu_sp_partial(const int32_t* rowptr, const int32_t* colidx, const float* vals, int owner_rows, const float* owner,
                     const float* panel, int rank, float beta, float* num, int r_pad, void* stream);
int nmfmu_sp_loss_neg(const int32_t* rowptr, const int32_t* colidx, const float* vals, int owner_rows, const float* owner,
                      const float* panel, int rank, float beta, double* part, double* out, void* stream);
size_t nmfmu_gram_part_bytes(int rank);   /* scratch of nmfmu_gram */
int nmfmu_gram(const float* f, int rows, int rank, float* part, float* gram, void* stream);
int nmfmu_rowmat(const float* owner, int rows, int rank, const float* gram, float* den, int r_pad, void* stream);

/* ---- PLCA's EM update (plca.py:248-290) -----------------------------------------------------------------------------
 * With G = Vn / (H diag(Z) W^T + eps) the factor "gradients" are (G^T H) * Z, (G W) * Z and Z.grad[r] = sum W * (G^T H);
 * nmfmu_mu_partial delivers the unscaled numerators G^T H / G W when the step's panel struct carries the image of the
 * Z-scaled factor as p1 (reconstruction) and of the unscaled factor as p2.  part: nmfmu_plca_part_bytes() of scratch.
 *   nmfmu_plca_em        f *= relu(num * z_old) (skipped when update == 0); colsum_out = column sums of the result;
 *                        zgrad_out (may be NULL) = sum_rows f_old * num
 *   nmfmu_plca_normalize f /= divider[r]; when alpha != 1: f += alpha - 1, clamped below at eps; colsum_out = column sums
 *   nmfmu_plca_scale     f /= colsum[r] */
size_t nmfmu_plca_part_bytes(int rows, int r_pad);
int nmfmu_plca_em(float* f, int rows, int rank, int r_pad, const float* num, int nslab, int rows_pad, const float* z_old,
                  int update, float* part, float* colsum_out, float* zgrad_out, void* stream);
int nmfmu_plca_normalize(float* f, int rows, int rank, int r_pad, const float* divider, float alpha, float* part,
                         float* colsum_out, void* stream);
int nmfmu_plca_scale(float* f, int rows, int rank, const float* colsum, void* stream);
/* plca.py:253-260 in one launch: prior[r] = z[r] * relu(zgrad[r]); z <- prior (+ alpha - 1, clamped below at eps when
 * alpha != 1), then z /= sum(z).  rank <= 256. */
int nmfmu_plca_z(float* z, const float* zgrad, int rank, float alpha, float* prior, void* stream);

/* ---- shift-invariant PLCA (SIPLCA / SIPLCA2 / SIPLCA3, plca.py:376-606) on the NMFD GEMM path ----------------------
 * Same EM update as PLCA with W (C, R, *T), H (B, R, *Lh): factors are addressed [outer][rank][inner], the unscaled
 * numerator as num[o * num_pitch + r * inner + i].
 *   nmfmu_conv_pack_w_scaled : Wm / WmT planes of W * scale[r] (the reconstruction operand W * Z); W is not modified
 *   nmfmu_convnd_fold        : out[b][r][j] = sum_t y[(r,t)][(b, j + t)]  (the fold of nmfmu_convnd_fold_apply_h alone)
 *   nmfmu_plca3              : mode 0 em (vec = z_old), 1 normalize (vec = divider, alpha), 2 scale (vec = colsum);
 *                              part: nmfmu_plca3_part_bytes(rank) of scratch */
int nmfmu_conv_pack_w_scaled(float* w, int channels, int rank, int taps, const float* scale, int c_pad, int rp_pad,
                             void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, void* stream);
int nmfmu_convnd_fold(float* out, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps, const float* y,
                      int bl_pad, void* stream);
/* The two small kernels around the window-operand GEMM (NMFMU_OPS_A_WIN), F = nmfmu_gemm_desc.win_fold (>= 1):
 *   nmfmu_conv_pack_wk:      Wk[r F + d][((to TQ + q) CK + ck) 64 + c'] = w[c = 64 ck + c'][r][to taps_last + F q + d]
 *                            (CK = ceil(channels / 64), TQ = taps_last / F; taps = the product over the shift axes, innermost
 *                            last), planes [rows_pad][k_pad] zero padded
 *   nmfmu_conv_apply_h_rows: h (batch, rank, lh_outer, lh_last) in place by nmf.py:78-92 from num / den
 *                            [(b, jo, j')][ld], j' < lh_last + F - 1 (den NULL: beta == 1, kl_den[r] = sum_{c,t} W[c][r][t]) */
int nmfmu_conv_pack_wk(const float* w, int channels, int rank, int taps, int taps_last, int fold, int rows_pad, int k_pad,
                       int precision, void* wk_hi, void* wk_lo, void* stream);
int nmfmu_conv_apply_h_rows(float* h, int batch, int rank, int lh_outer, int lh_last, int fold, const float* num,
                            const float* den, const float* kl_den, int ld, float l1, float l2, float gamma, void* stream);
/* (ABI 8) The same two with their neighbours riding along -- four launches less per iteration of NMF2D / NMF3D / NMFD with a
 * short kernel (beta == 1, taps >= 64 in total, rank <= 256):
 *   nmfmu_conv_apply_pack_w_wk    = nmfmu_conv_apply_pack_w_sums (update, Wm / WmT planes, kl_hpart in, wcol out) + the Wk planes
 *                                   of nmfmu_conv_pack_wk from the same tile (their zero padding is NOT rewritten: run
 *                                   nmfmu_conv_pack_wk once on these buffers first);
 *   nmfmu_conv_apply_h_rows_sums  = nmfmu_conv_apply_h_rows for beta == 1 with sum_{c,t} W[c][r][t] finished in the kernel from
 *                                   the tile sums kl_wcol [c_tiles][rp_pad / 64][2] (nmfmu_conv_apply_pack_w_sums / _wk), and
 *                                   sum_{b,j} of the NEW h left as hsum_part[r][nmfmu_conv_h_rows_parts()] partials (fixed order)
 *                                   -- what kl_hpart / n_hparts of the next W update take. */
int nmfmu_conv_apply_pack_w_wk(float* w, int channels, int rank, int taps, const float* num, const float* den,
                               const float* kl_den, const float* kl_hpart, int n_hparts, float* wcol, int num_slabs,
                               int c_pad, int rp_pad, float l1, float l2, float gamma, int update, int precision,
                               void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, int taps_last, int fold, int wk_rows_pad,
                               int wk_k_pad, void* wk_hi, void* wk_lo, void* stream);
int nmfmu_conv_h_rows_parts(int batch, int rank, int lh_outer, int lh_last);
int nmfmu_conv_apply_h_rows_sums(float* h, int batch, int rank, int lh_outer, int lh_last, int fold, const float* num,
                                 const float* kl_wcol, int c_tiles, int rp_pad, int taps, int ld, float l1, float l2, float gamma,
                                 float* hsum_part, void* stream);
/* ... and the sum alone: out (batch, rank, lh_outer, lh_last) = the folded numerator (shift-invariant PLCA's own update) */
int nmfmu_conv_rows_fold(float* out, int batch, int rank, int lh_outer, int lh_last, int fold, const float* num, int ld,
                         void* stream);
/* slabs[0][i] += slabs[1][i] + .. + slabs[nslab-1][i], fixed order: the partials of a split-K NMFMU_EPI_F32 launch
 * (k_split slabs of m_pad * n_ld floats) for a consumer that takes one slab.  slab_elems a multiple of 4. */
int nmfmu_slab_sum(float* slabs, int64_t slab_elems, int nslab, void* stream);
size_t nmfmu_plca3_part_bytes(int rank);
int nmfmu_plca3(int mode, float* f, int outer, int rank, int inner, const float* num, int64_t num_pitch, const float* vec,
                float alpha, int update, float* part, float* colsum_out, float* zgrad_out, void* stream);

/* ---- the collective of the column-sharded path ----------------------------------------------
 * One process (or host thread) per GPU, or one process driving several: the H half-step sums ONE packed fp32 buffer
 * [numerators (owner.rows_pad x r_pad) | denominators] over the ranks between nmfmu_mu_partial + nmfmu_slab_reduce and
 * nmfmu_mu_apply(..., nslab = 1); relu / eps / regularisers are applied after the sum, as on the unsharded matrix
 * (nmf.py:78-92).  RCCL is resolved at run time (dlopen): NMFMU_ERR_UNSUPPORTED when it is not installed.  RCCL errors
 * come back as 10000 + ncclResult_t.  (The Python host side of this repository uses torch.distributed, backend "nccl" =
 * RCCL, for the same exchange.) */
typedef struct nmfmu_comm nmfmu_comm;
int nmfmu_comm_available(void);                                   /* 1 when librccl could be loaded                      */
int nmfmu_comm_unique_id(void* id128);                            /* rank 0: 128 bytes to hand to every rank out of band */
int nmfmu_comm_init_rank(nmfmu_comm** comm, int nranks, const void* id128, int rank); /* current device = this rank's GPU */
int nmfmu_comm_init_all(nmfmu_comm** comms, int ndev, const int* devices);  /* one process, ndev GPUs (devices may be NULL) */
int nmfmu_comm_nranks(const nmfmu_comm* comm);
int nmfmu_comm_allreduce_sum_f32(nmfmu_comm* comm, float* buf, size_t count, void* stream);  /* in place, asynchronous */
/* one host thread driving ndev devices: the ndev all-reduces of one exchange inside ncclGroupStart / End */
int nmfmu_comm_allreduce_sum_f32_multi(nmfmu_comm* const* comms, float* const* bufs, size_t count, void* const* streams,
                                       int ndev);
int nmfmu_comm_destroy(nmfmu_comm* comm);
/* The column-sharded H half-step in one host call: nmfmu_mu_partial -> nmfmu_slab_reduce into xbuf =
 * [numerator owner_rows_pad x r_pad | beta == 1: the panel shard's column sums, r_pad floats; else the denominator,
 * owner_rows_pad x r_pad] -> ONE all-reduce of the whole buffer -> nmfmu_mu_apply(nslab = 1) -- enqueued back to back on
 * `stream`; relu / eps / regularisers are applied after the sum, as nmf.py:78-92 does on the unsharded matrix. */
int nmfmu_mu_step_allreduce(const nmfmu_step* st, nmfmu_comm* comm, float* xbuf, void* stream);

/* ---- instrumentation ------------------------------------------------------------------------------------------
 * hipEvent-based timers on the caller's stream (bench.py uses them to time the dominant kernel live). */
int nmfmu_timer_create(int n_events, void** timer);
int nmfmu_timer_record(void* timer, int idx, void* stream);
int nmfmu_timer_elapsed_ms(void* timer, int idx_from, int idx_to, float* ms); /* synchronises on idx_to */
int nmfmu_timer_destroy(void* timer);

/* ---- self-test probes (used by tests only; they validate the hardware assumptions the kernels rest on) ------ */
int nmfmu_probe_mfma(const uint16_t* a /*32x16 bf16 row-major*/, const uint16_t* b /*16x32*/, float* d /*32x32*/,
                     void* stream);
int nmfmu_probe_lds_dma(const uint32_t* src, uint32_t* dst, int n_dwords /* multiple of 1024 */, void* stream);
/* The zero-overhead ceiling of the fused MU step, measured in the run that quotes it (bench.py: roofline.ceiling_tflops):
 * every wave issues 32 MFMAs (32x32x16; f16 != 0: fp16, else bf16) per tile on fixed fragments read once from `operands`
 * (a REAL 16-bit factor image of >= 64 KiB, so that the matrix pipe sees the MU step's value distribution -- the clock the
 * power limit leaves follows the data) and streams kib_per_tile KiB (0, 1, 2, 4 or 8) per wave and tile from `stream_src`
 * by non-temporal LDS-DMA, never read back, at most three tiles in flight; no LDS reads, no VALU, no barriers.
 * kib_per_tile = 4 is the rank-128 MU step (256 flop per X byte).  grid workgroups of `waves` (4 or 8) waves, `tiles`
 * tiles each: stream_src holds grid * waves * tiles * kib_per_tile KiB, out grid * waves * 64 floats (a sink).  Flops =
 * grid * waves * tiles * 32 * 32768.  Asynchronous on `stream`; time it with events.  Not used by the product path. */
int nmfmu_ubench_mfma_hbm(const void* operands, size_t operand_bytes, int f16, const void* stream_src, int kib_per_tile,
                          int waves, int tiles, int grid, float* out, void* stream);
/* The same loop with its own clocks (ABI 9): `tiles` may exceed what stream_src holds -- the stream wraps after wrap_tiles
 * tiles per wave (0 = never) -- and every wave writes {shader cycles, 100 MHz ticks, start tick, end tick} of its tile loop
 * to stamps[4 * (workgroup * waves + wave) ..] (NULL = no stamps; grid * waves * 4 uint64).  The two waves of a SIMD do not
 * share the matrix pipe evenly, so the busy fraction is taken over a workgroup's span: 32 cycles x MFMAs per SIMD /
 * ((max end - min start) x clock), clock = cycles / ticks of any wave. */
int nmfmu_ubench_mfma_hbm2(const void* operands, size_t operand_bytes, int f16, const void* stream_src, int kib_per_tile,
                           int waves, int tiles, int wrap_tiles, int grid, float* out, uint64_t* stamps, void* stream);
/* Diagnostic hook of the ping-pong kernel (nmfmu_pp.h), live only in libraries built with -DNMFMU_DEBUG_HOOKS
 * (NMFMU_ERR_UNSUPPORTED otherwise): with a device buffer of >= (64 + 5 * workgroups) uint64 registered, every
 * workgroup records clock stamps at kernel entry, loop start, loop end and exit (tools/pp_timeline.py).  buf = NULL
 * unregisters.  Not used by the product path. */
int nmfmu_debug_set_buffer(void* buf);

#ifdef __cplusplus
}
#endif
#endif /* NMFMU_H_ */
