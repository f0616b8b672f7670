// Fused beta-divergence MU half-step for gfx950 (MI355X, CDNA4) -- the four-wave kernel.
//
// One launch computes, for a block of 128 owner rows and a chunk of the
// contraction axis, the MU numerator (and, for beta != 1, denominator)
//
//     S[m][k]   = sum_r A[m][r] * B[k][r]               (MFMA, never leaves registers)
//     Gn, Gp    = elementwise(S, X[m][k])                (nmf.py:61-74 of the reference)
//     num[m][r] = sum_k Gn[m][k] * B[k][r]               (MFMA, operand = the registers above)
//     den[m][r] = sum_k Gp[m][k] * B[k][r]               (beta != 1 only)
//
// i.e. reconstruct + both autograd backward passes of nmf.py:376-378 /
// 389-391 in one pass over X.  It is the "attention without softmax" shape:
// the S^T tile is produced by v_mfma_f32_32x32x16 with the panel as the
// A operand, so that each lane ends up holding one owner row and 16
// consecutive contraction columns -- exactly the A-operand layout of the
// second MFMA (contraction over k).  The trick that makes the columns
// consecutive is a row permutation pi of the panel tile (see a_row[] below);
// with it the X tile can be stored in HBM in "fragment order" (nmfmu_layout.h)
// and loaded straight into the right lanes with fully coalesced 16-byte loads.
//
// This kernel serves beta != 1 (two accumulator sets), the split-bf16 mode and padded rank 256; the beta == 1
// half-steps with one operand plane at padded rank <= 128 run on the eight-wave ping-pong kernel (nmfmu_pp.h).
//
// Precision modes (template parameter PREC = NMFMU_PREC_*)
//   bf16   : A, B and Gn/Gp rounded to bf16, X stored bf16, fp32 accumulate.
//   bf16x3 : every bf16 operand is a (hi, lo) pair and every product is
//            hi*hi + lo*hi + hi*lo (3 MFMAs); X stored fp32.  ~2^-16 relative
//            operand error: fp32-grade.
//   f16    : fp16 operands and X (11 significant bits at bf16's MFMA rate) -- the single-plane mode that meets the
//            1e-4 parity bar at the BASELINE shapes.  MODE.FP16_OVFL is set, so conversions
//            saturate at 65504.  For beta < 1 the elementwise terms are negative powers of S and can sit at the
//            bottom of fp16's range, so Gn and Gp are multiplied by a power of two derived from the factors' column
//            sums (identical in every workgroup) before the conversion; the epilogue scales the accumulators back.
//   f16r   : (round 6) as f16x at THREE bytes per element of X: the fp32 rounded (nearest even) to its top 24 bits -- 16
//            significant bits, fp32's range.  Bits 31..16 sit where the f16 layout has its 16-bit words, bits 15..8 in two
//            more chunks per lane and tile; ONE v_perm_b32 per element rebuilds the fp32, the ratio stage then runs as for an
//            fp32 target.  beta != 2 (there the target is an MFMA operand itself: 'f16x' keeps its hi + lo pair).  (A first
//            form -- fp16 head + 8-bit relative residual -- needed four VALU per element to decode and lost to f16x.)
//   f16x   : fp16 operands as above, but X stays fp32 in HBM (round 4) -- the target is never rounded, so the mode is
//            parity-grade on data fp16 does not hold exactly (STFT magnitudes, plain floats) at 1x MFMA work; the X
//            stream doubles (HBM-bound: 1.07 GB per half-step at configs[1]).  The ratio is formed in fp32 from the
//            unrounded x; for beta = 2 the target IS the second GEMM's operand and goes in as an fp16 hi + lo pair
//            (two MFMAs for that product only).
//
// Tried and measured slower (round 3, profiles/r03_clock.md): a cross-tile software pipeline for the two-accumulator
// instances (elementwise of tile t interleaved with GEMM2 of tile t-1 through sched_group_barrier, three LDS stages):
// hipcc's placement degraded the counted LDS waits to lgkmcnt(0) -- 2 120 instead of 2 330 it/s at beta = 2, rank 128;
// and the same interleave inside ONE tile with the order fixed in the source and fenced by sched_barrier(0) (elementwise
// of S^T tile 0 under GEMM1 of tile 1, of tile 1 under the first half of GEMM2): 1 640 instead of 1 790 it/s at
// beta = 0.5, 191 instead of 199 at the configs[4] shard -- the fences cost more than the overlap returns.
//
// Workgroup = 4 waves (one per SIMD), wave w owns rows 32w..32w+31 of the
// block.  The panel tile (64 rows of B, both images) is double-buffered in
// LDS, filled by LDS-DMA (global_load_lds, issued from inline asm so that hipcc keeps its counted lgkmcnt waits for
// the operand rings); the X tile is prefetched one tile ahead in VGPRs with non-temporal loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "nmfmu_layout.h"

#ifndef NMFMU_FUSED_G1_ASM
#define NMFMU_FUSED_G1_ASM 1
#endif
#ifndef NMFMU_FUSED_TR_MINR
#define NMFMU_FUSED_TR_MINR 32    // single-plane MU instances at padded rank >= this stage ONE panel image (FusedCfg::TR); 999 = off
#endif


namespace nmfmu {

constexpr float kEps = 1.1920928955078125e-07f;  // constants.py:3 of the reference

// kSqrt / kSqrt3: beta = 0.5 / 1.5 -- the generic branch of nmf.py:71-74 with v_rsq_f32 instead of log2 / exp2
enum BetaKind : int { kKL = 0, kEuc = 1, kIS = 2, kGen = 3, kSqrt = 4, kSqrt3 = 5 };
// kModeDen: the positive term alone, den = Gp(S) @ panel with no target at all (sparse targets with a generic beta:
// the reference's dense pass of nmf.py:628-636).  The loss mode also runs without a target (xp == nullptr: X = 0).
// kModeXB (round 4): beta == 2 without the reconstruction.  nmf.py:61-63 puts no eps inside the two grad_outputs, so the
// numerator is X @ panel exactly -- ONE streaming MFMA GEMM over X, no S tiles, no elementwise stage -- and the denominator
// owner @ (panel^T panel) needs only the rank x rank Gram matrix (nmfmu_gram_panel): in the fused-apply epilogue it is one
// more small MFMA product per workgroup (owner fragments x Gram image), otherwise the apply kernel forms it from the fp32
// master.  4*N*C*R flops per iteration instead of 12*N*C*R; HBM-bound on the X stream.
// kModeMU2: kModeMU for a SPLIT panel -- the row-major image (first GEMM) and the transposed image (second GEMM) hold
// different matrices (PLCA: the Z-scaled factor and the unscaled one), so both are staged (no FusedCfg::TR).
enum FusedMode : int { kModeMU = 0, kModeLoss = 1, kModeDen = 2, kModeXB = 3, kModeMU2 = 4 };
enum Precision : int { kPrecBf16 = 0, kPrecX3 = 1, kPrecF16 = 2, kPrecF16X = 3, kPrecF16R = 4 };   // = NMFMU_PREC_* of include/nmfmu.h

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// Compile-time loop: body(std::integral_constant<int, I>{}) for I = 0 .. N-1.  Used wherever the register-resident
// accumulator arrays are indexed outside the main loop: `#pragma unroll` is only a hint, and when hipcc declines it
// for a large body (the padded-rank-256 epilogue) the runtime index moves the whole array into scratch memory -- and
// keeps that copy updated from inside the main loop (measured: 21 KiB of scratch stores per wave and tile, 4x the
// L2 traffic, 2.3x the run time).
template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& body) {
  if constexpr (I < N) {
    body(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(body));
  }
}

struct FusedArgs {
  const void* xp;         // fragment-order X tiles (bf16 / fp16 or fp32)
  const uint16_t* p1_hi;  // panel, row-major image
  const uint16_t* p1_lo;
  const uint16_t* p2_hi;  // panel, transposed tiles
  const uint16_t* p2_lo;
  const uint16_t* a1_hi;  // owner, row-major image
  const uint16_t* a1_lo;
  float* slab_num;        // [nsplit][M_pad][R_PAD]
  float* slab_den;        // same, beta != 1
  float* loss_part;       // [gridDim.x], loss mode
  int M, K;               // logical sizes (loss masking, fp16 scale)
  int M_pad, ktiles, nsplit, tiles_per_split;
  float beta;
  // fused apply (beta == 1, nsplit == 1): the epilogue performs nmf.py:78-92 and re-emits the owner's images
  int fuse_apply;
  int rank;               // logical rank (row pitch of f)
  float* f;               // owner fp32 master [M][rank]
  const float* kl_den;    // [R_PAD] column sums of the panel
  uint16_t* o1_hi;        // owner images to refresh (same buffers a1_* were read from)
  uint16_t* o1_lo;
  uint16_t* o2_hi;
  uint16_t* o2_lo;
  float* colsum_part;     // [M_pad / BM][R_PAD]
  float l1, l2, gamma;
  // fp16 operands, beta < 1: column sums of owner and panel ([R_PAD] each) -> typical S -> power-of-two scale of Gn / Gp
  const float* cs_owner;
  const float* cs_panel;
  // kModeXB, fused apply: Gram matrix of the panel as 16-bit images [R_PAD][R_PAD] (row r = column r of the symmetric
  // matrix, scaled by 2^-gram_exp[r] so that it sits inside fp16's range; lo plane = the rounding remainder of hi)
  const uint16_t* gram_hi;
  const uint16_t* gram_lo;
  const float* gram_scale;   // [R_PAD]: 2^gram_exp[r], multiplied back onto the denominator column r
  uint32_t* status;       // or nullptr: bit 0 is set when the fused apply had to clamp an fp16 image value at 65504
  void* debug;            // nmfmu_step.stamps: clock stamps of the ping-pong / software-pipelined kernels (tools/pp_timeline.py); kModeXB: diagnostic builds
};

template <int R_PAD, int BETA, int PREC, int MODE>
struct FusedCfg {
  static constexpr bool X3 = PREC == kPrecX3;                           // two operand planes (hi / lo)
  static constexpr bool F16 = PREC == kPrecF16 || PREC == kPrecF16X || PREC == kPrecF16R;   // fp16 operand type
  static constexpr bool XR = PREC == kPrecF16R;                        // X = the top 24 bits of the fp32 (3 bytes per element)
  static constexpr bool XF32 = X3 || PREC == kPrecF16X;                // X stored fp32 (fragment order, 8 chunks per lane)
  static constexpr int BM = 128, WAVES = 4, THREADS = 256;
  static constexpr int KS = R_PAD / 16;      // k-steps of GEMM1 (contraction over rank)
  static constexpr int RT = R_PAD / 32;      // 32-wide rank tiles of GEMM2's output
  static constexpr int ROWB = 2 * R_PAD;     // bytes per P1 row
  static constexpr int IMG = kBK * ROWB;     // bytes of one image tile (= 128 * R_PAD)
  static constexpr int NPL = X3 ? 2 : 1;     // planes (hi / lo)
  static constexpr bool LOSS = MODE == kModeLoss;
  static constexpr bool DEN = MODE == kModeDen;
  static constexpr bool XB = MODE == kModeXB;   // numerator = X @ panel only (beta == 2); only the P2 image is staged
  // TR (round 4): padded rank 256, beta == 1, single plane -- the configs[4] shard.  Its k-tile is 64 KiB of panel (two
  // images of 32 KiB), i.e. SIXTEEN 1-KiB LDS-DMA pieces per wave and tile at ~150 issue cycles each next to 2 048 cycles
  // of MFMA, in ONE wave per SIMD: the kernel is DMA-issue-bound (and, at 1 345 W / 1.96 GHz, not at the power limit).
  // With TR only the row-major image is staged and the second GEMM's k-contiguous operands are gathered from it by the
  // transposing read ds_read_b64_tr_b16 (two per MFMA): half the pieces.  (The same idea lost in the ping-pong kernel,
  // whose DMA issue is hidden behind the other half's MFMAs and which sits at the power limit -- round 2.)
  static constexpr bool TR = R_PAD >= NMFMU_FUSED_TR_MINR && !X3 && MODE == kModeMU &&
                             (R_PAD < 256 || BETA == kKL);   // (rank 256 with two accumulator sets: no registers for it)
  static constexpr int NIMG = ((LOSS || XB || TR) ? 1 : 2) * NPL;
  static constexpr int P1HI = 0;
  static constexpr int P1LO = IMG;           // valid when X3
  static constexpr int P2HI = XB ? 0 : NPL * IMG;
  static constexpr int P2LO = P2HI + IMG;
  static constexpr int STAGE_BYTES = NIMG * IMG;
  static constexpr int NQ = XF32 ? 8 : (XR ? 6 : 4);    // 16-byte X chunks per lane per tile (f16r: 4 of high halves + 2 of third bytes)
  static constexpr bool TWO_ACC = (BETA != kKL) && !LOSS && !DEN;
  // fp32 target as an fp16 MFMA operand (f16x, beta = 2: Gn = X): hi + lo pair, two MFMAs for the numerator product
  static constexpr bool XSPLIT = PREC == kPrecF16X && BETA == kEuc && !LOSS && !DEN;   // (kModeXB included)
  static constexpr bool GNLO = X3 || XSPLIT;                           // Gn has a low plane
  static constexpr int PASSES = IMG / 4096;  // 256 threads x 16 B per pass
  // fp16 operands: Gn / Gp of the branches with negative powers of S carry a power-of-two scale
  static constexpr bool SCALE = F16 && !LOSS && (BETA == kIS || BETA == kGen || BETA == kSqrt);
  // DEEP: the HBM-bound kModeXB instances stream through a ring of NSTAGE LDS stages / register buffers with counted
  // waits (NSTAGE - 1 tiles in flight behind the one being multiplied).  (The fp32-target mode 'f16x' was tried on the
  // same loop with one workgroup per CU and three fp32 X buffers -- r4d: 1 975 vs 2 270-2 370 it/s on the generic loop with
  // two workgroups per CU at beta = 1, 1 640 vs 1 690 at beta = 0.5; not kept.)
  static constexpr bool DEEP = XB;
  // workgroups per CU the register budget is sized for: the streaming kModeXB kernel wants two (bytes in flight)
  static constexpr int MINW = XB ? (R_PAD <= 128 ? 2 : 1) : ((X3 || TWO_ACC || R_PAD > 128) ? 1 : 2);
  // GEMM1 as asm with VGPR constraints (see gemm1()): the single-plane instances that run one wave per SIMD
  static constexpr bool G1_ASM = NMFMU_FUSED_G1_ASM && !X3 && MINW == 1;
  // kModeXB streams: four (fp32 target: three) ring stages, tiles t+1 .. t+3 in flight behind tile t (counted waits); at padded rank 128 the
  // ring is exactly as large as its epilogue's staging tile (4 waves x 32 rows x R_PAD floats)
#ifndef NMFMU_XB_NSTAGE
#define NMFMU_XB_NSTAGE 4
#endif
  static constexpr int NSTAGE = DEEP ? (XF32 ? 3 : NMFMU_XB_NSTAGE) : 2;   // (fp32 X buffers are 32 registers each: three of them)
  // the fused-apply epilogue stages 4 waves x 32 rows x R_PAD floats in LDS (= two stages of two images; more than the
  // ring of the single-image instances)
  static constexpr int EPI_BYTES = (XB || TR) ? WAVES * 32 * R_PAD * 4 : 0;
  static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES > EPI_BYTES ? NSTAGE * STAGE_BYTES : EPI_BYTES;
};

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  bf16x2 r = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                 0, 0);
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }

// ---- operand element type of the single-plane kernels (here, nmfmu_pp.h, nmfmu_gemm.h): bf16, or fp16 (11 significant
// bits at the same MFMA rate; values clamped to 65504 when packed, conversions saturate under MODE.FP16_OVFL)
enum OperandType : int { kOpBf16 = 0, kOpF16 = 1 };

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

template <int OPT>
__device__ __forceinline__ f32x16 mfma_op(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (OPT == kOpF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return mfma_bf16(a, b, c);
}

// 16 x 16 x 32 (gfx950): A / B lane = (row lane & 15, k = 8 (lane >> 4) + i), C / D lane = (column lane & 15, rows 4 (lane >> 4) + i)
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int OPT>
__device__ __forceinline__ f32x4 mfma16_op(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (OPT == kOpF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return mfma16_bf16(a, b, c);
}

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  f32x2 v = {a, b};
  f16x2 r = __builtin_convertvector(v, f16x2);  // v_cvt_pk_f16_f32 (RNE; saturates under MODE.FP16_OVFL)
  return __builtin_bit_cast(uint32_t, r);
}
template <int OPT>
__device__ __forceinline__ uint32_t pack_op(float a, float b) {
  if constexpr (OPT == kOpF16) return pack_f16(a, b);
  else return pack_bf16(a, b);
}
template <int OPT>
__device__ __forceinline__ float unpack_lo(uint32_t w) {
  if constexpr (OPT == kOpF16) return (float)__builtin_bit_cast(f16x2, w)[0];
  else return bf16_lo(w);
}
template <int OPT>
__device__ __forceinline__ float unpack_hi(uint32_t w) {
  if constexpr (OPT == kOpF16) return (float)__builtin_bit_cast(f16x2, w)[1];
  else return bf16_hi(w);
}

__device__ __forceinline__ uint32_t pack_img(float a, float b, int f16) {
  return f16 ? pack_f16(fminf(a, 65504.f), fminf(b, 65504.f)) : pack_bf16(a, b);
}

// nmf.py:61-74: the two "grad_output" tensors, per element.  `s` already
// contains +eps (the S accumulator is initialised with eps) except for
// beta == 2, where the reference adds none.
template <int BETA>
__device__ __forceinline__ void mu_elem(float s, float x, float beta, float& gn, float& gp) {
  if constexpr (BETA == kKL) {
    gn = x * __builtin_amdgcn_rcpf(s);
    gp = 0.f;
  } else if constexpr (BETA == kEuc) {
    gn = x;
    gp = s;
  } else if constexpr (BETA == kIS) {
    gp = __builtin_amdgcn_rcpf(s);
    gn = gp * gp * x;
  } else if constexpr (BETA == kSqrt) {      // beta = 0.5: s^-0.5, s^-1.5 x
    gp = __builtin_amdgcn_rsqf(s);
    gn = gp * gp * gp * x;
  } else if constexpr (BETA == kSqrt3) {     // beta = 1.5: s^0.5, s^-0.5 x
    const float r = __builtin_amdgcn_rsqf(s);
    gp = s * r;
    gn = r * x;
  } else {
    const float lg = __builtin_amdgcn_logf(s);  // log2
    gp = __builtin_amdgcn_exp2f((beta - 1.f) * lg);
    gn = gp * __builtin_amdgcn_rcpf(s) * x;
  }
}

// The same terms times 2^ki (fp16 operands; ki is wave-uniform, so the scaling is one v_ldexp_f32 with a scalar
// operand, or rides in the exponent of the generic branch's exp2).
template <int BETA>
__device__ __forceinline__ void mu_elem_scaled(float s, float x, float beta, int ki, float& gn, float& gp) {
  if constexpr (BETA == kIS) {
    const float r = __builtin_amdgcn_rcpf(s);
    gp = __builtin_ldexpf(r, ki);
    gn = gp * r * x;
  } else if constexpr (BETA == kSqrt) {
    const float r = __builtin_amdgcn_rsqf(s);
    gp = __builtin_ldexpf(r, ki);
    gn = gp * r * r * x;
  } else if constexpr (BETA == kGen) {
    const float lg = __builtin_amdgcn_logf(s);
    gp = __builtin_amdgcn_exp2f((beta - 1.f) * lg + (float)ki);
    gn = gp * __builtin_amdgcn_rcpf(s) * x;
  } else {
    mu_elem<BETA>(s, x, beta, gn, gp);
  }
}

// (neg / pos) ^ gamma of nmf.py:89-92 inside the fused epilogues: both terms are >= eps, so the ratio is positive and
// finite -- exp2(gamma * log2(m)) with the hardware transcendentals (1 ulp each, ~1e-7 relative on the result) instead of
// libm's powf, whose ~40 instructions per element made the beta < 1 epilogues 70 us long at the loop's low clock.  gamma
// = 1/2 (beta = 0) is a square root.
__device__ __forceinline__ float mu_pow(float m, float gamma) {
  if (gamma == 0.5f) return __builtin_amdgcn_sqrtf(m);
  return __builtin_amdgcn_exp2f(gamma * __builtin_amdgcn_logf(m));
}

// metrics.py:6-96 per element.  `s` as above.
template <int BETA>
__device__ __forceinline__ float loss_elem(float s, float x, float beta) {
  constexpr float kLn2 = 0.6931471805599453f;
  if constexpr (BETA == kEuc) {
    const float d = s - x;
    return 0.5f * d * d;
  } else if constexpr (BETA == kKL) {
    const float lx = __builtin_amdgcn_logf(x + kEps), ls = __builtin_amdgcn_logf(s);
    return x * ((lx - ls) * kLn2) - x + (s - kEps);
  } else if constexpr (BETA == kIS) {
    const float xe = x + kEps;
    const float lx = __builtin_amdgcn_logf(xe), ls = __builtin_amdgcn_logf(s);
    return xe * __builtin_amdgcn_rcpf(s) - (lx - ls) * kLn2 - 1.f;
  } else {
    const float xb = beta < 0.f ? x + kEps : x;
    const float t1 = xb > 0.f ? __builtin_amdgcn_exp2f(beta * __builtin_amdgcn_logf(xb)) : 0.f;
    const float sb1 = __builtin_amdgcn_exp2f((beta - 1.f) * __builtin_amdgcn_logf(s));
    const float t2 = sb1 * s;
    return (t1 + (beta - 1.f) * t2 - beta * xb * sb1) / (beta * (beta - 1.f));
  }
}

// ---- helpers of the kModeXB stream (free functions: every register-array index must be a compile-time constant -- an
// un-unrolled `#pragma unroll` loop over an asm-loaded array demotes it to scratch memory, seen at padded rank 256)
template <int NQ, int Q = 0>
__device__ __forceinline__ void xb_load_x(const char* p, u32x4 (&x)[NQ]) {   // this lane's NQ 16-byte chunks of a tile, by asm
  if constexpr (Q < NQ) {
#ifndef NMFMU_XB_XPOL
#define NMFMU_XB_XPOL " nt"
#endif
#ifdef NMFMU_XB_ABL_NOX
    x[Q] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};   // timing-only: no X stream
    asm volatile("" : "+v"(x[Q]));
#else
    asm volatile("global_load_dwordx4 %0, %1, off" NMFMU_XB_XPOL : "=&v"(x[Q]) : "v"(p + Q * 1024) : "memory");
#endif
    xb_load_x<NQ, Q + 1>(p, x);
  }
}
template <int NQ, int Q = 0>
__device__ __forceinline__ void xb_tie(u32x4 (&x)[NQ]) {   // "these registers are written by the loads above"
  if constexpr (Q < NQ) {
    asm volatile("" : "+v"(x[Q]));
    xb_tie<NQ, Q + 1>(x);
  }
}
// the target IS the second GEMM's A operand: the stored 16-bit words as they are, or (fp32 target) an fp16 hi + lo pair
template <bool XF32, bool GNLO, int OPT, int NQ, class G>
__device__ __forceinline__ void xb_to_ops(const u32x4 (&x)[NQ], G& g) {
  static_for<16>([&](auto ic) {
    constexpr int tt = decltype(ic)::value >> 3, d = decltype(ic)::value & 7;
    if constexpr (XF32) {
      const uint32_t u0 = x[4 * tt + (d >> 1)][2 * (d & 1)], u1 = x[4 * tt + (d >> 1)][2 * (d & 1) + 1];
      const float x0 = __builtin_bit_cast(float, u0), x1 = __builtin_bit_cast(float, u1);
      const uint32_t nh = pack_op<OPT>(x0, x1);
      g.gnh[tt][d] = nh;
      if constexpr (GNLO) g.gnl[tt][d] = pack_op<OPT>(x0 - unpack_lo<OPT>(nh), x1 - unpack_hi<OPT>(nh));
    } else {
      g.gnh[tt][d] = x[2 * tt + (d >> 2)][d & 3];
    }
  });
}

template <int R_PAD, int BETA, int PREC, int MODE>
__global__ void __launch_bounds__(256, (FusedCfg<R_PAD, BETA, PREC, MODE>::MINW)) fused_kernel(const FusedArgs a) {
  using C = FusedCfg<R_PAD, BETA, PREC, MODE>;
  constexpr bool X3 = C::X3;
  constexpr int OPT = C::F16 ? kOpF16 : kOpBf16;
  constexpr int BM = C::BM;
  constexpr int KS = C::KS, RT = C::RT, ROWB = C::ROWB, IMG = C::IMG, NQ = C::NQ;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;   // MFMA column = owner row within the wave's 32
  const int hl = lane >> 5;  // lane half
  const int mb = blockIdx.x / a.nsplit;
  const int ks = blockIdx.x - mb * a.nsplit;
  const int t0 = ks * a.tiles_per_split;
  const int t1 = min(t0 + a.tiles_per_split, a.ktiles);
  const int m0 = mb * BM + wave * 32 + j;

  if constexpr (C::F16) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");  // FP16_OVFL: saturate
  // Diagnostic builds (make EXTRA=-DNMFMU_DEBUG_HOOKS; tools/xb_timeline.py), kModeXB: wave 0 of every workgroup records the
  // constant 100 MHz clock at kernel entry (slot 2), at the start (0) and the end (1) of the tile loop and at exit (3),
  // and where it ran (slot 4: XCC_ID << 32 | HW_ID) -- the layout of the ping-pong kernel's stamps (nmfmu_pp.h)
  auto stamp = [&](int slot) {
#ifdef NMFMU_DEBUG_HOOKS
    if constexpr (C::XB) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.debug);
      if (!dbg || wave != 0) return;
      const unsigned long long r = __builtin_amdgcn_s_memrealtime();
      if (lane == 0) {
        dbg[64 + 5 * blockIdx.x + slot] = r;
        if (slot == 2)
          dbg[64 + 5 * blockIdx.x + 4] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |
                                         (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));
      }
    }
#endif
  };
  stamp(2);

  // ---- fp16 operands, beta < 1: scale 2^ki of Gn / Gp from the typical S = sum_r colsum_A[r] colsum_B[r] / (M K)
  // (every workgroup computes the same value in the same order; ki ends up in a scalar register)
  int ki = 0;
  if constexpr (C::SCALE) {
    if (a.cs_owner && a.cs_panel) {
      float p = 0.f;
      for (int r = lane; r < R_PAD; r += 64) p += a.cs_owner[r] * a.cs_panel[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
      const float styp = p / ((float)a.M * (float)a.K) + kEps;
      if (styp > 0.f && styp < 3.0e38f) {
        const float bexp = BETA == kIS ? -1.f : (BETA == kSqrt ? -0.5f : a.beta - 1.f);
        ki = (int)fminf(fmaxf(-rintf(bexp * log2f(styp)), -40.f), 40.f);
      }
      ki = __builtin_amdgcn_readfirstlane(ki);
    }
  }

  // ---- owner fragments (B operand of GEMM1): row m, rank slice 16*kk + 8*hl .. +7
  u32x4 qh[KS];
  u32x4 ql[X3 ? KS : 1];
  auto load_owner_frags = [&]() {
    const int sw = P1Swz<R_PAD>::of(m0) << 4;
    const char* rowh = reinterpret_cast<const char*>(a.a1_hi) + (size_t)m0 * ROWB;
    const char* rowl = reinterpret_cast<const char*>(a.a1_lo) + (size_t)m0 * ROWB;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int off = (kk * 32 + hl * 16) ^ sw;
      qh[kk] = ld16(rowh + off);
      if constexpr (X3) ql[kk] = ld16(rowl + off);
    }
  };
  // (kModeXB has no first GEMM: its denominator product reads them in the epilogue and fetches them there -- 4 * KS
  // registers less across the stream loop, and nothing queued in front of the first X tile)
  if constexpr (!C::XB) load_owner_frags();

  // ---- per-lane LDS offsets
  // GEMM1 A operand: MFMA row i = j of S^T tile tt reads panel row pi_tt(j); with this
  // permutation accumulator register r of lane (j, hl) is contraction column 32*hl + 16*tt + r.
  int a_row[2], a_sw[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int row = 32 * ((j >> 2) & 1) + 16 * tt + (j & 3) + 4 * (j >> 3);
    a_row[tt] = row * ROWB;
    a_sw[tt] = P1Swz<R_PAD>::of(row) << 4;
  }
  // GEMM2 B operand: rank column 32*rt + j, contraction slice 32*hl + 16*tt + 8*m2 .. +7
  const int b_row = j * 128;
  int b_off[2][2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) b_off[tt][m2] = ((4 * hl + 2 * tt + m2) << 4) ^ (((j >> 1) & 7) << 4);

  f32x16 on[C::LOSS ? 1 : RT];
  f32x16 op[C::TWO_ACC ? RT : 1];
#pragma unroll
  for (int rt = 0; rt < (C::LOSS ? 1 : RT); ++rt)
#pragma unroll
    for (int e = 0; e < 16; ++e) on[rt][e] = 0.f;
#pragma unroll
  for (int rt = 0; rt < (C::TWO_ACC ? RT : 1); ++rt)
#pragma unroll
    for (int e = 0; e < 16; ++e) op[rt][e] = 0.f;
  float lacc = 0.f;

  const char* xbase = reinterpret_cast<const char*>(a.xp) + ((size_t)mb * a.ktiles * 4 + wave) * (NQ * 1024) + lane * 16;
  const bool no_x = C::DEN || a.xp == nullptr;   // (uniform) no target: the X registers stay zero
  auto load_x = [&](int t, u32x4(&x)[NQ]) {
    const char* p = xbase + (size_t)t * (4 * NQ * 1024);
    if (no_x) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) x[q] = u32x4{0u, 0u, 0u, 0u};
      return;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)   // read once: non-temporal, keeps the factor panel resident in L2
      x[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + q * 1024));
  };

  // ---- panel staging: every image tile is one contiguous, pre-swizzled block in HBM
  const char* img_src[C::NIMG];
  if constexpr (C::XB) {
    img_src[0] = reinterpret_cast<const char*>(a.p2_hi);
    if constexpr (X3) img_src[1] = reinterpret_cast<const char*>(a.p2_lo);
  } else {
    img_src[0] = reinterpret_cast<const char*>(a.p1_hi);
    if constexpr (X3) img_src[1] = reinterpret_cast<const char*>(a.p1_lo);
    if constexpr (!C::LOSS && !C::TR) {
      img_src[C::NPL] = reinterpret_cast<const char*>(a.p2_hi);
      if constexpr (X3) img_src[C::NPL + 1] = reinterpret_cast<const char*>(a.p2_lo);
    }
  }
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
  const unsigned wave_lds = (unsigned)wave * 1024u;
  // Issued from inline asm on purpose: while hipcc knows an LDS-DMA is in flight it turns every LDS wait into
  // lgkmcnt(0), which serialises the operand prefetch rings.  Completion is waited for explicitly (vmcnt(0) before the
  // tile's barrier, see the main loop).  M0 = LDS byte address of the wave's 1 KiB piece.
  auto stage_issue = [&](int t, int stage_off) {   // stage_off: byte offset of the LDS stage
#pragma unroll
    for (int im = 0; im < C::NIMG; ++im) {
      const char* src = img_src[im] + (size_t)t * IMG + tid * 16;
#pragma unroll
      for (int p = 0; p < C::PASSES; ++p) {
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage_off + im * IMG + p * 4096) + wave_lds);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :
                     : "v"(src + p * 4096), "s"(lds_addr)
                     : "memory", "m0");
      }
    }
  };

  f32x16 epsv;
#pragma unroll
  for (int e = 0; e < 16; ++e) epsv[e] = (BETA == kEuc) ? 0.f : kEps;
  // asm GEMM1: the seed tile is laundered through an empty asm so that hipcc keeps it resident instead of
  // re-materialising it with v_mov right in front of the (asm, hence unpadded) MFMA that reads it as its C operand --
  // a VALU write -> XDL SrcC read hazard (seen at padded rank 32, where nothing else sits between the two)
  if constexpr (C::G1_ASM && BETA != kEuc) asm volatile("" : "+v"(epsv));

  // The 16-bit A operands of GEMM2 (Gn, Gp; hi / lo planes), one set per tile
  struct GOps {
    uint32_t gnh[2][8], gnl[C::GNLO ? 2 : 1][8], gph[C::TWO_ACC ? 2 : 1][8], gpl[(C::TWO_ACC && X3) ? 2 : 1][8];
  };

  // ---------------- GEMM1: S^T tiles (panel rows x owner rows), contraction over rank.
  // The panel operands are fetched through a PF-deep register ring so that PF-1 ds_read_b128 are always in
  // flight behind the MFMA that is issuing (hipcc otherwise emits read -> lgkmcnt(0) -> mfma, one at a time).
  // The accumulators are seeded with eps through the C operand of each chain's first MFMA (seed tile `epsv`,
  // loop invariant) instead of being re-initialised with 16 moves per tile.
  auto gemm1 = [&](const char* sb, f32x16(&s)[2]) {
    constexpr int NSTEP = 2 * KS;
    constexpr int PF = NSTEP < 4 ? NSTEP : 4;
    u32x4 ring_h[PF];
    u32x4 ring_l[X3 ? PF : 1];
    // step -> (kk, tt): the two S^T tiles alternate, so consecutive MFMAs never share an accumulator
    auto a_off = [&](int step) {
      const int tt = step & 1, kk = step >> 1;
      return a_row[tt] + ((kk * 32 + hl * 16) ^ a_sw[tt]);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      ring_h[p] = ld16(sb + C::P1HI + a_off(p));
      if constexpr (X3) ring_l[p] = ld16(sb + C::P1LO + a_off(p));
    }
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int tt = step & 1, kk = step >> 1;
      const u32x4 ah = ring_h[step % PF];
      u32x4 al;
      if constexpr (X3) al = ring_l[step % PF];
      if (step + PF < NSTEP) {
        ring_h[step % PF] = ld16(sb + C::P1HI + a_off(step + PF));
        if constexpr (X3) ring_l[step % PF] = ld16(sb + C::P1LO + a_off(step + PF));
      }
      if constexpr (X3) {
        s[tt] = mfma_bf16(al, qh[kk], kk == 0 ? epsv : s[tt]);
        s[tt] = mfma_bf16(ah, ql[kk], s[tt]);
        s[tt] = mfma_bf16(ah, qh[kk], s[tt]);
      } else if constexpr (C::G1_ASM) {
        // The one-wave-per-SIMD instances keep their rank-wide accumulators in AGPRs, and hipcc then selects the
        // AGPR form for EVERY MFMA of the kernel: the S tiles would be seeded with 32 v_accvgpr_write and read back
        // with 32+ v_accvgpr_read per tile (a quarter of the loop's instructions).  Written as asm with VGPR
        // constraints the S tiles stay where the elementwise stage needs them.
        if (kk == 0) {
          if constexpr (BETA == kEuc) {
            if constexpr (OPT == kOpF16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s[tt]) : "v"(ah), "v"(qh[0]));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s[tt]) : "v"(ah), "v"(qh[0]));
          } else {
            if constexpr (OPT == kOpF16) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(s[tt]) : "v"(ah), "v"(qh[0]), "v"(epsv));
            else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(s[tt]) : "v"(ah), "v"(qh[0]), "v"(epsv));
          }
        } else {
          if constexpr (OPT == kOpF16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s[tt]) : "v"(ah), "v"(qh[kk]));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s[tt]) : "v"(ah), "v"(qh[kk]));
        }
      } else {
        s[tt] = mfma_op<OPT>(ah, qh[kk], kk == 0 ? epsv : s[tt]);
      }
    }
    // asm MFMAs are not padded by hipcc: the S tiles are read by the VALU next (XDL write -> VALU read, 18 wait states
    // after a 16-pass MFMA)
    if constexpr (C::G1_ASM) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(s[0]), "+v"(s[1]));
    // pin the software pipeline: PF reads up front, then one read behind every MFMA group (the asm MFMAs keep their
    // program order by themselves)
    if constexpr (!C::G1_ASM) {
      __builtin_amdgcn_sched_group_barrier(0x100, PF * C::NPL, 0);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        __builtin_amdgcn_sched_group_barrier(0x008, X3 ? 3 : 1, 0);
        if (step + PF < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, C::NPL, 0);
      }
    }
  };

  // ---------------- elementwise: Gn / Gp (or the loss terms) of tile t, packed to 16-bit A operands.
  // elem_pair: the two elements (columns 2d, 2d+1 of S^T tile tt) of this lane.
  auto elem_pair = [&](int t, const f32x16& stt, const u32x4(&x)[NQ], int tt, int d, GOps& g) {
    float x0, x1;
    if constexpr (C::XF32) {
      // NB: extract to scalars first -- __builtin_bit_cast on an ext-vector ELEMENT lvalue reads element 0
      // (hipcc 7.2), which silently turned every 16-byte chunk into a splat of its first float.
      const uint32_t u0 = x[4 * tt + (d >> 1)][2 * (d & 1)];
      const uint32_t u1 = x[4 * tt + (d >> 1)][2 * (d & 1) + 1];
      x0 = __builtin_bit_cast(float, u0);
      x1 = __builtin_bit_cast(float, u1);
    } else if constexpr (C::XR) {
      // 3-byte target: bits 31..16 in the 16-bit word of the f16 layout, bits 15..8 = byte (16 tt + 2 d [+ 1]) of this lane's
      // 32 third bytes (chunks 4, 5) -- ONE v_perm_b32 per element puts the fp32 back together (selector bytes: 4..7 = the
      // first source, 0..3 = the second, 0x0c = zero)
      const uint32_t w = x[2 * tt + (d >> 2)][d & 3], uw = x[4 + tt][d >> 1];
      const uint32_t j0 = 2 * (d & 1);
      x0 = __builtin_bit_cast(float, __builtin_amdgcn_perm(w, uw, 0x0504000cu | (j0 << 8)));
      x1 = __builtin_bit_cast(float, __builtin_amdgcn_perm(w, uw, 0x0706000cu | ((j0 + 1) << 8)));
    } else {
      const uint32_t w = x[2 * tt + (d >> 2)][d & 3];
      x0 = unpack_lo<OPT>(w);
      x1 = unpack_hi<OPT>(w);
    }
    const float s0 = stt[2 * d], s1 = stt[2 * d + 1];
    if constexpr (C::LOSS) {
      constexpr int LB = (BETA == kSqrt || BETA == kSqrt3) ? (int)kGen : BETA;
      const int k0 = t * kBK + 32 * hl + 16 * tt + 2 * d;
      const bool rowok = m0 < a.M;
      lacc += (rowok && k0 < a.K) ? loss_elem<LB>(s0, x0, a.beta) : 0.f;
      lacc += (rowok && k0 + 1 < a.K) ? loss_elem<LB>(s1, x1, a.beta) : 0.f;
    } else {
      float n0, n1, p0, p1;
      // fp16 target: the factor of Gn that does not depend on x first, then ONE v_fma_mix_f32 per element multiplies it
      // by the fp16 half of the stored word (no separate conversion of x)
      constexpr bool MIX = C::F16 && !C::XF32 && !C::XR && BETA != kEuc;
      const float xa = MIX ? 1.f : x0, xb = MIX ? 1.f : x1;
#ifdef NMFMU_FUSED_ABL_NOELEM
      // timing-only ablation (wrong results; round 5): the transcendental / multiply chain of nmf.py:61-74 is
      // gone, conversions and packing stay -- an UPPER bound of what hiding that chain behind the MFMAs could return, since the
      // chain's own energy is removed as well (tools/gpu_r5g.sh, profiles/r05g_beta_lt1.md)
      n0 = xa, n1 = xb, p0 = s0, p1 = s1;
#else
      if constexpr (C::SCALE) {
        mu_elem_scaled<BETA>(s0, xa, a.beta, ki, n0, p0);
        mu_elem_scaled<BETA>(s1, xb, a.beta, ki, n1, p1);
      } else {
        mu_elem<BETA>(s0, xa, a.beta, n0, p0);
        mu_elem<BETA>(s1, xb, a.beta, n1, p1);
      }
#endif
      if constexpr (MIX) {
        const uint32_t w = x[2 * tt + (d >> 2)][d & 3];
        float m0, m1;
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(m0) : "v"(w), "v"(n0));
        asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(m1) : "v"(w), "v"(n1));
        n0 = m0, n1 = m1;
      }
      if constexpr (C::DEN) n0 = p0, n1 = p1;   // denominator-only pass: the one operand set carries Gp
      // (beta == 2 with one operand plane: Gn IS the stored target word -- no unpack / re-pack)
      const uint32_t nh = (BETA == kEuc && !C::XF32 && !C::DEN) ? x[2 * tt + (d >> 2)][d & 3] : pack_op<OPT>(n0, n1);
      g.gnh[tt][d] = nh;
      if constexpr (C::GNLO) g.gnl[tt][d] = pack_op<OPT>(n0 - unpack_lo<OPT>(nh), n1 - unpack_hi<OPT>(nh));
      if constexpr (C::TWO_ACC) {
        const uint32_t ph = pack_op<OPT>(p0, p1);
        g.gph[tt][d] = ph;
        if constexpr (X3) g.gpl[tt][d] = pack_bf16(p0 - bf16_lo(ph), p1 - bf16_hi(ph));
      }
    }
  };
  auto elementwise = [&](int t, const f32x16(&s)[2], const u32x4(&x)[NQ], GOps& g) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int d = 0; d < 8; ++d) elem_pair(t, s[tt], x, tt, d, g);
  };

  // ---------------- GEMM2: num/den (owner rows x rank), contraction over the tile's 64 columns
  // TR: ds_read_b64_tr_b16 works on groups of 16 lanes; lane 4a+b of a group receives element b of the 8-byte chunks
  // addressed by lanes a, a+4, a+8, a+12 (probed on gfx950: tools/ubench/tr_probe.hip).  With source lane s = a + 4i
  // pointing at (panel row k0 + i, ranks 4 (c0 + a) .. +3) the group reads a [4 rows] x [16 ranks] block and lane l ends
  // up with rank 4 c0 + l for rows k0 .. k0+3: two such reads (rows +0..3 and +4..7) are the B operand of GEMM2, whose
  // lane (j, hl) needs the 8 contraction rows 32 hl + 16 tt + 8 m2 + (0..7) of rank 32 rt + j.
  // t_base[tt][m2][h]: byte offset of this lane's chunk for rank tile 0; rank tile rt = XOR with rt * 64 (slot bits 2-4).
  int t_base[2][2][2] = {};
  if constexpr (C::TR) {
    const int grp = lane >> 4, s16 = lane & 15;
    const int cslot = 2 * (grp & 1) + ((s16 & 3) >> 1);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = 32 * (grp >> 1) + 16 * tt + 8 * m2 + 4 * h + (s16 >> 2);
          t_base[tt][m2][h] = row * ROWB + ((cslot ^ P1Swz<R_PAD>::of(row)) << 4) + 8 * (s16 & 1);
        }
  }
  using s16x4_t = __attribute__((ext_vector_type(4))) short;
  using u32x2_t = __attribute__((ext_vector_type(2))) uint32_t;
  auto rd_tr = [&](const char* sb, int off) -> u32x2_t {
    auto* p = (__attribute__((address_space(3))) s16x4_t*)(size_t)((unsigned)(size_t)(__attribute__((address_space(3))) char*)sb + (unsigned)off);
    return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(p));
  };
  auto gemm2 = [&](const char* sb, const GOps& g) {
    constexpr int NSTEP = RT * 4;
    constexpr int PF = 4;
    u32x4 ring_h[PF];
    u32x4 ring_l[X3 ? PF : 1];
    // step -> ((tt, m2), rt): rank tiles innermost, so consecutive MFMAs cycle through the RT accumulators
    auto b_offs = [&](int step) {
      const int rt = step % RT, c = step / RT;
      return rt * 4096 + b_row + b_off[c >> 1][c & 1];
    };
    auto fetch_h = [&](int step) -> u32x4 {   // GEMM2's B operand of `step`: from the transposed image, or gathered (TR)
      if constexpr (C::TR) {
        const int rt = step % RT, c = step / RT;
        const u32x2_t lo = rd_tr(sb, t_base[c >> 1][c & 1][0] ^ (rt * 64)), hi = rd_tr(sb, t_base[c >> 1][c & 1][1] ^ (rt * 64));
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
      } else {
        return ld16(sb + C::P2HI + b_offs(step));
      }
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      ring_h[p] = fetch_h(p);
      if constexpr (X3) ring_l[p] = ld16(sb + C::P2LO + b_offs(p));
    }
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int rt = step % RT, c = step / RT;
      const int tt = c >> 1, m2 = c & 1;
      const u32x4 bh = ring_h[step % PF];
      u32x4 bl;
      if constexpr (X3) bl = ring_l[step % PF];
      if (step + PF < NSTEP) {
        ring_h[step % PF] = fetch_h(step + PF);
        if constexpr (X3) ring_l[step % PF] = ld16(sb + C::P2LO + b_offs(step + PF));
      }
      const u32x4 nh = {g.gnh[tt][4 * m2], g.gnh[tt][4 * m2 + 1], g.gnh[tt][4 * m2 + 2], g.gnh[tt][4 * m2 + 3]};
      if constexpr (X3) {
        const u32x4 nl = {g.gnl[tt][4 * m2], g.gnl[tt][4 * m2 + 1], g.gnl[tt][4 * m2 + 2], g.gnl[tt][4 * m2 + 3]};
        on[rt] = mfma_bf16(nl, bh, on[rt]);
        on[rt] = mfma_bf16(nh, bl, on[rt]);
      } else if constexpr (C::XSPLIT) {
        const u32x4 nl = {g.gnl[tt][4 * m2], g.gnl[tt][4 * m2 + 1], g.gnl[tt][4 * m2 + 2], g.gnl[tt][4 * m2 + 3]};
        on[rt] = mfma_op<OPT>(nl, bh, on[rt]);
      }
      on[rt] = mfma_op<OPT>(nh, bh, on[rt]);
      if constexpr (C::TWO_ACC && !C::XB) {
        const u32x4 ph = {g.gph[tt][4 * m2], g.gph[tt][4 * m2 + 1], g.gph[tt][4 * m2 + 2], g.gph[tt][4 * m2 + 3]};
        if constexpr (X3) {
          const u32x4 pl = {g.gpl[tt][4 * m2], g.gpl[tt][4 * m2 + 1], g.gpl[tt][4 * m2 + 2], g.gpl[tt][4 * m2 + 3]};
          op[rt] = mfma_bf16(pl, bh, op[rt]);
          op[rt] = mfma_bf16(ph, bl, op[rt]);
        }
        op[rt] = mfma_op<OPT>(ph, bh, op[rt]);
      }
    }
    constexpr int RDS = C::TR ? 2 : C::NPL;   // LDS reads per operand
    __builtin_amdgcn_sched_group_barrier(0x100, PF * RDS, 1);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      __builtin_amdgcn_sched_group_barrier(0x008, X3 ? ((C::TWO_ACC && !C::XB) ? 6 : 3) : (((C::TWO_ACC && !C::XB) ? 2 : 1) + (C::XSPLIT ? 1 : 0)), 1);
      if (step + PF < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, RDS, 1);
    }
  };

  auto drain = [&]() {   // the asm DMA of this tile's successor has landed; every wave is done with this tile's stage
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // vmcnt(0)
    __syncthreads();
  };
  if constexpr (C::DEEP) {
    // ---------------- kModeXB: nothing but X @ panel (16 MFMAs per tile and wave).  HBM-bound and latency-sensitive, so
    // the streams run deep: the panel's P2 tiles through an NSTAGE-slot LDS ring and X through NSTAGE register buffers,
    // both NSTAGE - 1 tiles ahead, every load issued from inline asm with ONE counted wait per tile (tile t has landed when
    // at most the loads of tiles t+1 .. t+NSTAGE-2 are outstanding; tile t+NSTAGE-1 is issued right behind that wait's
    // barrier, into the stage / registers tile t-1 has just vacated).
    stamp(0);
    if (t0 < t1) {
      const int nt = t1 - t0;
      constexpr int NST = C::NSTAGE;
      // (timing-only ablations NMFMU_XB_ABL_{NOX,NODMA,NOGEMM}: tools/gpu_r5q.sh, profiles/r05q_beta2_stream.md)
#ifdef NMFMU_XB_ABL_NOX
      constexpr int OPS_X = 0;
#else
      constexpr int OPS_X = NQ;
#endif
#ifdef NMFMU_XB_ABL_NODMA
      constexpr int OPS_P = 0;
#else
      constexpr int OPS_P = C::NIMG * C::PASSES;
#endif
      constexpr int OPS = OPS_P + OPS_X;   // vm operations per tile and thread (LDS-DMA pieces + X chunks)
      static_assert((NST - 2) * OPS <= 63, "vmcnt range");
      u32x4 xr[NST][NQ];
#ifdef NMFMU_XB_ABL_NODMA
      auto stage_issue = [&](int, int) {};
#endif
#ifdef NMFMU_XB_ABL_TILEMAJOR
      // timing-only (wrong results): the same bytes in tile-major order -- at step t the workgroups read ADJACENT 16 KiB
      // blocks instead of blocks ktiles * 16 KiB apart (is the stream's rate a matter of DRAM channel / page locality?)
      const char* xtm = reinterpret_cast<const char*>(a.xp) + ((size_t)mb * 4 + wave) * (NQ * 1024) + lane * 16;
      const size_t xtm_stride = (size_t)(gridDim.x / a.nsplit) * (4 * NQ * 1024);
      auto load_x_asm = [&](int t, u32x4(&x)[NQ]) { xb_load_x<NQ>(xtm + (size_t)t * xtm_stride, x); };
#else
      auto load_x_asm = [&](int t, u32x4(&x)[NQ]) { xb_load_x<NQ>(xbase + (size_t)t * (4 * NQ * 1024), x); };
#endif
      auto landed = [&](int later, u32x4(&x)[NQ]) {   // `later` tiles issued after this one may still be in flight
        if (later >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * OPS) : "memory");
        else if (later == 1 && NST > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xb_tie<NQ>(x);
        __syncthreads();
      };
      auto to_ops = [&](const u32x4(&x)[NQ], GOps& g) { xb_to_ops<C::XF32, C::GNLO, OPT, NQ>(x, g); };
      auto tile_step = [&](int i, auto bc) {   // tile i lives in ring slot / register buffer b = i % NST
        constexpr int b = decltype(bc)::value, bn = (b + NST - 1) % NST;
        landed(min(nt - 1 - i, NST - 2), xr[b]);
        if (i + NST - 1 < nt) {
          stage_issue(t0 + i + NST - 1, bn * C::STAGE_BYTES);
          load_x_asm(t0 + i + NST - 1, xr[bn]);
        }
        GOps g;
        to_ops(xr[b], g);
#ifdef NMFMU_XB_ABL_NOGEMM
#pragma unroll
        for (int d = 0; d < 8; ++d) asm volatile("" ::"v"(g.gnh[0][d]), "v"(g.gnh[1][d]));   // timing-only: no MFMA, no LDS reads
#else
        gemm2(smem + b * C::STAGE_BYTES, g);
#endif
      };
      static_for<NST - 1>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        if (d < nt) {
          stage_issue(t0 + d, d * C::STAGE_BYTES);
          load_x_asm(t0 + d, xr[d]);
        }
      });
      int i = 0;
      for (; i + NST <= nt; i += NST)
        static_for<NST>([&](auto bc) { tile_step(i + decltype(bc)::value, bc); });
      static_for<NST - 1>([&](auto bc) {
        if (i + decltype(bc)::value < nt) tile_step(i + decltype(bc)::value, bc);
      });
      __syncthreads();   // the epilogue re-uses the ring as its staging tile
    }
    stamp(1);
  } else if (t0 < t1) {
    // ---------------- main loop: LDS double buffer for the panel, X in ONE register buffer that is refilled with the
    // next tile right after its last use; one drain + barrier per tile.
    u32x4 xc[NQ];
    const int nt = t1 - t0;
    stage_issue(t0, 0);
    load_x(t0, xc);
    drain();
    for (int i = 0; i < nt; ++i) {
      const int t = t0 + i, buf = i & 1;
      const bool more = i + 1 < nt;
      if (more) stage_issue(t + 1, (buf ^ 1) * C::STAGE_BYTES);
      const char* sb = smem + buf * C::STAGE_BYTES;
      f32x16 s[2];
      GOps g;
      gemm1(sb, s);
      elementwise(t, s, xc, g);
      // X's registers are dead from here on: fetch the next tile into them now (single X buffer; the loads have the
      // whole GEMM2 + barrier + next GEMM1 to land).
      if (more) load_x(t + 1, xc);
      if constexpr (!C::LOSS) gemm2(sb, g);
      drain();
    }
  }

  // ---------------- epilogue
  if constexpr (C::LOSS) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lacc += __shfl_xor(lacc, o, 64);
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();
    if (lane == 0) red[wave] = lacc;
    __syncthreads();
    if (tid == 0) a.loss_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  } else {
    // accumulator register e of lane (j, hl): row (e&3) + 8*(e>>2) + 4*hl, column 32*rt + j
    bool fused_done = false;
    if constexpr (C::XB && !X3) {
      // fused apply (unsplit contraction), or -- split contraction -- the denominator is left as ONE slab next to the
      // nsplit numerator slabs (a.slab_den; the apply kernel then has nothing to form).
      // (round 5) split contraction: the rank tiles of the denominator are shared out over the first min(nsplit, RT)
      // workgroups of the row block (tile rt belongs to ks == rt % nd) instead of all landing on ks == 0, whose epilogue was
      // the launch's straggler (28 us against 2-5 us, tools/xb_timeline.py)
      const int nd = a.nsplit < RT ? a.nsplit : RT;
      const bool den_fused = a.fuse_apply && R_PAD <= 128;
      const bool den_split = !a.fuse_apply && ks < nd && a.slab_den && a.gram_hi;
      if (den_fused || den_split) {
        // denominator of kModeXB: den[m][r] = sum_q owner[m][q] G[q][r] as MFMA(owner fragments, Gram image rows r): the
        // same (row from the register index, column from the lane) layout as the numerator accumulators.  The image row r
        // carries 2^-exp[r]; hi + lo planes make the matrix itself exact to 2^-22, what is left is the owner's own rounding.
        // (round 5) At padded rank 64 / 128 the image rows come through LDS: the ring is free by now, one LDS-DMA pass
        // fetches 4 KiB and all passes are in flight together, where the 2 * KS * RT dependent 16-byte global loads per
        // lane were a chain of L2 latencies.  LDS slot s of row r holds image slot s ^ gswz(r), so that the 16 lanes of a
        // ds_read_b128 group (16 consecutive rows, one image slot) touch 16 different bank groups.
        constexpr bool GLDS = R_PAD == 64 || R_PAD == 128;
        constexpr int GSP = R_PAD / 8, GRPL = GSP >= 16 ? 1 : 16 / GSP;   // slots per row, rows per 256-byte bank line
        constexpr int GPLANE = R_PAD * ROWB;                              // bytes of one image
        load_owner_frags();
        // the fused apply's master rows are TOUCHED here (one dword per 128-byte line of the wave's 32 rows) so that their
        // HBM latency overlaps the Gram staging and its MFMAs; the apply's own loads then hit L2.  (Holding the 64 values
        // themselves across the product was tried: 43 registers spilled.)
        constexpr int NTOUCH = 32 * R_PAD * 4 / 128 / 64;   // lines of the wave's rows per lane
        uint32_t touch[NTOUCH > 0 ? NTOUCH : 1];
        const bool touched = den_fused && NTOUCH > 0 && a.rank == R_PAD && mb * BM + wave * 32 + 32 <= a.M;
        if (touched) {
#pragma unroll
          for (int k = 0; k < NTOUCH; ++k) {
            const char* tp = reinterpret_cast<const char*>(a.f) + ((size_t)(mb * BM + wave * 32) * R_PAD * 4) + (size_t)(k * 64 + lane) * 128;
            asm volatile("global_load_dword %0, %1, off" : "=&v"(touch[k]) : "v"(tp) : "memory");
          }
        }
        if constexpr (GLDS) {
          constexpr int GP = 32 * ROWB / 4096;                            // passes per rank tile and plane
          static_for<RT>([&](auto rtc) {
            constexpr int rt = decltype(rtc)::value;
            if (den_fused || rt % nd == ks) {
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) {
                const char* gsrc = reinterpret_cast<const char*>(pl ? a.gram_lo : a.gram_hi);
#pragma unroll
                for (int p = 0; p < GP; ++p) {
                  const int o = rt * 32 * ROWB + p * 4096 + tid * 16, r = o / ROWB, sl = (o % ROWB) >> 4;
                  const char* src = gsrc + r * ROWB + ((sl ^ ((r / GRPL) & (GSP - 1))) << 4);
                  const unsigned lds_addr = __builtin_amdgcn_readfirstlane(
                      lds_base + (unsigned)(pl * GPLANE + rt * 32 * ROWB + p * 4096) + wave_lds);
                  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                               :
                               : "v"(src), "s"(lds_addr)
                               : "memory", "m0");
                }
              }
            }
          });
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
        if (touched) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < NTOUCH; ++k) asm volatile("" ::"v"(touch[k]));   // (their registers were reserved until here)
        }
        static_for<RT>([&](auto rtc) {   // (compile-time indices: op / qh must stay in registers)
          constexpr int rt = decltype(rtc)::value;
          if (den_fused || rt % nd == ks) {
            const int gr = rt * 32 + j;
            const char* grow = reinterpret_cast<const char*>(a.gram_hi) + (size_t)gr * ROWB + hl * 16;
            const char* grow_lo = reinterpret_cast<const char*>(a.gram_lo) + (size_t)gr * ROWB + hl * 16;
            const char* lrow = smem + gr * ROWB;
            const int gsw = (gr / GRPL) & (GSP - 1);
            static_for<KS>([&](auto kc) {
              constexpr int kk = decltype(kc)::value;
              if constexpr (GLDS) {
                const int off = ((2 * kk + hl) ^ gsw) << 4;
                op[rt] = mfma_op<OPT>(qh[kk], ld16(lrow + off), op[rt]);
                op[rt] = mfma_op<OPT>(qh[kk], ld16(lrow + GPLANE + off), op[rt]);
              } else {
                op[rt] = mfma_op<OPT>(qh[kk], ld16(grow + kk * 32), op[rt]);
                op[rt] = mfma_op<OPT>(qh[kk], ld16(grow_lo + kk * 32), op[rt]);
              }
            });
            const float gs = a.gram_scale[rt * 32 + j];
#pragma unroll
            for (int e = 0; e < 16; ++e) op[rt][e] *= gs;
          }
        });
        if constexpr (GLDS) __syncthreads();   // the fused apply re-uses the same LDS as its staging tile
      }
    }
    if constexpr (BETA == kKL || (C::TWO_ACC && !X3 && R_PAD <= 128)) {   // (rank pad 256 x two sets: no registers left)
      if (a.fuse_apply) {
        // ---- nmf.py:78-92 in the epilogue (the workgroup owns complete rows: nsplit == 1).  The new factor values
        // replace the accumulators, go to the fp32 master, to the transposed image (8-byte pieces straight from
        // registers) and, through a wave-private LDS tile, to the row-major image (16-byte pieces).  beta == 1: the
        // denominators are the panel's column sums (closed form); beta != 1: relu(den accumulator) + eps -- no slab round
        // trip, no apply launch (64 MiB of slab stores + a 160 MiB apply kernel less per W half-step at configs[1]).
        fused_done = true;
        constexpr int LDT = R_PAD;                     // wave tile [32][R_PAD] fp32 = 128*R_PAD bytes per wave
        float* tile = reinterpret_cast<float*>(smem) + wave * (32 * LDT);
        float den[RT], csum[RT];
        float unsc_f = 1.f;                            // fp16 scale of the elementwise terms (exact power of two)
        if constexpr (C::SCALE) unsc_f = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((127 - ki) << 23));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          den[rt] = BETA == kKL ? a.kl_den[rt * 32 + j] : 0.f;
          csum[rt] = 0.f;
        }
        const int mrow0 = mb * BM + wave * 32;  // first owner row of this wave
        // master loads first (independent, fully pipelined), then the dependent compute + stores -- in chunks of at
        // most four 32-wide rank tiles so that the staging array stays at 64 registers
        constexpr int RC = RT > 4 ? 4 : RT;
        // unregularised full tiles (the usual case) take a branch-free form: no bounds tests, no per-element uniform
        // branches, a reciprocal instead of the IEEE division (<= 1 ulp apart; as in the ping-pong kernel's epilogue)
        // (beta = 2 only: +2 % there; at beta = 0.5 / 0 the second copy of the epilogue cost the W half-step 5 % -- r3pl)
        const bool plain = BETA == kEuc && a.l1 <= 0.f && a.l2 <= 0.f && a.gamma == 1.f && a.rank == R_PAD && mrow0 + 32 <= a.M;
        auto update = [&](auto plain_c) {
        constexpr bool PL = decltype(plain_c)::value;
        float rden[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) rden[rt] = PL && !C::TWO_ACC ? 1.f / den[rt] : 0.f;
        static_for<RT / RC>([&](auto chunk) {
          constexpr int rt0 = decltype(chunk)::value * RC;
          float fold[RC][16];
          static_for<RC>([&](auto rcc) {
            constexpr int rc = decltype(rcc)::value;
            const int r = (rt0 + rc) * 32 + j;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = mrow0 + (e & 3) + 8 * (e >> 2) + 4 * hl;
              if constexpr (PL) fold[rc][e] = a.f[(size_t)row * R_PAD + r];
              else fold[rc][e] = (row < a.M && r < a.rank) ? a.f[(size_t)row * a.rank + r] : 0.f;
            }
          });
          static_for<RC>([&](auto rcc) {
            constexpr int rc = decltype(rcc)::value, rt = rt0 + rc;
            const int r = rt * 32 + j;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = mrow0 + (e & 3) + 8 * (e >> 2) + 4 * hl;
              float fv = fold[rc][e];
              if constexpr (PL) {
                const float neg = fmaxf(C::SCALE ? on[rt][e] * unsc_f : on[rt][e], 0.f) + kEps;
                if constexpr (C::TWO_ACC)
                  fv *= neg * __builtin_amdgcn_rcpf(fmaxf(C::SCALE ? op[rt][e] * unsc_f : op[rt][e], 0.f) + kEps);
                else
                  fv *= neg * rden[rt];
                a.f[(size_t)row * R_PAD + r] = fv;
              } else if (row < a.M && r < a.rank) {
                const float neg = fmaxf(C::SCALE ? on[rt][e] * unsc_f : on[rt][e], 0.f) + kEps;
                float pos = den[rt];
                if constexpr (C::TWO_ACC) pos = fmaxf(C::SCALE ? op[rt][e] * unsc_f : op[rt][e], 0.f) + kEps;
                if (a.l1 > 0.f) pos += a.l1;
                if (a.l2 > 0.f) pos += a.l2 * fv;
                float mult = neg / pos;
                if (a.gamma != 1.f) mult = mu_pow(mult, a.gamma);
                fv *= mult;
                a.f[(size_t)row * a.rank + r] = fv;
              }
              fold[rc][e] = fv;   // (not written back into the accumulator array)
              csum[rt] += fv;
              tile[((e & 3) + 8 * (e >> 2) + 4 * hl) * LDT + r] = fv;
            }
            // transposed image: 4 consecutive owner rows of column r = 8 bytes
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
              const float v0 = fold[rc][4 * q4], v1 = fold[rc][4 * q4 + 1], v2 = fold[rc][4 * q4 + 2],
                          v3 = fold[rc][4 * q4 + 3];
              const uint32_t h0 = pack_op<OPT>(v0, v1), h1 = pack_op<OPT>(v2, v3);
              const int64_t off = p2_offset(mrow0 + 8 * q4 + 4 * hl, r, R_PAD);
              *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.o2_hi) + off) = make_uint2(h0, h1);
              if constexpr (X3) {
                const uint32_t l0 = pack_bf16(v0 - bf16_lo(h0), v1 - bf16_hi(h0));
                const uint32_t l1 = pack_bf16(v2 - bf16_lo(h1), v3 - bf16_hi(h1));
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.o2_lo) + off) = make_uint2(l0, l1);
              }
            }
          });
        });
        };
        if constexpr (BETA == kEuc) {
          if (plain) update(std::integral_constant<bool, true>{});
          else update(std::integral_constant<bool, false>{});
        } else {
          update(std::integral_constant<bool, false>{});
        }
        __syncthreads();
        // row-major image from the LDS tile: 32 rows x R_PAD/8 sixteen-byte slots per wave
        constexpr int SP = R_PAD / 8;
        bool clamped = false;
#pragma unroll
        for (int i = 0; i < (32 * SP) / 64; ++i) {
          const int chunk = i * 64 + lane, rl = chunk / SP, slot = chunk % SP;
          const float* src = tile + rl * LDT + slot * 8;
          u32x4 hi, lo;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x0 = src[2 * q], x1 = src[2 * q + 1];
            if constexpr (C::F16) clamped |= fmaxf(x0, x1) > 65504.f;
            const uint32_t h = pack_op<OPT>(x0, x1);
            hi[q] = h;
            if constexpr (X3) lo[q] = pack_bf16(x0 - bf16_lo(h), x1 - bf16_hi(h));
          }
          const int64_t off = p1_offset(mrow0 + rl, slot * 8, R_PAD);
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o1_hi) + off) = hi;
          if constexpr (X3) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o1_lo) + off) = lo;
        }
        if constexpr (C::F16) {
          if (a.status && __any(clamped) && lane == 0) atomicOr(a.status, 1u);
        }
        __syncthreads();
        // partial column sums of this workgroup's rows: lane halves, then the waves (fixed order)
        float* red = reinterpret_cast<float*>(smem);  // [WAVES][R_PAD]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float tot = csum[rt] + __shfl_xor(csum[rt], 32, 64);
          if (hl == 0) red[wave * R_PAD + rt * 32 + j] = tot;
        }
        __syncthreads();
        for (int r = tid; r < R_PAD; r += C::THREADS)
          a.colsum_part[(size_t)mb * R_PAD + r] = (red[r] + red[R_PAD + r]) + (red[2 * R_PAD + r] + red[3 * R_PAD + r]);
      }
    }
    if (!fused_done) {
      // (fp16 scale: back by 2^-ki, exact; the factor sits in a scalar register)
      const float unsc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((127 - ki) << 23));
      const size_t slab = ((size_t)ks * a.M_pad + (size_t)mb * BM + wave * 32) * R_PAD;
      static_for<RT>([&](auto rtc) {
        constexpr int rt = decltype(rtc)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * hl;
          const size_t idx = slab + (size_t)row * R_PAD + rt * 32 + j;
          a.slab_num[idx] = C::SCALE ? on[rt][e] * unsc : on[rt][e];
          if constexpr (C::TWO_ACC && !C::XB) a.slab_den[idx] = C::SCALE ? op[rt][e] * unsc : op[rt][e];
          if constexpr (C::XB) {   // this workgroup's share of the ONE denominator slab (see den_split above)
            const int nd = a.nsplit < RT ? a.nsplit : RT;
            if (ks < nd && rt % nd == ks && a.slab_den && a.gram_hi)
              a.slab_den[((size_t)mb * BM + wave * 32 + row) * R_PAD + rt * 32 + j] = op[rt][e];
          }
        }
      });
    }
  }
#ifdef NMFMU_DEBUG_HOOKS
  if constexpr (C::XB) {
    if (a.debug) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the epilogue's stores have left the CU
    stamp(3);
  }
#endif
}

// Host-side launcher, one per (R_PAD) translation unit.  prec = NMFMU_PREC_*, beta_kind = BetaKind.
int launch_fused_r32(int beta_kind, int prec, int mode, const FusedArgs& a, int grid, hipStream_t s);
int launch_fused_r64(int beta_kind, int prec, int mode, const FusedArgs& a, int grid, hipStream_t s);
int launch_fused_r128(int beta_kind, int prec, int mode, const FusedArgs& a, int grid, hipStream_t s);
int launch_fused_r256(int beta_kind, int prec, int mode, const FusedArgs& a, int grid, hipStream_t s);

// Per-device "attribute set" memo (one host thread may drive several devices)
inline bool* attr_flag(bool (&flags)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  return &flags[dev];
}

template <int R_PAD, int BETA, int PREC, int MODE>
int launch_one(const FusedArgs& a, int grid, hipStream_t s) {
  using C = FusedCfg<R_PAD, BETA, PREC, MODE>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
  auto kern = fused_kernel<R_PAD, BETA, PREC, MODE>;
  static bool done[64] = {};   // the dynamic-LDS attribute is per device: a single-process multi-device host sets it on each
  bool* flag = attr_flag(done);
  if (!*flag) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       C::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    *flag = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), C::LDS_BYTES, s, a);
  return (int)hipGetLastError();
}

// Instantiation table.  beta == 1 with one operand plane at padded rank <= 128 belongs to the ping-pong kernel
// (nmfmu_pp.h); its four-wave form stays for 128-row tiles (W half-steps that want two workgroups per CU, tests).
// The loss and the denominator-only pass use the generic branch for beta = 0.5 / 1.5.
template <int R_PAD, bool ALLOW_X3>
int launch_fused_dispatch(int beta_kind, int prec, int mode, const FusedArgs& a, int grid, hipStream_t s) {
  if (mode != kModeMU && (beta_kind == kSqrt || beta_kind == kSqrt3)) beta_kind = kGen;
#define NMFMU_CASE(B, P, M) \
  if (beta_kind == B && prec == P && mode == M) return launch_one<R_PAD, B, P, M>(a, grid, s);
#define NMFMU_CASE_MU(P) \
  NMFMU_CASE(kKL, P, kModeMU) NMFMU_CASE(kEuc, P, kModeMU) NMFMU_CASE(kIS, P, kModeMU) NMFMU_CASE(kGen, P, kModeMU) \
  NMFMU_CASE(kSqrt, P, kModeMU) NMFMU_CASE(kSqrt3, P, kModeMU)
#define NMFMU_CASE_LOSS(P) \
  NMFMU_CASE(kKL, P, kModeLoss) NMFMU_CASE(kEuc, P, kModeLoss) NMFMU_CASE(kIS, P, kModeLoss) NMFMU_CASE(kGen, P, kModeLoss)
  NMFMU_CASE_MU(kPrecBf16)
  NMFMU_CASE_LOSS(kPrecBf16)
  NMFMU_CASE_MU(kPrecF16)
  NMFMU_CASE_LOSS(kPrecF16)
  NMFMU_CASE_MU(kPrecF16X)
  NMFMU_CASE_LOSS(kPrecF16X)
  // f16r (3-byte target): every beta but 2 (whose target is an MFMA operand: 'f16x')
  NMFMU_CASE(kKL, kPrecF16R, kModeMU) NMFMU_CASE(kIS, kPrecF16R, kModeMU) NMFMU_CASE(kGen, kPrecF16R, kModeMU)
  NMFMU_CASE(kSqrt, kPrecF16R, kModeMU) NMFMU_CASE(kSqrt3, kPrecF16R, kModeMU)
  NMFMU_CASE(kKL, kPrecF16R, kModeLoss) NMFMU_CASE(kIS, kPrecF16R, kModeLoss) NMFMU_CASE(kGen, kPrecF16R, kModeLoss)
  NMFMU_CASE(kGen, kPrecBf16, kModeDen)
  NMFMU_CASE(kKL, kPrecBf16, kModeMU2) NMFMU_CASE(kKL, kPrecF16, kModeMU2)
  NMFMU_CASE(kEuc, kPrecBf16, kModeXB) NMFMU_CASE(kEuc, kPrecF16, kModeXB)
  if constexpr (R_PAD <= 128) {   // (fp32 X buffers + rank-256 accumulators do not fit the 256 architectural VGPRs)
    NMFMU_CASE(kEuc, kPrecF16X, kModeXB)
  }
  if constexpr (ALLOW_X3) {
    NMFMU_CASE(kGen, kPrecX3, kModeDen)
    NMFMU_CASE_MU(kPrecX3)
    NMFMU_CASE_LOSS(kPrecX3)
  }
#undef NMFMU_CASE_LOSS
#undef NMFMU_CASE_MU
#undef NMFMU_CASE
  return -2;  // unsupported combination
}

}  // namespace nmfmu
