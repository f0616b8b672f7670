// Fused beta-divergence MU half-step for gfx950 (MI355X, CDNA4).
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
// the S^T tile is produced by v_mfma_f32_32x32x16_bf16 with the panel as the
// A operand, so that each lane ends up holding one owner row and 16
// consecutive contraction columns -- exactly the A-operand layout of the
// second MFMA (contraction over k).  The trick that makes the columns
// consecutive is a row permutation pi of the panel tile (see a_row[] below);
// with it the X tile can be stored in HBM in "fragment order" (nmfmu_layout.h)
// and loaded straight into the right lanes with fully coalesced 16-byte loads.
//
// Precision modes
//   bf16   : A, B and Gn/Gp rounded to bf16, X stored bf16, fp32 accumulate.
//   bf16x3 : every bf16 operand is a (hi, lo) pair and every product is
//            hi*hi + lo*hi + hi*lo (3 MFMAs); X stored fp32.  ~2^-16 relative
//            operand error -- this is the mode that meets the 1e-4 parity bar
//            on the factors.
//
// Workgroup = 4 waves (one per SIMD), wave w owns rows 32w..32w+31 of the
// block.  The panel tile (64 rows of B, both images) is double-buffered in
// LDS, filled either by LDS-DMA (global_load_lds, STAGE = 1) or through
// registers (STAGE = 0); the X tile is prefetched one tile ahead in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "nmfmu_layout.h"

#ifndef NMFMU_PIN_SCHED
#define NMFMU_PIN_SCHED 1
#endif
#ifndef NMFMU_SP_FENCE
#define NMFMU_SP_FENCE 1
#endif
#ifndef NMFMU_SP
#define NMFMU_SP 1  // eight-wave cross-tile software pipelining for beta == 1 / bf16 / rank pad 128 (256-row tiles): +2-6 %
#endif
#ifndef NMFMU_X_NT
#define NMFMU_X_NT 1  // non-temporal loads for the X stream (read once; keeps the factor panel resident in L2)
#endif
#ifndef NMFMU_DMA_ASM
#define NMFMU_DMA_ASM 1  // issue the LDS-DMA from inline asm (keeps hipcc's counted lgkmcnt waits)
#endif

namespace nmfmu {

constexpr float kEps = 1.1920928955078125e-07f;  // constants.py:3 of the reference

enum BetaKind : int { kKL = 0, kEuc = 1, kIS = 2, kGen = 3 };
// kModeDen: the positive term alone, den = Gp(S) @ panel with no target at all (sparse targets with a generic beta:
// the reference's dense pass of nmf.py:628-636).  The loss mode also runs without a target (xp == nullptr: X = 0).
enum FusedMode : int { kModeMU = 0, kModeLoss = 1, kModeDen = 2 };

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
  const void* xp;         // fragment-order X tiles (bf16 or fp32)
  const uint16_t* p1_hi;  // panel, row-major image
  const uint16_t* p1_lo;
  const uint16_t* p2_hi;  // panel, transposed tiles
  const uint16_t* p2_lo;
  const uint16_t* a1_hi;  // owner, row-major image
  const uint16_t* a1_lo;
  float* slab_num;        // [nsplit][M_pad][R_PAD]
  float* slab_den;        // same, beta != 1
  float* loss_part;       // [gridDim.x], loss mode
  int M, K;               // logical sizes (loss masking only)
  int M_pad, ktiles, nsplit, tiles_per_split;
  float beta;
  // fused apply (beta == 1, nsplit == 1): the epilogue performs nmf.py:78-92 and re-emits the owner's images
  int fuse_apply;
  int rank;               // logical rank (row pitch of f)
  float* f;               // owner fp32 master [M][rank]
  const float* kl_den;    // [R_PAD] column sums of the panel
  const float* kl_part;   // or: [kl_nparts][R_PAD] partial column sums, reduced by every workgroup itself (nmfmu_pp.h)
  int kl_nparts;
  uint16_t* o1_hi;        // owner images to refresh (same buffers a1_* were read from)
  uint16_t* o1_lo;
  uint16_t* o2_hi;
  uint16_t* o2_lo;
  float* colsum_part;     // [M_pad / BM][R_PAD]
  float l1, l2, gamma;
};

// G = 32-row groups per wave: G = 1 -> 128-row workgroup tile, 2 waves/SIMD; G = 2 -> 256-row tile, every LDS
// operand read feeds two MFMAs and the wave has the whole 512-register file (one wave per SIMD).
template <int R_PAD, int BETA, bool X3, int MODE, int G = 1>
struct FusedCfg {
  static constexpr int BM = 128 * G;
  // NMFMU_SP: the 256-row instance of the beta == 1 / bf16 / rank-pad-128 MU kernel runs as EIGHT waves of one
  // 32-row group each (two waves per SIMD, <= 256 VGPRs) with the cross-tile software-pipelined main loop, instead
  // of four waves of two groups.  Same X packing (wave w' = 2 w + g owns the same rows), same epilogue.
  static constexpr bool SP = NMFMU_SP && R_PAD == 128 && G == 2 && BETA == kKL && !X3 && MODE == kModeMU;
  static constexpr int WAVES = SP ? 8 : 4;
  static constexpr int GW = SP ? 1 : G;      // 32-row groups per wave
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int KS = R_PAD / 16;      // k-steps of GEMM1 (contraction over rank)
  static constexpr int RT = R_PAD / 32;      // 32-wide rank tiles of GEMM2's output
  static constexpr int ROWB = 2 * R_PAD;     // bytes per P1 row
  static constexpr int IMG = kBK * ROWB;     // bytes of one image tile (= 128 * R_PAD)
  static constexpr int NPL = X3 ? 2 : 1;     // planes (hi / lo)
  static constexpr bool LOSS = MODE == kModeLoss;
  static constexpr bool DEN = MODE == kModeDen;
  static constexpr int NIMG = (LOSS ? 1 : 2) * NPL;
  static constexpr int P1HI = 0;
  static constexpr int P1LO = IMG;           // valid when X3
  static constexpr int P2HI = NPL * IMG;
  static constexpr int P2LO = NPL * IMG + IMG;
  static constexpr int STAGE_BYTES = NIMG * IMG;
  // software-pipelined path: 3-slot rings for P1, P2 and the X tile (bf16, 128 rows x 64 columns)
  static constexpr int XTILE = BM * kBK * 2;
  // pipelined path: two-slot rings for P1 and P2 (one tile of lead), three-slot ring for X (two tiles of lead)
  static constexpr int LDS_BYTES = SP ? 2 * 2 * IMG + 3 * XTILE : 2 * STAGE_BYTES;
  static constexpr int NQ = X3 ? 8 : 4;      // 16-byte X chunks per lane per tile
  static constexpr bool TWO_ACC = (BETA != kKL) && !LOSS && !DEN;
  static constexpr int PASSES = IMG / 4096;  // 256 threads x 16 B per pass
  // the software-pipelined beta == 1 kernel keeps two S tiles live: it gets the whole register file (one wave per SIMD)
  static constexpr int MINW = (X3 || TWO_ACC || R_PAD > 128 || G > 1) ? 1 : 2;   // (SP: 512 threads, 1 workgroup)
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

// ---- operand element type of the single-plane kernels (nmfmu_pp.h, nmfmu_gemm.h): bf16, or fp16 (11 significant bits
// at the same MFMA rate; values clamped to 65504 when packed, conversions saturate under MODE.FP16_OVFL)
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
  } else {
    const float lg = __builtin_amdgcn_logf(s);  // log2
    gp = __builtin_amdgcn_exp2f((beta - 1.f) * lg);
    gn = gp * __builtin_amdgcn_rcpf(s) * x;
  }
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

template <int R_PAD, int BETA, bool X3, int MODE, int STAGE, int GT>
__global__ void __launch_bounds__((FusedCfg<R_PAD, BETA, X3, MODE, GT>::THREADS), (FusedCfg<R_PAD, BETA, X3, MODE, GT>::MINW))
    fused_kernel(const FusedArgs a) {
  using C = FusedCfg<R_PAD, BETA, X3, MODE, GT>;
  constexpr int G = C::GW;   // 32-row groups per wave
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
  const int m0 = mb * BM + wave * (32 * G) + j;  // row of group g is m0 + 32 * g

  // ---- owner fragments (B operand of GEMM1): row m, rank slice 16*kk + 8*hl .. +7
  u32x4 qh[G][KS];
  u32x4 ql[G][X3 ? KS : 1];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int m = m0 + 32 * g;
    const int sw = P1Swz<R_PAD>::of(m) << 4;
    const char* rowh = reinterpret_cast<const char*>(a.a1_hi) + (size_t)m * ROWB;
    const char* rowl = reinterpret_cast<const char*>(a.a1_lo) + (size_t)m * ROWB;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int off = (kk * 32 + hl * 16) ^ sw;
      qh[g][kk] = ld16(rowh + off);
      if constexpr (X3) ql[g][kk] = ld16(rowl + off);
    }
  }

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

  f32x16 on[G][C::LOSS ? 1 : RT];
  f32x16 op[G][C::TWO_ACC ? RT : 1];
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int rt = 0; rt < (C::LOSS ? 1 : RT); ++rt)
#pragma unroll
      for (int e = 0; e < 16; ++e) on[g][rt][e] = 0.f;
#pragma unroll
    for (int rt = 0; rt < (C::TWO_ACC ? RT : 1); ++rt)
#pragma unroll
      for (int e = 0; e < 16; ++e) op[g][rt][e] = 0.f;
  }
  float lacc = 0.f;

  const char* xbase =
      reinterpret_cast<const char*>(a.xp) + ((size_t)mb * a.ktiles * 4 + wave) * (G * NQ * 1024) + lane * 16;
  const bool no_x = C::DEN || a.xp == nullptr;   // (uniform) no target: the X registers stay zero
  auto load_x = [&](int t, u32x4(&x)[G][NQ]) {
    const char* p = xbase + (size_t)t * (4 * G * NQ * 1024);
    if (no_x) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int q = 0; q < NQ; ++q) x[g][q] = u32x4{0u, 0u, 0u, 0u};
      return;
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#if NMFMU_X_NT
        x[g][q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (g * NQ + q) * 1024));
#else
        x[g][q] = ld16(p + (g * NQ + q) * 1024);
#endif
      }
  };

  // ---- panel staging: every image tile is one contiguous, pre-swizzled block in HBM
  const char* img_src[C::NIMG];
  img_src[0] = reinterpret_cast<const char*>(a.p1_hi);
  if constexpr (X3) img_src[1] = reinterpret_cast<const char*>(a.p1_lo);
  if constexpr (!C::LOSS) {
    img_src[C::NPL] = reinterpret_cast<const char*>(a.p2_hi);
    if constexpr (X3) img_src[C::NPL + 1] = reinterpret_cast<const char*>(a.p2_lo);
  }
  u32x4 st[STAGE == 0 ? C::NIMG * C::PASSES : 1];
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
  const unsigned wave_lds = (unsigned)wave * 1024u;
  auto stage_issue = [&](int t, int buf) {
#pragma unroll
    for (int im = 0; im < C::NIMG; ++im) {
      const char* src = img_src[im] + (size_t)t * IMG + tid * 16;
#pragma unroll
      for (int p = 0; p < C::PASSES; ++p) {
        if constexpr (STAGE == 1) {
          char* dst = smem + buf * C::STAGE_BYTES + im * IMG + p * 4096 + wave * 1024;  // wave-uniform base
#if NMFMU_DMA_ASM
          // Issued from inline asm on purpose: while hipcc knows an LDS-DMA is in flight it turns every LDS wait
          // into lgkmcnt(0), which serialises the operand prefetch rings.  Completion is waited for explicitly
          // (vmcnt(0) before the tile's barrier, see the main loop).  M0 = LDS byte address of the wave's 1 KiB
          // piece (integer arithmetic on the workgroup's LDS base; M0 is clobbered, nothing else uses it here).
          (void)dst;
          const unsigned lds_addr = lds_base + (unsigned)(buf * C::STAGE_BYTES + im * IMG + p * 4096) + wave_lds;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                       :
                       : "v"(src + p * 4096), "s"(lds_addr)
                       : "memory", "m0");
#else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 4096),
                                           (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
#endif
        } else {
          st[im * C::PASSES + p] = ld16(src + p * 4096);
        }
      }
    }
  };
  auto stage_commit = [&](int buf) {
    if constexpr (STAGE == 0) {
#pragma unroll
      for (int im = 0; im < C::NIMG; ++im)
#pragma unroll
        for (int p = 0; p < C::PASSES; ++p)
          *reinterpret_cast<u32x4*>(smem + buf * C::STAGE_BYTES + im * IMG + p * 4096 + tid * 16) = st[im * C::PASSES + p];
    }
  };

  f32x16 epsv;
#pragma unroll
  for (int e = 0; e < 16; ++e) epsv[e] = (BETA == kEuc) ? 0.f : kEps;

  auto compute = [&](int t, int buf, u32x4(&x)[G][NQ], int t_next) {
    const char* sb = smem + buf * C::STAGE_BYTES;
    // ---------------- GEMM1: S^T tiles (panel rows x owner rows), contraction over rank.
    // The panel operands are fetched through a PF-deep register ring so that PF-1 ds_read_b128 are always in
    // flight behind the MFMA that is issuing (hipcc otherwise emits read -> lgkmcnt(0) -> mfma, one at a time).
    // The accumulators are seeded with eps through the C operand of each chain's first MFMA (seed tile `epsv`,
    // loop invariant) instead of being re-initialised with 16 moves per tile.
    f32x16 s[G][2];
    {
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
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if constexpr (X3) {
            s[g][tt] = mfma_bf16(al, qh[g][kk], kk == 0 ? epsv : s[g][tt]);
            s[g][tt] = mfma_bf16(ah, ql[g][kk], s[g][tt]);
            s[g][tt] = mfma_bf16(ah, qh[g][kk], s[g][tt]);
          } else {
            s[g][tt] = mfma_bf16(ah, qh[g][kk], kk == 0 ? epsv : s[g][tt]);
          }
        }
      }
#if NMFMU_PIN_SCHED
      // pin the software pipeline: PF reads up front, then one read behind every MFMA group
      __builtin_amdgcn_sched_group_barrier(0x100, PF * C::NPL, 0);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        __builtin_amdgcn_sched_group_barrier(0x008, G * (X3 ? 3 : 1), 0);
        if (step + PF < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, C::NPL, 0);
      }
#endif
    }
    // ---------------- elementwise: Gn / Gp (or the loss terms), packed to bf16 A operands
    uint32_t gnh[G][2][8], gnl[G][X3 ? 2 : 1][8], gph[G][C::TWO_ACC ? 2 : 1][8], gpl[G][(C::TWO_ACC && X3) ? 2 : 1][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          float x0, x1;
          if constexpr (X3) {
            // NB: extract to scalars first -- __builtin_bit_cast on an ext-vector ELEMENT lvalue reads element 0
            // (hipcc 7.2), which silently turned every 16-byte chunk into a splat of its first float.
            const uint32_t u0 = x[g][4 * tt + (d >> 1)][2 * (d & 1)];
            const uint32_t u1 = x[g][4 * tt + (d >> 1)][2 * (d & 1) + 1];
            x0 = __builtin_bit_cast(float, u0);
            x1 = __builtin_bit_cast(float, u1);
          } else {
            const uint32_t w = x[g][2 * tt + (d >> 2)][d & 3];
            x0 = bf16_lo(w);
            x1 = bf16_hi(w);
          }
          const float s0 = s[g][tt][2 * d], s1 = s[g][tt][2 * d + 1];
          if constexpr (C::LOSS) {
            const int k0 = t * kBK + 32 * hl + 16 * tt + 2 * d;
            const bool rowok = m0 + 32 * g < a.M;
            lacc += (rowok && k0 < a.K) ? loss_elem<BETA>(s0, x0, a.beta) : 0.f;
            lacc += (rowok && k0 + 1 < a.K) ? loss_elem<BETA>(s1, x1, a.beta) : 0.f;
          } else {
            float n0, n1, p0, p1;
            mu_elem<BETA>(s0, x0, a.beta, n0, p0);
            mu_elem<BETA>(s1, x1, a.beta, n1, p1);
            if constexpr (C::DEN) n0 = p0, n1 = p1;   // denominator-only pass: the one operand set carries Gp
            const uint32_t nh = pack_bf16(n0, n1);
            gnh[g][tt][d] = nh;
            if constexpr (X3) gnl[g][tt][d] = pack_bf16(n0 - bf16_lo(nh), n1 - bf16_hi(nh));
            if constexpr (C::TWO_ACC) {
              const uint32_t ph = pack_bf16(p0, p1);
              gph[g][tt][d] = ph;
              if constexpr (X3) gpl[g][tt][d] = pack_bf16(p0 - bf16_lo(ph), p1 - bf16_hi(ph));
            }
          }
        }
      }
    }
    // X's registers are dead from here on: fetch the next tile into them now (single X buffer; the loads have the
    // whole GEMM2 + barrier + next GEMM1 to land).
    if (t_next >= 0) load_x(t_next, x);
    // ---------------- GEMM2: num/den (owner rows x rank), contraction over the tile's 64 columns
    if constexpr (!C::LOSS) {
      constexpr int NSTEP = RT * 4;
      constexpr int PF = 4;
      u32x4 ring_h[PF];
      u32x4 ring_l[X3 ? PF : 1];
      // step -> ((tt, m2), rt): rank tiles innermost, so consecutive MFMAs cycle through the RT accumulators
      auto b_offs = [&](int step) {
        const int rt = step % RT, c = step / RT;
        return rt * 4096 + b_row + b_off[c >> 1][c & 1];
      };
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        ring_h[p] = ld16(sb + C::P2HI + b_offs(p));
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
          ring_h[step % PF] = ld16(sb + C::P2HI + b_offs(step + PF));
          if constexpr (X3) ring_l[step % PF] = ld16(sb + C::P2LO + b_offs(step + PF));
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const u32x4 nh = {gnh[g][tt][4 * m2], gnh[g][tt][4 * m2 + 1], gnh[g][tt][4 * m2 + 2],
                            gnh[g][tt][4 * m2 + 3]};
          if constexpr (X3) {
            const u32x4 nl = {gnl[g][tt][4 * m2], gnl[g][tt][4 * m2 + 1], gnl[g][tt][4 * m2 + 2],
                              gnl[g][tt][4 * m2 + 3]};
            on[g][rt] = mfma_bf16(nl, bh, on[g][rt]);
            on[g][rt] = mfma_bf16(nh, bl, on[g][rt]);
          }
          on[g][rt] = mfma_bf16(nh, bh, on[g][rt]);
          if constexpr (C::TWO_ACC) {
            const u32x4 ph = {gph[g][tt][4 * m2], gph[g][tt][4 * m2 + 1], gph[g][tt][4 * m2 + 2],
                              gph[g][tt][4 * m2 + 3]};
            if constexpr (X3) {
              const u32x4 pl = {gpl[g][tt][4 * m2], gpl[g][tt][4 * m2 + 1], gpl[g][tt][4 * m2 + 2],
                                gpl[g][tt][4 * m2 + 3]};
              op[g][rt] = mfma_bf16(pl, bh, op[g][rt]);
              op[g][rt] = mfma_bf16(ph, bl, op[g][rt]);
            }
            op[g][rt] = mfma_bf16(ph, bh, op[g][rt]);
          }
        }
      }
#if NMFMU_PIN_SCHED
      __builtin_amdgcn_sched_group_barrier(0x100, PF * C::NPL, 1);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        __builtin_amdgcn_sched_group_barrier(0x008, G * (X3 ? 3 : 1) * (C::TWO_ACC ? 2 : 1), 1);
        if (step + PF < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, C::NPL, 1);
      }
#endif
    }
  };

  constexpr bool kSP = C::SP;
  if constexpr (kSP) {
    // ---------------- cross-tile software-pipelined main loop, eight waves (two per SIMD).
    // Tile i's elementwise stage is interleaved step by step with GEMM1 of tile i+1 (the S tile ping-pongs between
    // two register sets), then GEMM2(i) runs.  LDS: P1 ring [2][IMG] | P2 ring [2][IMG] | X ring [3][XTILE] = 160 KiB.
    // Everything arrives by LDS-DMA issued from inline asm (hipcc must not see it); every tile ends with ONE counted
    // vmcnt + raw s_barrier that lets exactly this tile's X prefetch stay in flight.
    // Invariant at the start of tile i: P1(i+1), P2(i), X(i) have landed, X(i+1) may be in flight;
    // tile i issues P1(i+2), P2(i+1) (needed by the end of this tile) and then X(i+2).
    static_assert(G == 1 && NQ == 4 && IMG == 16384, "pipelined path: rank pad 128, bf16");
    if (t0 < t1) {
      const int nt = t1 - t0;
      constexpr int PF = 4;
      constexpr int XT = C::XTILE;                  // 32 KiB: 256 rows x 64 columns bf16
      constexpr int PASS = C::THREADS * 16;         // bytes one DMA instruction of the whole workgroup moves (8 KiB)
      constexpr int NP = IMG / PASS, NX = XT / PASS;   // 2 and 4 DMA instructions per thread
      constexpr unsigned P1_BASE = 0, P2_BASE = 2 * IMG, X_BASE = 4 * IMG;
      const char* xtile0 = reinterpret_cast<const char*>(a.xp) + (size_t)mb * a.ktiles * XT + tid * 16;
      auto dma = [&](const char* src, unsigned lds_off, auto nc) {
#pragma unroll
        for (int p = 0; p < decltype(nc)::value; ++p) {
          const unsigned lds_addr = lds_base + lds_off + (unsigned)(p * PASS) + wave_lds;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                       :
                       : "v"(src + p * PASS), "s"(lds_addr)
                       : "memory", "m0");
        }
      };
      using NPc = std::integral_constant<int, NP>;
      using NXc = std::integral_constant<int, NX>;
      auto issue_p1 = [&](int i, unsigned slot) { dma(img_src[0] + (size_t)(t0 + i) * IMG + tid * 16, P1_BASE + slot * IMG, NPc{}); };
      auto issue_p2 = [&](int i, unsigned slot) { dma(img_src[1] + (size_t)(t0 + i) * IMG + tid * 16, P2_BASE + slot * IMG, NPc{}); };
      auto issue_x = [&](int i, unsigned slot) { dma(xtile0 + (size_t)(t0 + i) * XT, X_BASE + slot * XT, NXc{}); };
      auto a_off = [&](int step) { return a_row[step & 1] + (((step >> 1) * 32 + hl * 16) ^ a_sw[step & 1]); };
      auto b_offs = [&](int step) {
        const int rt = step % RT, c = step / RT;
        return rt * 4096 + b_row + b_off[c >> 1][c & 1];
      };
      auto tile_end = [&](bool x_in_flight) {
        if (x_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NX) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      uint32_t gn[2][8];
      f32x16 sA[2], sB[2];
      // ---- prologue: tile 0's data and P1(1); X(1) may stay in flight
      issue_p1(0, 0);
      issue_p2(0, 0);
      issue_x(0, 0);
      if (nt > 1) issue_p1(1, 1);
      if (nt > 1) issue_x(1, 1);
      tile_end(nt > 1);
      {
        const char* sb = smem + P1_BASE;
        u32x4 ring[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) ring[p] = ld16(sb + a_off(p));
#pragma unroll
        for (int step = 0; step < 2 * KS; ++step) {
          const u32x4 ah = ring[step % PF];
          if (step + PF < 2 * KS) ring[step % PF] = ld16(sb + a_off(step + PF));
          sA[step & 1] = mfma_bf16(ah, qh[0][step >> 1], (step >> 1) == 0 ? epsv : sA[step & 1]);
        }
      }
      __builtin_amdgcn_s_barrier();           // every wave is done with P1 slot 0 before tile 0 re-fills it
      unsigned xs_cur = 0;                    // X ring slot of tile i (i % 3)
      auto tile_body = [&](int i, auto parc, auto nextc, f32x16(&sc)[2], f32x16(&sn)[2]) {
        constexpr int par = decltype(parc)::value;            // i & 1
        constexpr bool has_next = decltype(nextc)::value;     // tile i+1 exists
        const unsigned xs1 = xs_cur == 2 ? 0 : xs_cur + 1, xs2 = xs1 == 2 ? 0 : xs1 + 1;
        const bool xf = i + 2 < nt;
        if (xf) issue_p1(i + 2, par);           // slot of P1(i): GEMM1(i) ran during tile i-1
        if (has_next) issue_p2(i + 1, par ^ 1); // slot of P2(i-1)
        if (xf) issue_x(i + 2, xs2);            // slot of X(i-1)
        // X(i): this lane's four 16-byte chunks from the ring
        u32x4 x[NQ];
        {
          const char* xs = smem + X_BASE + xs_cur * XT + wave * 4096 + lane * 16;
#pragma unroll
          for (int q = 0; q < NQ; ++q) x[q] = ld16(xs + q * 1024);
        }
        // ---- phase A: GEMM1(i+1) beside the elementwise stage of tile i
        {
          const char* sb = smem + P1_BASE + (par ^ 1) * IMG;
          u32x4 ring[PF];
          if constexpr (has_next) {
#pragma unroll
            for (int p = 0; p < PF; ++p) ring[p] = ld16(sb + a_off(p));
          }
#pragma unroll
          for (int step = 0; step < 16; ++step) {
            if constexpr (has_next) {
              const u32x4 ah = ring[step % PF];
              if (step + PF < 2 * KS) ring[step % PF] = ld16(sb + a_off(step + PF));
              sn[step & 1] = mfma_bf16(ah, qh[0][step >> 1], (step >> 1) == 0 ? epsv : sn[step & 1]);
            }
            const int tt = step >> 3, d = step & 7;
            const uint32_t w = x[2 * tt + (d >> 2)][d & 3];
            const float n0 = bf16_lo(w) * __builtin_amdgcn_rcpf(sc[tt][2 * d]);
            const float n1 = bf16_hi(w) * __builtin_amdgcn_rcpf(sc[tt][2 * d + 1]);
            gn[tt][d] = pack_bf16(n0, n1);
#if NMFMU_SP_FENCE
            __builtin_amdgcn_sched_barrier(0);   // keep the ring depth and the per-step interleave as written
#endif
          }
        }
        // ---- phase B: GEMM2(i)
        {
          const char* sb = smem + P2_BASE + par * IMG;
          u32x4 ring[PF];
#pragma unroll
          for (int p = 0; p < PF; ++p) ring[p] = ld16(sb + b_offs(p));
#pragma unroll
          for (int step = 0; step < 4 * RT; ++step) {
            const int rt = step % RT, c = step / RT, tt = c >> 1, m2 = c & 1;
            const u32x4 bh = ring[step % PF];
            if (step + PF < 4 * RT) ring[step % PF] = ld16(sb + b_offs(step + PF));
            const u32x4 nh = {gn[tt][4 * m2], gn[tt][4 * m2 + 1], gn[tt][4 * m2 + 2], gn[tt][4 * m2 + 3]};
            on[0][rt] = mfma_bf16(nh, bh, on[0][rt]);
#if NMFMU_SP_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
        }
        tile_end(xf);
        xs_cur = xs1;
      };
      using P0 = std::integral_constant<int, 0>;
      using P1c = std::integral_constant<int, 1>;
      int i = 0;
      for (; i + 2 < nt; i += 2) {   // both tiles of the pair have a successor
        tile_body(i, P0{}, std::true_type{}, sA, sB);
        tile_body(i + 1, P1c{}, std::true_type{}, sB, sA);
      }
      if (i == nt - 1) {
        tile_body(i, P0{}, std::false_type{}, sA, sB);
      } else {                       // i == nt - 2
        tile_body(i, P0{}, std::true_type{}, sA, sB);
        tile_body(i + 1, P1c{}, std::false_type{}, sB, sA);
      }
      __syncthreads();               // LDS is reused by the epilogue
    }
  }
  // ---------------- default main loop: LDS double buffer for the panel, X in ONE register buffer that compute()
  // refills with the next tile right after its last use; one drain + barrier per tile.
  if constexpr (!kSP)
  if (t0 < t1) {
    u32x4 xc[G][NQ];
    const int nt = t1 - t0;
    stage_issue(t0, 0);
    load_x(t0, xc);
    stage_commit(0);
    if constexpr (STAGE == 1) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // vmcnt(0): the asm DMA landed
    __syncthreads();
    for (int i = 0; i < nt; ++i) {
      const int t = t0 + i, buf = i & 1;
      const bool more = i + 1 < nt;
      if (more) stage_issue(t + 1, buf ^ 1);
      compute(t, buf, xc, more ? t + 1 : -1);
      if (more) stage_commit(buf ^ 1);
      if constexpr (STAGE == 1) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
      __syncthreads();
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
    if constexpr (BETA == kKL) {
      if (a.fuse_apply) {
        // ---- nmf.py:78-92 in the epilogue (the workgroup owns complete rows: nsplit == 1).  The new factor values
        // replace the accumulators, go to the fp32 master, to the transposed image (8-byte pieces straight from
        // registers) and, through a wave-private LDS tile, to the row-major image (16-byte pieces).
        fused_done = true;
        constexpr int LDT = R_PAD;                     // wave tile [32][R_PAD] fp32 = 128*R_PAD bytes per wave
        float* tile = reinterpret_cast<float*>(smem) + wave * (32 * LDT);
        float den[RT], csum[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          den[rt] = a.kl_den[rt * 32 + j];
          csum[rt] = 0.f;
        }
        auto apply_group = [&](auto gc) {
          constexpr int g = decltype(gc)::value;
          const int mrow0 = mb * BM + wave * (32 * G) + 32 * g;  // first owner row of this group
          // master loads first (independent, fully pipelined), then the dependent compute + stores -- in chunks of at
          // most four 32-wide rank tiles so that the staging array stays at 64 registers
          constexpr int RC = RT > 4 ? 4 : RT;
          static_for<RT / RC>([&](auto chunk) {
            constexpr int rt0 = decltype(chunk)::value * RC;
            float fold[RC][16];
            static_for<RC>([&](auto rcc) {
              constexpr int rc = decltype(rcc)::value;
              const int r = (rt0 + rc) * 32 + j;
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int row = mrow0 + (e & 3) + 8 * (e >> 2) + 4 * hl;
                fold[rc][e] = (row < a.M && r < a.rank) ? a.f[(size_t)row * a.rank + r] : 0.f;
              }
            });
            static_for<RC>([&](auto rcc) {
              constexpr int rc = decltype(rcc)::value, rt = rt0 + rc;
              const int r = rt * 32 + j;
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int row = mrow0 + (e & 3) + 8 * (e >> 2) + 4 * hl;
                float fv = fold[rc][e];
                if (row < a.M && r < a.rank) {
                  const float neg = fmaxf(on[g][rt][e], 0.f) + kEps;
                  float pos = den[rt];
                  if (a.l1 > 0.f) pos += a.l1;
                  if (a.l2 > 0.f) pos += a.l2 * fv;
                  float mult = neg / pos;
                  if (a.gamma != 1.f) mult = powf(mult, a.gamma);
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
                const uint32_t h0 = pack_bf16(v0, v1), h1 = pack_bf16(v2, v3);
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
          __syncthreads();
          // row-major image from the LDS tile: 32 rows x R_PAD/8 sixteen-byte slots per wave
          constexpr int SP = R_PAD / 8;
#pragma unroll
          for (int i = 0; i < (32 * SP) / 64; ++i) {
            const int chunk = i * 64 + lane, rl = chunk / SP, slot = chunk % SP;
            const float* src = tile + rl * LDT + slot * 8;
            u32x4 hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x0 = src[2 * q], x1 = src[2 * q + 1];
              const uint32_t h = pack_bf16(x0, x1);
              hi[q] = h;
              lo[q] = pack_bf16(x0 - bf16_lo(h), x1 - bf16_hi(h));
            }
            const int64_t off = p1_offset(mrow0 + rl, slot * 8, R_PAD);
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o1_hi) + off) = hi;
            if constexpr (X3) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o1_lo) + off) = lo;
          }
          __syncthreads();
        };
        apply_group(std::integral_constant<int, 0>{});
        if constexpr (G == 2) apply_group(std::integral_constant<int, 1>{});
        // partial column sums of this workgroup's rows: lane halves, then the waves (fixed order)
        float* red = reinterpret_cast<float*>(smem);  // [WAVES][R_PAD]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float tot = csum[rt] + __shfl_xor(csum[rt], 32, 64);
          if (hl == 0) red[wave * R_PAD + rt * 32 + j] = tot;
        }
        __syncthreads();
        for (int r = tid; r < R_PAD; r += C::THREADS) {
          float tot = (red[r] + red[R_PAD + r]) + (red[2 * R_PAD + r] + red[3 * R_PAD + r]);
          if constexpr (C::WAVES == 8)
            tot += (red[4 * R_PAD + r] + red[5 * R_PAD + r]) + (red[6 * R_PAD + r] + red[7 * R_PAD + r]);
          a.colsum_part[(size_t)mb * R_PAD + r] = tot;
        }
      }
    }
    if (!fused_done) {
      static_for<G>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        const size_t slab = ((size_t)ks * a.M_pad + (size_t)mb * BM + wave * (32 * G) + 32 * g) * R_PAD;
        static_for<RT>([&](auto rtc) {
          constexpr int rt = decltype(rtc)::value;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * hl;
            const size_t idx = slab + (size_t)row * R_PAD + rt * 32 + j;
            a.slab_num[idx] = on[g][rt][e];
            if constexpr (C::TWO_ACC) a.slab_den[idx] = op[g][rt][e];
          }
        });
      });
    }
  }
}

// Host-side launcher, one per (R_PAD) translation unit.
int launch_fused_r32(int beta_kind, int x3, int mode, int stage, int g, const FusedArgs& a, int grid, hipStream_t s);
int launch_fused_r64(int beta_kind, int x3, int mode, int stage, int g, const FusedArgs& a, int grid, hipStream_t s);
int launch_fused_r128(int beta_kind, int x3, int mode, int stage, int g, const FusedArgs& a, int grid, hipStream_t s);
int launch_fused_r256(int beta_kind, int x3, int mode, int stage, int g, const FusedArgs& a, int grid, hipStream_t s);

// Per-device "attribute set" memo (one host thread may drive several devices)
inline bool* attr_flag(bool (&flags)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  return &flags[dev];
}

template <int R_PAD, int BETA, bool X3, int MODE, int STAGE, int G>
int launch_one(const FusedArgs& a, int grid, hipStream_t s) {
  using C = FusedCfg<R_PAD, BETA, X3, MODE, G>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
  auto kern = fused_kernel<R_PAD, BETA, X3, MODE, STAGE, G>;
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

// Which (beta, precision, mode) combinations get the 256-row (G = 2) tile: only those whose accumulators fit the
// 512-register file without spilling -- beta == 1 numerators and the loss, bf16 operands, padded rank <= 128.
constexpr bool has_g2(int r_pad, int beta, bool x3, int mode) {
  return !x3 && r_pad <= 128 && mode != kModeDen && (beta == kKL || mode == kModeLoss);
}

template <int R_PAD, bool ALLOW_X3>
int launch_fused_dispatch(int beta_kind, int x3, int mode, int stage, int g, const FusedArgs& a, int grid,
                          hipStream_t s) {
#define NMFMU_CASE(B, X, M, S)                                                                       \
  if (beta_kind == B && x3 == (X ? 1 : 0) && mode == M && stage == S) {                              \
    if (g == 1) return launch_one<R_PAD, B, X, M, S, 1>(a, grid, s);                                 \
    if constexpr (has_g2(R_PAD, B, X, M)) {                                                          \
      if (g == 2) return launch_one<R_PAD, B, X, M, S, 2>(a, grid, s);                               \
    }                                                                                                \
    return -2;                                                                                       \
  }
#define NMFMU_CASE_BETA(X, M, S) NMFMU_CASE(kKL, X, M, S) NMFMU_CASE(kEuc, X, M, S) NMFMU_CASE(kIS, X, M, S) NMFMU_CASE(kGen, X, M, S)
  NMFMU_CASE_BETA(false, kModeMU, 0)
  NMFMU_CASE_BETA(false, kModeMU, 1)
  NMFMU_CASE_BETA(false, kModeLoss, 0)
  NMFMU_CASE_BETA(false, kModeLoss, 1)
  NMFMU_CASE(kGen, false, kModeDen, 1)
  if constexpr (ALLOW_X3) {
    NMFMU_CASE(kGen, true, kModeDen, 1)
    NMFMU_CASE_BETA(true, kModeMU, 0)
    NMFMU_CASE_BETA(true, kModeMU, 1)
    NMFMU_CASE_BETA(true, kModeLoss, 0)
    NMFMU_CASE_BETA(true, kModeLoss, 1)
  }
#undef NMFMU_CASE_BETA
#undef NMFMU_CASE
  return -2;  // unsupported combination
}

}  // namespace nmfmu
