// Convolutive NMF (NMFD, nmf.py:700-779 of the reference) on MI355X: GEMM instantiations + the small
// memory-bound kernels around them (strided packing, Toeplitz unfold of H, fold + MU apply).
//
// With W (C, R, T) viewed as Wm (C x R*T) and Hu[(b,l)][(r,t)] = H[b][r][l-t] (zero outside 0 <= l-t < Lh):
//   reconstruction   S[c][(b,l)]  = sum_{r'} Wm[c][r'] Hu[(b,l)][r']                    (nmf.py:776-779)
//   W numerator      num[c][r']   = sum_{(b,l)} Gn[c][(b,l)] Hu[(b,l)][r']              (conv backward wrt W)
//   H numerator      Y[r'][(b,l)] = sum_c Wm[c][r'] Gn[c][(b,l)] ;  neg[b][r][j] = sum_t Y[(r,t)][(b,j+t)]
//   beta == 1 denominators: sum_{b,j} H[b][r][j]  and  sum_{c,t} W[c][r][t]              (nmf.py:122-131)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/nmfmu.h"
#include "nmfmu_gemm.h"

namespace nmfmu {

int launch_gemm(int x3, int epi, int beta_kind, int ops, int f16, const GemmArgs& a, hipStream_t s) {
  if (f16) {
    // fp16 operands: the combinations of the beta == 1 NMFD iteration on implicit operands (nmfd_engine.py, precision 'f16')
    if (x3) return -2;
    if (ops == kOpsAWin) {   // H numerator on shifted ratio rows (fp16 ratio planes and Wk planes)
      if (epi != kEpiF32) return -2;
      if (a.n_pad % 128 == 0) return launch_gemm_one<false, kEpiF32, kEuc, kOpsAWin, GemmSmall, kOpF16>(a, s);
      if (a.n_pad % 64 == 0) return launch_gemm_one<false, kEpiF32, kEuc, kOpsAWin, GemmN64, kOpF16>(a, s);
      return launch_gemm_one<false, kEpiF32, kEuc, kOpsAWin, GemmN32, kOpF16>(a, s);
    }
    if (a.koff) {            // several shift axes; at most 64 channels: 64-row (resp. 64-column) tiles on the channel side
#define NF16(E, B, O, SH) \
  if (epi == E && (E == kEpiF32 || beta_kind == B) && ops == O) \
    return launch_gemm_one<false, E, B, O, SH, kOpF16, true>(a, s);
      if (ops != kOpsAHu && a.m_pad == 64) {
        NF16(kEpiRatio, kKL, kOpsBHu, GemmM64) NF16(kEpiLoss, kKL, kOpsBHu, GemmM64) NF16(kEpiF32, kEuc, kOpsBHuT, GemmM64)
        return -2;
      }
      if (ops == kOpsAHu && a.n_pad == 64) {
        NF16(kEpiRatio, kKL, kOpsAHu, GemmN64)
        return -2;
      }
      NF16(kEpiRatio, kKL, kOpsBHu, GemmSmall) NF16(kEpiRatio, kKL, kOpsAHu, GemmSmall) NF16(kEpiLoss, kKL, kOpsBHu, GemmSmall)
      NF16(kEpiF32, kEuc, kOpsBHuT, GemmSmall)
#undef NF16
      return -2;
    }
#define GF16(E, B, O) \
  if (epi == E && (E == kEpiF32 || E == kEpiFold || beta_kind == B) && ops == O) \
    return launch_gemm_one<false, E, B, O, GemmSmall, kOpF16>(a, s);
    GF16(kEpiRatio, kKL, kOpsBHu) GF16(kEpiRatio, kKL, kOpsAHu) GF16(kEpiLoss, kKL, kOpsBHu)
    GF16(kEpiF32, kEuc, kOpsBHuT) GF16(kEpiFold, kEuc, kOpsPlanes)
#undef GF16
    return -2;
  }
  if (a.koff) {
    // implicit operands with several shift axes (NMF2D / NMF3D): the same combinations, ND instances
#define N1(X, E, B, O, SH) \
  if (x3 == (X ? 1 : 0) && epi == E && (E == kEpiF32 || beta_kind == B) && ops == O) \
    return launch_gemm_one<X, E, B, O, SH, kOpBf16, true>(a, s);
#define NB(X, E, O, SH) N1(X, E, kKL, O, SH) N1(X, E, kEuc, O, SH) N1(X, E, kIS, O, SH) N1(X, E, kGen, O, SH)
    if (ops != kOpsAHu && a.m_pad == 64) {      // at most 64 channels: 64-row tiles on the channel side
      NB(false, kEpiRatio, kOpsBHu, GemmM64) NB(true, kEpiRatio, kOpsBHu, GemmM64)
      NB(false, kEpiLoss, kOpsBHu, GemmM64) NB(true, kEpiLoss, kOpsBHu, GemmM64)
      N1(false, kEpiF32, kEuc, kOpsBHuT, GemmM64) N1(true, kEpiF32, kEuc, kOpsBHuT, GemmM64)
      return -2;
    }
    if (ops == kOpsAHu && a.n_pad == 64) {      // ... 64-column tiles in the transposed problem
      NB(false, kEpiRatio, kOpsAHu, GemmN64) NB(true, kEpiRatio, kOpsAHu, GemmN64)
      return -2;
    }
    NB(false, kEpiRatio, kOpsBHu, GemmSmall) NB(true, kEpiRatio, kOpsBHu, GemmSmall)
    NB(false, kEpiRatio, kOpsAHu, GemmSmall) NB(true, kEpiRatio, kOpsAHu, GemmSmall)
    NB(false, kEpiLoss, kOpsBHu, GemmSmall) NB(true, kEpiLoss, kOpsBHu, GemmSmall)
    N1(false, kEpiF32, kEuc, kOpsBHuT, GemmSmall) N1(true, kEpiF32, kEuc, kOpsBHuT, GemmSmall)
#undef NB
#undef N1
    return -2;
  }
  // operand combinations that occur (nmfd_engine.py): RATIO with planes | B = Hu | A = Hu; F32 with planes | B = HuT;
  // LOSS with planes | B = Hu
#define G1(X, E, B, O) \
  if (x3 == (X ? 1 : 0) && epi == E && beta_kind == B && ops == O) return launch_gemm_one<X, E, B, O>(a, s);
#define GB(X, E, O) G1(X, E, kKL, O) G1(X, E, kEuc, O) G1(X, E, kIS, O) G1(X, E, kGen, O)
  GB(false, kEpiRatio, kOpsPlanes) GB(true, kEpiRatio, kOpsPlanes) GB(false, kEpiLoss, kOpsPlanes) GB(true, kEpiLoss, kOpsPlanes)
  GB(false, kEpiRatio, kOpsBHu) GB(true, kEpiRatio, kOpsBHu) GB(false, kEpiRatio, kOpsAHu) GB(true, kEpiRatio, kOpsAHu)
  GB(false, kEpiLoss, kOpsBHu) GB(true, kEpiLoss, kOpsBHu)
  if (epi == kEpiF32 && ops == kOpsPlanes)
    return x3 ? launch_gemm_one<true, kEpiF32, kEuc>(a, s) : launch_gemm_one<false, kEpiF32, kEuc>(a, s);
  if (epi == kEpiF32 && ops == kOpsBHuT)
    return x3 ? launch_gemm_one<true, kEpiF32, kEuc, kOpsBHuT>(a, s) : launch_gemm_one<false, kEpiF32, kEuc, kOpsBHuT>(a, s);
  if (epi == kEpiFold && ops == kOpsPlanes)
    return x3 ? launch_gemm_one<true, kEpiFold, kEuc>(a, s) : launch_gemm_one<false, kEpiFold, kEuc>(a, s);
  if (epi == kEpiF32 && ops == kOpsAWin) {   // N = the (padded) rank: 32-, 64- or 128-wide tiles
#define GW(SH) return x3 ? launch_gemm_one<true, kEpiF32, kEuc, kOpsAWin, SH>(a, s) : launch_gemm_one<false, kEpiF32, kEuc, kOpsAWin, SH>(a, s);
    if (a.n_pad % 128 == 0) GW(GemmSmall)
    if (a.n_pad % 64 == 0) GW(GemmN64)
    GW(GemmN32)
#undef GW
  }
#undef GB
#undef G1
  return -2;
}

// dst[row][col] (padded, zero filled) = src[(row / rin) * ros + (row % rin) * ris + (col / cin) * cos + (col % cin) * cis]
struct Pack2D {
  const float* src;
  int rows, cols;
  int rin, cin;
  int64_t ros, ris, cos, cis;
  int rows_pad, cols_pad;
};

__device__ __forceinline__ float pack2d_get(const Pack2D& p, int row, int col) {
  if (row >= p.rows || col >= p.cols) return 0.f;
  return p.src[(int64_t)(row / p.rin) * p.ros + (int64_t)(row % p.rin) * p.ris + (int64_t)(col / p.cin) * p.cos +
               (int64_t)(col % p.cin) * p.cis];
}

// one thread = 8 consecutive columns of one row
template <bool BF16>
__global__ void __launch_bounds__(256) pack2d_kernel(Pack2D p, float* dst_f32, uint16_t* dst_hi, uint16_t* dst_lo,
                                                     uint32_t* flags) {
  const int64_t n8 = (int64_t)p.rows_pad * (p.cols_pad / 8);
  uint32_t bad = 0, mn = 0x7f800000u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / (p.cols_pad / 8)), c0 = (int)(i % (p.cols_pad / 8)) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = pack2d_get(p, row, c0 + e);
      if (flags && row < p.rows && c0 + e < p.cols) {
        bad |= !(v[e] >= 0.f) ? 1u : 0u;
        mn = min(mn, __builtin_bit_cast(uint32_t, v[e]) & 0x7fffffffu);
      }
    }
    const size_t o = (size_t)row * p.cols_pad + c0;
    if constexpr (BF16) {
      u32x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t h = pack_bf16(v[2 * e], v[2 * e + 1]);
        hi[e] = h;
        lo[e] = pack_bf16(v[2 * e] - bf16_lo(h), v[2 * e + 1] - bf16_hi(h));
      }
      *reinterpret_cast<u32x4*>(dst_hi + o) = hi;
      if (dst_lo) *reinterpret_cast<u32x4*>(dst_lo + o) = lo;
    } else {
      *reinterpret_cast<float4*>(dst_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(dst_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
  if (flags) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      bad |= __shfl_xor(bad, o, 64);
      mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      if (bad) atomicOr(&flags[0], 1u);
      atomicMin(&flags[1], mn);
    }
  }
}

// Window tables of H for the implicit (never materialised) Toeplitz operands -- layout in include/nmfmu.h
// (nmfmu_conv_tables).  One thread per chunk; the tables are ~8x H (2 MB at BASELINE configs[3]) where the explicit
// Hu / HuT matrices are T x H each (2 x 52 MB written and 3 x 52 MB read per iteration).
__global__ void __launch_bounds__(256) conv_tables_kernel(const float* __restrict__ H, int B, int R, int Lh, int T,
                                                          u32x4* rev_hi, u32x4* rev_lo, u32x4* fwd_hi, u32x4* fwd_lo,
                                                          int f16) {
  const int JJ = Lh + 2 * T - 2;
  const int64_t n = 1 + (int64_t)B * R * JJ;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float vr[8], vf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) vr[e] = vf[e] = 0.f;
    if (i > 0) {
      const int64_t k = i - 1;
      const int br = (int)(k / JJ), j = (int)(k - (int64_t)br * JJ) - (T - 1);
      const float* row = H + (size_t)br * Lh;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int jr = j - e, jf = j + e;
        if (jr >= 0 && jr < Lh) vr[e] = row[jr];
        if (jf >= 0 && jf < Lh) vf[e] = row[jf];
      }
    }
    u32x4 rh, rl, fh, fl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t h = pack_img(vr[2 * e], vr[2 * e + 1], f16);
      rh[e] = h, rl[e] = pack_bf16(vr[2 * e] - bf16_lo(h), vr[2 * e + 1] - bf16_hi(h));
      h = pack_img(vf[2 * e], vf[2 * e + 1], f16);
      fh[e] = h, fl[e] = pack_bf16(vf[2 * e] - bf16_lo(h), vf[2 * e + 1] - bf16_hi(h));
    }
    rev_hi[i] = rh, fwd_hi[i] = fh;
    if (rev_lo) rev_lo[i] = rl, fwd_lo[i] = fl;
  }
}

// The same tables for H with up to three shift axes (B, R, lh0, lh1, lh2): one 1-D table (along the last axis) per line
// of H zero-padded by t_d - 1 lines on both sides of every outer axis, lines in (b, r, p0, p1) order -- see
// nmfmu_convnd_tables in include/nmfmu.h.  One thread per chunk.
struct ConvGeom {
  int lh[3], t[3], l[3];   // H extent, taps, V extent (= lh + t - 1) per shift axis
  int lh_tot, t_tot, l_tot;
};

__global__ void __launch_bounds__(256) convnd_tables_kernel(const float* __restrict__ H, int B, int R, ConvGeom g,
                                                            u32x4* rev_hi, u32x4* rev_lo, u32x4* fwd_hi, u32x4* fwd_lo,
                                                            int f16) {
  const int jj2 = g.l[2] + g.t[2] - 1, jj1 = g.l[1] + g.t[1] - 1, jj0 = g.l[0] + g.t[0] - 1;
  const int64_t n = 1 + (int64_t)B * R * jj0 * jj1 * jj2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float vr[8], vf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) vr[e] = vf[e] = 0.f;
    if (i > 0) {
      const int64_t k = i - 1;
      const int p2 = (int)(k % jj2);
      const int64_t k1 = k / jj2;
      const int p1 = (int)(k1 % jj1);
      const int64_t k0 = k1 / jj1;
      const int p0 = (int)(k0 % jj0), br = (int)(k0 / jj0);
      const int j0 = p0 - (g.t[0] - 1), j1 = p1 - (g.t[1] - 1), j = p2 - (g.t[2] - 1);
      if (j0 >= 0 && j0 < g.lh[0] && j1 >= 0 && j1 < g.lh[1]) {
        const float* row = H + (((size_t)br * g.lh[0] + j0) * g.lh[1] + j1) * g.lh[2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int jr = j - e, jf = j + e;
          if (jr >= 0 && jr < g.lh[2]) vr[e] = row[jr];
          if (jf >= 0 && jf < g.lh[2]) vf[e] = row[jf];
        }
      }
    }
    u32x4 rh, rl, fh, fl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t h = pack_img(vr[2 * e], vr[2 * e + 1], f16);
      rh[e] = h, rl[e] = pack_bf16(vr[2 * e] - bf16_lo(h), vr[2 * e + 1] - bf16_hi(h));
      h = pack_img(vf[2 * e], vf[2 * e + 1], f16);
      fh[e] = h, fl[e] = pack_bf16(vf[2 * e] - bf16_lo(h), vf[2 * e + 1] - bf16_hi(h));
    }
    rev_hi[i] = rh, fwd_hi[i] = fh;
    if (rev_lo) rev_lo[i] = rl, fwd_lo[i] = fl;
  }
}

// Toeplitz unfold of H (B, R, Lh): Hu [(b,l)][(r,t)] and HuT [(r,t)][(b,l)], both bf16 (hi[, lo]) zero padded.
__global__ void __launch_bounds__(256) conv_unfold_kernel(const float* __restrict__ H, int B, int R, int Lh, int T,
                                                          uint16_t* hu_hi, uint16_t* hu_lo, uint16_t* hut_hi,
                                                          uint16_t* hut_lo, int bl_pad, int rp_pad) {
  const int L = Lh + T - 1;
  const int64_t n_hu = (int64_t)bl_pad * (rp_pad / 8), n_hut = (int64_t)rp_pad * (bl_pad / 8);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_hu + n_hut; i += (int64_t)gridDim.x * 256) {
    const bool tr = i >= n_hu;
    const int64_t k = tr ? i - n_hu : i;
    float v[8];
    size_t o;
    if (!tr) {  // row (b,l), 8 consecutive (r,t): one division, then (r,t) advances incrementally
      const int row = (int)(k / (rp_pad / 8)), c0 = (int)(k % (rp_pad / 8)) * 8;
      const int b = row / L, l = row % L;
      int r = c0 / T, t = c0 - r * T;
      const bool rowok = row < B * L;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int jx = l - t;
        v[e] = (rowok && r < R && jx >= 0 && jx < Lh) ? H[((size_t)b * R + r) * Lh + jx] : 0.f;
        if (++t == T) t = 0, ++r;
      }
      o = (size_t)row * rp_pad + c0;
    } else {    // row (r,t), 8 consecutive (b,l)
      const int row = (int)(k / (bl_pad / 8)), c0 = (int)(k % (bl_pad / 8)) * 8;
      const int r = row / T, t = row % T;
      int b = c0 / L, l = c0 - b * L;
      const bool rowok = row < R * T;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int jx = l - t;
        v[e] = (rowok && b < B && jx >= 0 && jx < Lh) ? H[((size_t)b * R + r) * Lh + jx] : 0.f;
        if (++l == L) l = 0, ++b;
      }
      o = (size_t)row * bl_pad + c0;
    }
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t h = pack_bf16(v[2 * e], v[2 * e + 1]);
      hi[e] = h;
      lo[e] = pack_bf16(v[2 * e] - bf16_lo(h), v[2 * e + 1] - bf16_hi(h));
    }
    uint16_t* dh = tr ? hut_hi : hu_hi;
    uint16_t* dl = tr ? hut_lo : hu_lo;
    *reinterpret_cast<u32x4*>(dh + o) = hi;
    if (dl) *reinterpret_cast<u32x4*>(dl + o) = lo;
  }
}

// slabs[0][i] += slabs[1][i] + ... (fixed order): the split-K partials of a GEMM, before a consumer that takes one slab
__global__ void __launch_bounds__(256) slab_sum_kernel(float* __restrict__ slabs, int64_t n4, int nslab) {
  float4* p = reinterpret_cast<float4*>(slabs);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 acc = p[i];
    int sl = 1;
    for (; sl + 4 <= nslab; sl += 4) {     // four loads in flight, added in slab order
      const float4 v0 = p[sl * n4 + i], v1 = p[(sl + 1) * n4 + i], v2 = p[(sl + 2) * n4 + i], v3 = p[(sl + 3) * n4 + i];
      acc.x = (((acc.x + v0.x) + v1.x) + v2.x) + v3.x, acc.y = (((acc.y + v0.y) + v1.y) + v2.y) + v3.y;
      acc.z = (((acc.z + v0.z) + v1.z) + v2.z) + v3.z, acc.w = (((acc.w + v0.w) + v1.w) + v2.w) + v3.w;
    }
    for (; sl < nslab; ++sl) {
      const float4 v = p[sl * n4 + i];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    p[i] = acc;
  }
}

// out[r] = sum_{o, i} src[(o * R + r) * inner + i].  Two stages, fixed order (deterministic): grid (R, kRankChunks)
// partial sums over contiguous `inner` runs, then one small block per r.
constexpr int kRankChunks = 128;

__global__ void __launch_bounds__(256) rank_sums_partial_kernel(const float* __restrict__ src, int outer, int R, int inner,
                                                                float* __restrict__ part) {
  __shared__ float red[256];
  const int r = blockIdx.x, ch = blockIdx.y;
  // the (o, i) pairs of rank r, flattened, in kRankChunks equal ranges (whatever the shape: many short runs -- W of a
  // small-kernel NMF2D, 64 runs of 128 -- or few long ones -- H); four independent partial sums per thread
  const int64_t total = (int64_t)outer * inner;
  const int64_t per = (total + kRankChunks - 1) / kRankChunks;
  const int64_t lo = ch * per, hi = min(total, lo + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  auto at = [&](int64_t e) {
    const int o = (int)(e / inner), i = (int)(e - (int64_t)o * inner);
    return src[((size_t)o * R + r) * inner + i];
  };
  int64_t e = lo + threadIdx.x;
  for (; e + 768 < hi; e += 1024) s0 += at(e), s1 += at(e + 256), s2 += at(e + 512), s3 += at(e + 768);
  for (; e < hi; e += 256) s0 += at(e);
  const float s = (s0 + s1) + (s2 + s3);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[r * kRankChunks + ch] = red[0];
}

__global__ void __launch_bounds__(128) rank_sums_final_kernel(const float* __restrict__ part, float* __restrict__ out) {
  __shared__ float red[kRankChunks];
  red[threadIdx.x] = part[blockIdx.x * kRankChunks + threadIdx.x];
  __syncthreads();
  for (int o = kRankChunks / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

__device__ __forceinline__ float mu_update(float f, float neg, float pos, bool closed_form, float l1, float l2,
                                           float gamma) {
  neg = fmaxf(neg, 0.f) + kEps;                       // nmf.py:78
  if (!closed_form) pos = fmaxf(pos, 0.f) + kEps;     // nmf.py:83
  if (l1 > 0.f) pos += l1;
  if (l2 > 0.f) pos += l2 * f;
  float mult = neg / pos;
  if (gamma != 1.f) mult = powf(mult, gamma);
  return f * mult;
}

// W (C, R*T) in place.  num/den: fp32 [c_pad][rp_pad].  kl_den[r] = sum_{b,j} H[b][r][j] for beta == 1.
__global__ void __launch_bounds__(256) conv_apply_w_kernel(float* __restrict__ W, int C, int RT, int T,
                                                           const float* __restrict__ num, const float* __restrict__ den,
                                                           const float* __restrict__ kl_den, int rp_pad, float l1,
                                                           float l2, float gamma) {
  const int64_t n = (int64_t)C * RT;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / RT), rp = (int)(i % RT);
    const size_t o = (size_t)c * rp_pad + rp;
    const float pos = kl_den ? kl_den[rp / T] : den[o];
    W[i] = mu_update(W[i], num[o], pos, kl_den != nullptr, l1, l2, gamma);
  }
}

// Third layout of the same tile (round 5, the launch diet of NMF2D / NMF3D): the B operand of the window-operand H-numerator
// GEMM, Wk[r F + d][((to TQ + q) CK + ck) 64 + c'] = W[c = 64 ck + c'][r][t = to T_last + F q + d] (conv_pack_wk_kernel's
// layout).  A 64-channel tile is one ck, so a row of the transposed tile -- 64 channels of one (r, t) -- is one contiguous
// 128-byte run of Wk: the WmT chunks go out a second time to another address.  hi == nullptr: off.
struct WkOut {
  uint16_t* hi = nullptr;
  uint16_t* lo = nullptr;
  int t_last = 1, fold = 1, ck = 1, k_pad = 0;
};

// conv_apply_w + both operand packings in one pass: a 64 (c) x 64 (k = (r,t)) tile of W is updated in place
// (nmf.py:78-92), kept in LDS, and re-emitted as bf16 planes in both layouts the GEMMs read -- Wm [c_pad][rp_pad]
// and WmT [rp_pad][c_pad] (zero in the padding).  Replaces conv_apply_w_kernel + two pack2d_kernel launches
// (11 + 14 + 14 us and two inter-kernel gaps at BASELINE configs[3]).  update == 0: pack only.
template <bool X3>
__global__ void __launch_bounds__(256) conv_apply_pack_w_kernel(float* __restrict__ W, int C, int RT, int T,
                                                                const float* __restrict__ num,
                                                                const float* __restrict__ den,
                                                                const float* __restrict__ kl_den, int c_pad, int rp_pad,
                                                                float l1, float l2, float gamma, int update,
                                                                uint16_t* wm_hi, uint16_t* wm_lo, uint16_t* wmt_hi,
                                                                uint16_t* wmt_lo, const float* __restrict__ scale,
                                                                const float* __restrict__ kl_hpart, int n_hparts,
                                                                float* __restrict__ wcol, int num_slabs, int f16,
                                                                WkOut wk = WkOut{}) {
  constexpr int LDT = 65;
  __shared__ float tile[64 * LDT];
  const int tid = threadIdx.x;
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  // channel tiles that are padding only: their planes and tile sums were zeroed by the pack-only pass and stay zero
  if (update && c0 >= C) return;
  // the tile's own loads first (sixteen elements per thread, consecutive threads along k: coalesced fp32 traffic) ...
  const bool klm = kl_den != nullptr || kl_hpart != nullptr;
  float wv[16], nv[16], dv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = i * 256 + tid, c = c0 + (idx >> 6), kk = k0 + (idx & 63);
    const bool in = c < C && kk < RT;
    wv[i] = in ? W[(size_t)c * RT + kk] : 0.f;
    nv[i] = (in && update) ? num[(size_t)c * rp_pad + kk] : 0.f;
    for (int sl = 1; sl < num_slabs; ++sl)     // split-K partials of the numerator GEMM ([slab][c_pad][rp_pad])
      nv[i] += (in && update) ? num[(size_t)sl * c_pad * rp_pad + (size_t)c * rp_pad + kk] : 0.f;
    dv[i] = (in && update && !klm) ? den[(size_t)c * rp_pad + kk] : 0.f;
    for (int sl = 1; sl < num_slabs; ++sl)
      dv[i] += (in && update && !klm) ? den[(size_t)sl * c_pad * rp_pad + (size_t)c * rp_pad + kk] : 0.f;
  }
  // ... then the beta == 1 denominators sum_{b,j} H[b][r][j], handed over as per-block partials of the kernel that
  // updated H (kl_hpart[r][n_hparts]).  With T >= 64 a 64-wide k range touches at most two ranks: every wave finishes
  // both sums itself (butterfly: every lane ends up with the total, fixed order) -- no LDS, no barrier.
  const int r_lo = k0 / T;
  float den_lo = 0.f, den_hi = 0.f;
  if (update && kl_hpart) {
    const int lane = tid & 63, r_hi = min(r_lo + 1, (RT - 1) / T);
    for (int pp = lane; pp < n_hparts; pp += 64) {
      den_lo += kl_hpart[(size_t)r_lo * n_hparts + pp];
      den_hi += kl_hpart[(size_t)r_hi * n_hparts + pp];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) den_lo += __shfl_xor(den_lo, o, 64), den_hi += __shfl_xor(den_hi, o, 64);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = i * 256 + tid, cl = idx >> 6, kl = idx & 63, c = c0 + cl, kk = k0 + kl;
    float v = wv[i];
    if (c < C && kk < RT) {
      if (update) {
        const float pos = kl_hpart ? (kk < (r_lo + 1) * T ? den_lo : den_hi) : kl_den ? kl_den[kk / T] : dv[i];
        v = mu_update(v, nv[i], pos, klm, l1, l2, gamma);
        W[(size_t)c * RT + kk] = v;
      }
      if (scale) v *= scale[kk / T];   // planes of W * Z (shift-invariant PLCA); the master is not scaled
    }
    tile[cl * LDT + kl] = v;
  }
  __syncthreads();
  if (wcol && tid < 64) {
    // sum of this 64 x 64 tile per rank (at most two: T >= 64), [c tile][k tile][2]: the next H half-step finishes
    // sum_{c,t} W[c][r][t] from these (nmf.py:122-131) instead of two reduction launches
    float sacc = 0.f;
#pragma unroll 8
    for (int cl = 0; cl < 64; ++cl) sacc += tile[cl * LDT + tid];
    const bool second = (k0 + tid) / T != r_lo;
    float a0 = second ? 0.f : sacc, a1 = second ? sacc : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a0 += __shfl_xor(a0, o, 64), a1 += __shfl_xor(a1, o, 64);
    if (tid == 0) {
      float* po = wcol + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
      po[0] = a0, po[1] = a1;
    }
  }
  for (int idx = tid; idx < 2 * 64 * 8; idx += 256) {  // 16-byte chunks: 512 of Wm (8 consecutive k), 512 of WmT (8 c)
    const bool tr = idx >= 512;
    const int q = idx & 511, row = q >> 3, ch = (q & 7) * 8;
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x0 = tr ? tile[(ch + 2 * e) * LDT + row] : tile[row * LDT + ch + 2 * e];
      const float x1 = tr ? tile[(ch + 2 * e + 1) * LDT + row] : tile[row * LDT + ch + 2 * e + 1];
      const uint32_t h = pack_img(x0, x1, f16);
      hi[e] = h;
      lo[e] = pack_bf16(x0 - bf16_lo(h), x1 - bf16_hi(h));
    }
    const size_t o = tr ? (size_t)(k0 + row) * c_pad + c0 + ch : (size_t)(c0 + row) * rp_pad + k0 + ch;
    *reinterpret_cast<u32x4*>((tr ? wmt_hi : wm_hi) + o) = hi;
    if constexpr (X3) *reinterpret_cast<u32x4*>((tr ? wmt_lo : wm_lo) + o) = lo;
    if (tr && wk.hi && k0 + row < RT && c0 < wk.ck * 64) {
      const int kk = k0 + row, r = kk / T, t = kk - r * T;
      const int to = t / wk.t_last, tl = t - to * wk.t_last, q = tl / wk.fold, d = tl - q * wk.fold;
      const size_t ow = (size_t)(r * wk.fold + d) * wk.k_pad + ((size_t)(to * (wk.t_last / wk.fold) + q) * wk.ck + c0 / 64) * 64 + ch;
      *reinterpret_cast<u32x4*>(wk.hi + ow) = hi;
      if constexpr (X3) *reinterpret_cast<u32x4*>(wk.lo + ow) = lo;
    }
  }
}

// The same update from the diagonal sums an EPI_FOLD GEMM left behind (nmfmu_gemm.h): part[(tm, tn)][seg][dd], 128 x 128
// tiles of Y, seg = 2 * (second r of the tile row) + (second b of the tile column), dd = (n - m) - 128 (tn - tm) + 127.
// A (b, r, j) collects one value per tile its diagonal crosses: <= ceil(T/128)+1 tile rows x <= 2 tile columns, always
// in the same order.  One thread per j: consecutive threads read consecutive dd.
//
// TAB (round 3): the same launch also rewrites the window tables of the new H (nmfmu_conv_tables: a launch of its own
// before, 5 us of a 250 us iteration).  A table entry spans eight consecutive j, so a block owns kTabOwn = 242 positions
// and recomputes a halo of 7 on either side (the update is cheap; 6 % more blocks).  Recomputing a neighbour's element
// needs its OLD value while the neighbour overwrites it: the old values are therefore read from a shadow copy of H that
// nobody writes during this launch, and the new ones go to the master and to a second shadow, the two shadows swapping
// roles every iteration (tb.h_old / tb.h_next).  Positions run over [-7, Lh + 6]: the entries whose windows reach into
// [0, Lh) -- all others are zero for ever and are written once by nmfmu_conv_tables.
struct FoldTables {
  const float* h_old;
  float* h_next;
  u32x4 *rev_hi, *rev_lo, *fwd_hi, *fwd_lo;
  int f16;
};
constexpr int kTabOwn = 256 - 14;

template <bool TAB>
__global__ void __launch_bounds__(256) conv_fold_parts_apply_h_kernel(float* __restrict__ H, int B, int R, int Lh, int T,
                                                                      const float* __restrict__ pnum,
                                                                      const float* __restrict__ pden,
                                                                      const float* __restrict__ kl_den,
                                                                      const float* __restrict__ kl_wcol, int c_tiles,
                                                                      int rp_pad, float* __restrict__ hsum_part,
                                                                      int tiles_n, float l1, float l2, float gamma,
                                                                      int tail_tm0, int tail_split, size_t slab,
                                                                      FoldTables tb) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int L = Lh + T - 1;
  const int jblocks = TAB ? (Lh + 14 + kTabOwn - 1) / kTabOwn : (Lh + 255) / 256;
  const int jb = blockIdx.x % jblocks, r = (blockIdx.x / jblocks) % R, b = blockIdx.x / (jblocks * R);
  const int jx = TAB ? jb * kTabOwn + tid - 14 : jb * 256 + tid;
  const bool valid = jx >= 0 && jx < Lh;
  const bool owned = !TAB || (tid >= 7 && tid < 7 + kTabOwn);
  auto block_sum = [&](float v) {   // fixed-order tree: identical bits in every block that sums the same values
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const float tot = red[0];
    __syncthreads();
    return tot;
  };
  // beta == 1 denominator sum_{c,t} W[c][r][t] (nmf.py:122-131): either finished (kl_den) or as the per-tile sums
  // conv_apply_pack_w left behind (kl_wcol: no separate reduction launches)
  const bool kl = kl_den != nullptr || kl_wcol != nullptr;
  float den = 0.f;
  if (kl_wcol) {
    // [c tile][k tile][2] tile sums of W (conv_apply_pack_w): the k tiles that hold taps of rank r, slot = second rank
    const int kt_lo = (r * T) / 64, kt_n = (r * T + T - 1) / 64 - kt_lo + 1, k_tiles = rp_pad / 64;
    float sacc = 0.f;
    for (int i = tid; i < c_tiles * kt_n; i += 256) {
      const int ct = i / kt_n, kt = kt_lo + (i - ct * kt_n);
      sacc += kl_wcol[((size_t)ct * k_tiles + kt) * 2 + (r > (kt * 64) / T ? 1 : 0)];
    }
    den = block_sum(sacc);
  } else if (kl_den) {
    den = kl_den[r];
  }
  const int m_lo = r * T, m_hi = m_lo + T;
  const int diag = jx + b * L - m_lo;        // n - m of every element of this sum
  float neg = 0.f, pos = 0.f, hv = 0.f;
  const int tm_lo = m_lo / 128, tm_hi = (m_hi - 1) / 128;
  if (valid && tm_hi - tm_lo < 6 && tm_hi < tail_tm0) {
    // the usual case (<= 640 taps, no tail-split rows): <= 6 tile rows x 2 tile columns, every load issued before the
    // first add (the general loop below serialises the L2 latencies: its trip counts are run-time values)
    float vn[12], vd[12];
#pragma unroll
    for (int sr = 0; sr < 6; ++sr) {
      const int tm = tm_lo + sr;
      const int tap_a = max(m_lo, tm * 128) - m_lo, tap_b = min(m_hi, tm * 128 + 128) - m_lo;   // (not `tb`: the tables)
      const int rbit = r > (tm * 128) / T ? 2 : 0;
      const int tn_a = (b * L + jx + tap_a) / 128, tn_z = (b * L + jx + tap_b - 1) / 128;
#pragma unroll
      for (int sc = 0; sc < 2; ++sc) {
        const int tn = sc ? tn_z : tn_a;
        const bool ok = tm <= tm_hi && (sc == 0 || tn_z > tn_a);
        const int seg = rbit + (b > (tn * 128) / L ? 1 : 0);
        const int dd = diag - 128 * (tn - tm) + 127;
        const size_t i = ((size_t)(tm * tiles_n + tn) * 4 + seg) * 256 + dd;
        vn[2 * sr + sc] = ok ? pnum[i] : 0.f;
        vd[2 * sr + sc] = (ok && !kl) ? pden[i] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) neg += vn[q], pos += vd[q];
    if (kl) pos = den;
    const size_t i = ((size_t)b * R + r) * Lh + jx;
    hv = mu_update(TAB ? tb.h_old[i] : H[i], neg, pos, kl, l1, l2, gamma);
    if (owned) {
      H[i] = hv;
      if constexpr (TAB) tb.h_next[i] = hv;
    }
  } else if (valid) {
    for (int tm = m_lo / 128; tm <= (m_hi - 1) / 128; ++tm) {
      const int tap_a = max(m_lo, tm * 128) - m_lo, tap_b = min(m_hi, tm * 128 + 128) - m_lo;   // taps inside this tile row
      const int rbit = r > (tm * 128) / T ? 2 : 0;
      const int na = b * L + jx + tap_a, nz = b * L + jx + tap_b - 1;
      for (int tn = na / 128; tn <= nz / 128; ++tn) {
        const int seg = rbit + (b > (tn * 128) / L ? 1 : 0);
        const int dd = diag - 128 * (tn - tm) + 127;
        const size_t i = ((size_t)(tm * tiles_n + tn) * 4 + seg) * 256 + dd;
        const int nz = tm >= tail_tm0 ? tail_split : 1;   // tile rows of the GEMM's tail-round split: one slab per part
        for (int z = 0; z < nz; ++z) {
          neg += pnum[i + z * slab];
          if (!kl) pos += pden[i + z * slab];
        }
      }
    }
    if (kl) pos = den;
    const size_t i = ((size_t)b * R + r) * Lh + jx;
    hv = mu_update(TAB ? tb.h_old[i] : H[i], neg, pos, kl, l1, l2, gamma);
    if (owned) {
      H[i] = hv;
      if constexpr (TAB) tb.h_next[i] = hv;
    }
  }
  if (hsum_part) {   // partial sum_{b,j} H[b][r][j] of the NEW H for the next W half-step: [r][b * jblocks + jb]
    const float tot = block_sum(owned ? hv : 0.f);
    if (tid == 0) hsum_part[(size_t)r * (B * jblocks) + b * jblocks + jb] = tot;
  }
  if constexpr (TAB) {
    // window tables (layout: nmfmu_conv_tables): entry 1 + (b R + r) JJ + j + (T - 1) holds {H[j], H[j-1], .., H[j-7]}
    // (reversed) resp. {H[j], .., H[j+7]} (forward), zeros outside [0, Lh)
    red[tid] = hv;                     // (0 where not valid; every block_sum ended with a barrier)
    __syncthreads();
    if (owned) {
      const int JJ = Lh + 2 * T - 2;
      const size_t e = 1 + ((size_t)b * R + r) * JJ + (size_t)(jx + T - 1);
      if (jx <= Lh + 6) {              // jx >= -7 always; reversed windows with j in [0, Lh + 6]
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = red[tid - q];
        u32x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t h = pack_img(v[2 * q], v[2 * q + 1], tb.f16);
          hi[q] = h, lo[q] = pack_bf16(v[2 * q] - bf16_lo(h), v[2 * q + 1] - bf16_hi(h));
        }
        if (jx >= 0) {
          tb.rev_hi[e] = hi;
          if (tb.rev_lo) tb.rev_lo[e] = lo;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = red[tid + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t h = pack_img(v[2 * q], v[2 * q + 1], tb.f16);
          hi[q] = h, lo[q] = pack_bf16(v[2 * q] - bf16_lo(h), v[2 * q + 1] - bf16_hi(h));
        }
        if (jx <= Lh - 1) {            // forward windows with j in [-7, Lh - 1]
          tb.fwd_hi[e] = hi;
          if (tb.fwd_lo) tb.fwd_lo[e] = lo;
        }
      }
    }
  }
}

// H (B, R, Lh) in place; neg[b][r][j] = sum_t Y[(r,t)][(b, j+t)], Y fp32 [rp_pad][bl_pad].
// Block = 64 consecutive j x 4 tap groups (coalesced 256-byte reads along j); the four partial sums are combined
// through LDS in a fixed order.
__global__ void __launch_bounds__(256) conv_fold_apply_h_kernel(float* __restrict__ H, int B, int R, int Lh, int T,
                                                                const float* __restrict__ ynum,
                                                                const float* __restrict__ yden,
                                                                const float* __restrict__ kl_den, int bl_pad, float l1,
                                                                float l2, float gamma) {
  __shared__ float red[2][4][64];
  const int L = Lh + T - 1;
  const int jl = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int jblocks = (Lh + 63) / 64;
  const int jb = blockIdx.x % jblocks, r = (blockIdx.x / jblocks) % R, b = blockIdx.x / (jblocks * R);
  const int jx = jb * 64 + jl;
  float neg = 0.f, pos = 0.f;
  if (jx < Lh) {
    const size_t base = (size_t)r * T * bl_pad + (size_t)b * L + jx;
    for (int t = tg; t < T; t += 4) neg += ynum[base + (size_t)t * bl_pad + t];
    if (!kl_den)
      for (int t = tg; t < T; t += 4) pos += yden[base + (size_t)t * bl_pad + t];
  }
  red[0][tg][jl] = neg;
  red[1][tg][jl] = pos;
  __syncthreads();
  if (tg == 0 && jx < Lh) {
    neg = (red[0][0][jl] + red[0][1][jl]) + (red[0][2][jl] + red[0][3][jl]);
    pos = kl_den ? kl_den[r] : (red[1][0][jl] + red[1][1][jl]) + (red[1][2][jl] + red[1][3][jl]);
    const size_t i = ((size_t)b * R + r) * Lh + jx;
    H[i] = mu_update(H[i], neg, pos, kl_den != nullptr, l1, l2, gamma);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Ragged channels of the NMFD reconstruction.  The GEMMs work on 128-row tiles; a spectrogram with 2^k + 1 bins (1025 at
// BASELINE configs[3]) pays a whole tile row -- 64 of 576 workgroups, a second scheduling round on 512 slots, 30 us per
// reconstruction -- for ONE channel.  The host instead runs the GEMM over the first floor(C/128)*128 channels and this
// kernel over the rest: S[c][(b,l)] = sum_{r,t} W[c][r][t] H[b][r][l-t] (nmf.py:776-779) by direct summation from the
// fp32 masters (rounded to bf16 first in the single-plane mode, like the GEMM's operands), then the same elementwise
// epilogue.  mode 0: ratio planes [c][(b,l)] (W half-step), 1: [(b,l)][c] (H half-step), 2: loss partials.
// Block = 64 frames x 8 rank groups (one wave each, private W row + H window in LDS), combined in a fixed order.
// ------------------------------------------------------------------------------------------------------------
struct RaggedArgs {
  const float* w;   // (C, R, T)
  const float* h;   // (B, R, Lh)
  int C, R, T, B, Lh, c0, x3, mode, f16;
  float beta;
  const float* x;   // mode 0 / 2: [c][ld], mode 1: [(b,l)][ld]
  int64_t ld;
  uint16_t *gn_hi, *gn_lo, *gp_hi, *gp_lo;
  float* loss_part; // mode 2: gridDim.x * gridDim.y partials
};

template <int BETA>
__global__ void __launch_bounds__(512) conv_ragged_rows_kernel(RaggedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rsm[];
  const int tid = threadIdx.x, ll = tid & 63, rg = tid >> 6;
  const int L = a.Lh + a.T - 1, lblocks = (L + 63) / 64;
  const int b = blockIdx.x / lblocks, l0 = (blockIdx.x - b * lblocks) * 64;
  const int c = a.c0 + blockIdx.y;
  const int HS = a.T + 63;                       // H window of a 64-frame block: j = l0 - (T-1) .. l0 + 63
  const int TP = (a.T + 3) & ~3, WS = (TP + HS + 3) & ~3;   // 16-byte aligned pieces: W is read four taps at a time
  float* wl = rsm + rg * WS;                     // this wave's W[c][r][:] ...
  float* hs = wl + TP;                           // ... and H[b][r][window]
  // the target element of this lane's frame, fetched now so that its latency hides behind the summation
  const int lx = l0 + ll;
  const size_t xidx = a.mode == 1 ? ((size_t)b * L + lx) * a.ld + c : (size_t)c * a.ld + (size_t)b * L + lx;
  const float xval = (rg == 0 && lx < L) ? a.x[xidx] : 0.f;
  auto rnd = [&](float v) {     // what the GEMM's operand planes hold
    return a.x3 ? v : a.f16 ? unpack_lo<kOpF16>(pack_img(v, 0.f, 1)) : bf16_lo(pack_bf16(v, 0.f));
  };
  float s = 0.f;
  for (int r0 = 0; r0 < a.R; r0 += 8) {
    const int r = r0 + rg;
    if (r < a.R) {
      // eight loads in flight per lane, then the LDS writes (a load -> write loop serialises the miss latencies)
      const float* wp = a.w + ((size_t)c * a.R + r) * a.T;
      const float* hp = a.h + ((size_t)b * a.R + r) * a.Lh;
      for (int t0 = 0; t0 < a.T; t0 += 512) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (t0 + u * 64 + ll < a.T) ? wp[t0 + u * 64 + ll] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (t0 + u * 64 + ll < a.T) wl[t0 + u * 64 + ll] = rnd(v[u]);
      }
      for (int k0 = 0; k0 < HS; k0 += 512) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int jx = l0 - (a.T - 1) + k0 + u * 64 + ll;
          v[u] = (k0 + u * 64 + ll < HS && jx >= 0 && jx < a.Lh) ? hp[jx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u * 64 + ll < HS) hs[k0 + u * 64 + ll] = rnd(v[u]);
      }
    }
    __syncthreads();
    if (r < a.R) {
      const float* hq = hs + ll + a.T - 1;       // H[b][r][l - t] = hq[-t]
      float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
      int t = 0;
      for (; t + 16 <= a.T; t += 16) {           // LDS latency, not bandwidth, bounds this loop: 32 reads in flight
        float wv[16], hv[16];
#pragma unroll
        for (int u = 0; u < 16; u += 4) *reinterpret_cast<float4*>(wv + u) = *reinterpret_cast<const float4*>(wl + t + u);
#pragma unroll
        for (int u = 0; u < 16; ++u) hv[u] = hq[-t - u];
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
          p0 = fmaf(wv[u], hv[u], p0), p1 = fmaf(wv[u + 1], hv[u + 1], p1);
          p2 = fmaf(wv[u + 2], hv[u + 2], p2), p3 = fmaf(wv[u + 3], hv[u + 3], p3);
        }
      }
      for (; t < a.T; ++t) p0 = fmaf(wl[t], hq[-t], p0);
      s += (p0 + p1) + (p2 + p3);
    }
    __syncthreads();
  }
  float* red = rsm;                              // [8][64]
  red[rg * 64 + ll] = s;
  __syncthreads();
  float lv = 0.f;
  if (rg == 0) {
    float S = (BETA != kEuc) ? kEps : 0.f;       // the GEMM seeds its accumulators the same way
#pragma unroll
    for (int g = 0; g < 8; ++g) S += red[g * 64 + ll];
    const int l = l0 + ll;
    if (l < L) {
      const size_t idx = xidx;
      const float x = xval;
      if (a.mode == 2) {
        lv = loss_elem<BETA>(S, x, a.beta);
      } else {
        float gn, gp;
        mu_elem<BETA>(S, x, a.beta, gn, gp);
        const uint32_t nh = pack_img(gn, 0.f, a.f16);
        a.gn_hi[idx] = (uint16_t)nh;
        if (a.x3) a.gn_lo[idx] = (uint16_t)pack_bf16(gn - bf16_lo(nh), 0.f);
        if constexpr (BETA != kKL) {
          const uint32_t ph = pack_bf16(gp, 0.f);
          a.gp_hi[idx] = (uint16_t)ph;
          if (a.x3) a.gp_lo[idx] = (uint16_t)pack_bf16(gp - bf16_lo(ph), 0.f);
        }
      }
    }
  }
  if (a.mode == 2) {
    __syncthreads();
    if (rg == 0) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) lv += __shfl_xor(lv, o, 64);
      if (ll == 0) a.loss_part[blockIdx.y * gridDim.x + blockIdx.x] = lv;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// NMF2D / NMF3D (nmf.py:782-942): the same unfold / fold with up to three shift axes.  Axes are stored outermost
// first and missing leading axes have extent 1, so 1-D .. 3-D share this code.  Used by the explicit-operand path
// only (these layers are small: the reference's own examples are (33,50) x 3x3 and (64,64,100) x 5x5x20).
// ------------------------------------------------------------------------------------------------------------
// Hu [(b,l)][(r,t)] and HuT [(r,t)][(b,l)], bf16 (hi[, lo]) zero padded: one thread per 8 consecutive columns.
// The row index is decomposed once per thread and the column index once, then advanced with carries (the naive
// per-element div/mod chain made this kernel ALU-bound at ~2.5 ms for a 256 x 512 frame).
__global__ void __launch_bounds__(256) convnd_unfold_kernel(const float* __restrict__ H, int B, int R, ConvGeom g,
                                                            uint16_t* hu_hi, uint16_t* hu_lo, uint16_t* hut_hi,
                                                            uint16_t* hut_lo, int bl_pad, int rp_pad) {
  const int64_t n_hu = (int64_t)bl_pad * (rp_pad / 8), n_hut = (int64_t)rp_pad * (bl_pad / 8);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_hu + n_hut; i += (int64_t)gridDim.x * 256) {
    const bool tr = i >= n_hu;
    const int64_t k = tr ? i - n_hu : i;
    const int cols8 = (tr ? bl_pad : rp_pad) / 8;
    const int row = (int)(k / cols8), c0 = (int)(k % cols8) * 8;
    // (b, l0, l1, l2) of the (b,l) index and (r, t0, t1, t2) of the (r,t) index: one is the row, the other starts at c0
    const int bl0 = tr ? c0 : row, rt0 = tr ? row : c0;
    int b = bl0 / g.l_tot, lf = bl0 - b * g.l_tot;
    int l2 = lf % g.l[2], l01 = lf / g.l[2], l1 = l01 % g.l[1], l0 = l01 / g.l[1];
    int r = rt0 / g.t_tot, tf = rt0 - r * g.t_tot;
    int t2 = tf % g.t[2], t01 = tf / g.t[2], t1 = t01 % g.t[1], t0 = t01 / g.t[1];
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = 0.f;
      const int j0 = l0 - t0, j1 = l1 - t1, j2 = l2 - t2;
      if (b < B && r < R && j0 >= 0 && j0 < g.lh[0] && j1 >= 0 && j1 < g.lh[1] && j2 >= 0 && j2 < g.lh[2])
        x = H[((size_t)b * R + r) * g.lh_tot + (j0 * g.lh[1] + j1) * g.lh[2] + j2];
      v[e] = x;
      if (tr) {   // next (b,l)
        if (++l2 == g.l[2]) { l2 = 0; if (++l1 == g.l[1]) { l1 = 0; if (++l0 == g.l[0]) { l0 = 0; ++b; } } }
      } else {    // next (r,t)
        if (++t2 == g.t[2]) { t2 = 0; if (++t1 == g.t[1]) { t1 = 0; if (++t0 == g.t[0]) { t0 = 0; ++r; } } }
      }
    }
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t h = pack_bf16(v[2 * e], v[2 * e + 1]);
      hi[e] = h;
      lo[e] = pack_bf16(v[2 * e] - bf16_lo(h), v[2 * e + 1] - bf16_hi(h));
    }
    const size_t o = (size_t)row * (tr ? bl_pad : rp_pad) + c0;
    *reinterpret_cast<u32x4*>((tr ? hut_hi : hu_hi) + o) = hi;
    if (hu_lo) *reinterpret_cast<u32x4*>((tr ? hut_lo : hu_lo) + o) = lo;
  }
}

// H (B, R, *lh) in place; neg[b][r][j] = sum_t Y[(r,t)][(b, j + t)].  One thread per element, taps in a fixed order.
__global__ void __launch_bounds__(256) convnd_fold_apply_h_kernel(float* __restrict__ H, int B, int R, ConvGeom g,
                                                                  const float* __restrict__ ynum,
                                                                  const float* __restrict__ yden,
                                                                  const float* __restrict__ kl_den, int bl_pad,
                                                                  float l1, float l2, float gamma,
                                                                  float* __restrict__ fold_out) {
  const int64_t n = (int64_t)B * R * g.lh_tot;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int jf = (int)(i % g.lh_tot), br = (int)(i / g.lh_tot), r = br % R, b = br / R;
    const int j2 = jf % g.lh[2], j01 = jf / g.lh[2], j1 = j01 % g.lh[1], j0 = j01 / g.lh[1];
    float neg = 0.f, pos = 0.f;
    int tf = 0;
    for (int t0 = 0; t0 < g.t[0]; ++t0)
      for (int t1 = 0; t1 < g.t[1]; ++t1)
        for (int t2 = 0; t2 < g.t[2]; ++t2, ++tf) {
          const int lf = ((j0 + t0) * g.l[1] + (j1 + t1)) * g.l[2] + (j2 + t2);
          const size_t o = ((size_t)r * g.t_tot + tf) * bl_pad + (size_t)b * g.l_tot + lf;
          neg += ynum[o];
          if (!kl_den) pos += yden[o];
        }
    if (fold_out) fold_out[i] = neg;   // fold only (shift-invariant PLCA applies its own update)
    else H[i] = mu_update(H[i], neg, kl_den ? kl_den[r] : pos, kl_den != nullptr, l1, l2, gamma);
  }
}

// ------------------------------------------------------------------------------------------------------------
// H numerator as a window-operand GEMM (NMFMU_OPS_A_WIN): the two small kernels around it.
// ------------------------------------------------------------------------------------------------------------
// Wk[r * F + d][((to * TQ + q) * CK + ck) * 64 + c'] = W[c = ck * 64 + c'][r][to * T_last + F q + d]  (TQ = T_last / F; F = 1:
// Wk[r][(t * CK + ck) * 64 + c']), zero padded: the B operand.  One thread per 16-byte chunk.
__global__ void __launch_bounds__(256) conv_pack_wk_kernel(const float* __restrict__ W, int C, int R, int T, int T_last, int F,
                                                           int CK, int rows_pad, int k_pad, uint16_t* hi, uint16_t* lo, int f16) {
  const int64_t n8 = (int64_t)rows_pad * (k_pad / 8);
  const int ct = CK * 64, TQ = T_last / F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / (k_pad / 8)), k0 = (int)(i % (k_pad / 8)) * 8;
    const int r = n / F, dlt = n - r * F;
    const int tq = k0 / ct, c0 = k0 - tq * ct;            // tq = to * TQ + q
    const int to = tq / TQ, q = tq - to * TQ;
    const int t = to * T_last + F * q + dlt;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r < R && t < T && c0 + e < C) ? W[((size_t)(c0 + e) * R + r) * T + t] : 0.f;
    u32x4 h4, l4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t h = pack_img(v[2 * e], v[2 * e + 1], f16);
      h4[e] = h;
      l4[e] = pack_bf16(v[2 * e] - bf16_lo(h), v[2 * e + 1] - bf16_hi(h));
    }
    const size_t o = (size_t)n * k_pad + k0;
    *reinterpret_cast<u32x4*>(hi + o) = h4;
    if (lo) *reinterpret_cast<u32x4*>(lo + o) = l4;
  }
}

// H (B, R, lh_outer, lh_last) in place from num / den [(b, jo, j')][ld], j' in [0, lh_last + F - 1): with F taps folded
// into the columns, num[b][r][jo][j] = sum_d out[(b, jo, j + d)][r F + d] (fixed order).  One thread per (position, rank),
// rank fastest: the lanes of a position read neighbouring columns of the same F rows.  (A block-per-line version that
// staged the rows in LDS and one thread per position were both measured at 20-30 us for 124 k positions -- too few,
// too serial blocks; this form is bandwidth-shaped.)
__global__ void __launch_bounds__(256) conv_apply_h_rows_kernel(float* __restrict__ H, int B, int R, int lh_outer, int lh_last,
                                                                int F, const float* __restrict__ num,
                                                                const float* __restrict__ den,
                                                                const float* __restrict__ kl_den, int ld, float l1,
                                                                float l2, float gamma, float* __restrict__ fold_out) {
  const int64_t n = (int64_t)B * lh_outer * lh_last * R;
  const int lw = lh_last + F - 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i % R);
    const int64_t pos = i / R;
    const int j = (int)(pos % lh_last);
    const int64_t bo = pos / lh_last;                        // (b, jo)
    const int b = (int)(bo / lh_outer), jo = (int)(bo - (int64_t)b * lh_outer);
    const size_t row0 = ((size_t)bo * lw + j) * ld + (size_t)r * F;
    float neg = 0.f, pos_ = 0.f;
    for (int d = 0; d < F; ++d) {
      neg += num[row0 + (size_t)d * (ld + 1)];
      if (den) pos_ += den[row0 + (size_t)d * (ld + 1)];
    }
    const size_t o = (((size_t)b * R + r) * lh_outer + jo) * lh_last + j;
    if (fold_out) fold_out[o] = neg;    // the sum only (shift-invariant PLCA applies its own update)
    else H[o] = mu_update(H[o], neg, kl_den ? kl_den[r] : pos_, kl_den != nullptr, l1, l2, gamma);
  }
}

// The same update with the beta == 1 rank sums riding in it (round 5, launch diet): sum_{c,t} W[c][r][t] arrives as the tile
// sums conv_apply_pack_w left behind (kl_wcol [c tile][k tile][2], as in the fold-parts kernels above) and is finished per
// block for all ranks; sum_{b,j} H_new[b][r][j] leaves as one partial per block and rank (hsum_part[r][gridDim.x]) for the next
// W half-step.  Thread layout: P = 256 / R positions per block step, r = tid % R is fixed per thread (any R <= 256; the
// last 256 - P R threads idle), so the block's partial per rank is a fixed-order sum of P register sums: deterministic.
__global__ void __launch_bounds__(256) conv_apply_h_rows_sums_kernel(float* __restrict__ H, int B, int R, int lh_outer, int lh_last,
                                                                     int F, const float* __restrict__ num,
                                                                     const float* __restrict__ kl_wcol, int c_tiles, int k_tiles,
                                                                     int T, int ld, float l1, float l2, float gamma,
                                                                     float* __restrict__ hsum_part) {
  __shared__ float den_s[256];
  __shared__ float red[256];
  const int tid = threadIdx.x, P = 256 / R;
  if (tid < R) {
    // the k tiles that hold taps of rank tid, slot = second rank of the tile (fixed order)
    const int kt_lo = (tid * T) / 64, kt_n = (tid * T + T - 1) / 64 - kt_lo + 1;
    float sacc = 0.f;
    for (int ct = 0; ct < c_tiles; ++ct)
      for (int kt = kt_lo; kt < kt_lo + kt_n; ++kt) sacc += kl_wcol[((size_t)ct * k_tiles + kt) * 2 + (tid > (kt * 64) / T ? 1 : 0)];
    den_s[tid] = sacc;
  }
  __syncthreads();
  const int r = tid % R, pl = tid / R;
  const int64_t npos = (int64_t)B * lh_outer * lh_last;
  const int lw = lh_last + F - 1;
  float mine = 0.f;
  if (pl < P) {
    const float pos_ = den_s[r];
    for (int64_t pos = (int64_t)blockIdx.x * P + pl; pos < npos; pos += (int64_t)gridDim.x * P) {
      const int j = (int)(pos % lh_last);
      const int64_t bo = pos / lh_last;                        // (b, jo)
      const int b = (int)(bo / lh_outer), jo = (int)(bo - (int64_t)b * lh_outer);
      const size_t row0 = ((size_t)bo * lw + j) * ld + (size_t)r * F;
      float neg = 0.f;
      for (int d = 0; d < F; ++d) neg += num[row0 + (size_t)d * (ld + 1)];
      const size_t o = (((size_t)b * R + r) * lh_outer + jo) * lh_last + j;
      const float hv = mu_update(H[o], neg, pos_, true, l1, l2, gamma);
      H[o] = hv;
      mine += hv;
    }
  }
  red[tid] = mine;
  __syncthreads();
  if (tid < R) {
    float tot = 0.f;
    for (int p = 0; p < P; ++p) tot += red[p * R + tid];
    hsum_part[(size_t)tid * gridDim.x + blockIdx.x] = tot;
  }
}

// the two rank-sum stages in one launch for small inputs: one block per rank, fixed order
__global__ void __launch_bounds__(1024) rank_sums_one_kernel(const float* __restrict__ src, int outer, int R, int inner,
                                                             float* __restrict__ out) {
  __shared__ float red[1024];
  const int r = blockIdx.x;
  const int64_t total = (int64_t)outer * inner;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  auto at = [&](int64_t e) {
    const int o = (int)(e / inner), i = (int)(e - (int64_t)o * inner);
    return src[((size_t)o * R + r) * inner + i];
  };
  int64_t e = threadIdx.x;
  for (; e + 3072 < total; e += 4096) s0 += at(e), s1 += at(e + 1024), s2 += at(e + 2048), s3 += at(e + 3072);
  for (; e < total; e += 1024) s0 += at(e);
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[r] = red[0];
}

}  // namespace nmfmu

using namespace nmfmu;

namespace {
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16)); }

static int make_geom(int nd, const int32_t* lh, const int32_t* taps, ConvGeom* g) {
  if (nd < 1 || nd > 3 || !lh || !taps) return NMFMU_ERR_ARG;
  for (int d = 0; d < 3; ++d) g->lh[d] = g->t[d] = g->l[d] = 1;
  int64_t lt = 1, tt = 1, ht = 1;
  for (int d = 0; d < nd; ++d) {
    const int s = 3 - nd + d;
    if (lh[d] <= 0 || taps[d] <= 0) return NMFMU_ERR_ARG;
    g->lh[s] = lh[d], g->t[s] = taps[d], g->l[s] = lh[d] + taps[d] - 1;
    lt *= g->l[s], tt *= taps[d], ht *= lh[d];
  }
  if (lt > INT32_MAX / 4 || tt > INT32_MAX / 4) return NMFMU_ERR_ARG;
  g->l_tot = (int)lt, g->t_tot = (int)tt, g->lh_tot = (int)ht;
  return 0;
}

}  // namespace

extern "C" {

int nmfmu_gemm_f16_supported(float beta, int epilogue, int ops) {
  const int kl = nmfmu_beta_kind(beta) == NMFMU_BETA_KL;
  switch (epilogue) {
    case NMFMU_EPI_RATIO: return kl && (ops == NMFMU_OPS_B_HU || ops == NMFMU_OPS_A_HU);
    case NMFMU_EPI_LOSS: return kl && ops == NMFMU_OPS_B_HU;
    case NMFMU_EPI_F32: return ops == NMFMU_OPS_B_HUT;
    case NMFMU_EPI_FOLD: return ops == NMFMU_OPS_PLANES;
    default: return 0;
  }
}

// descriptor -> kernel arguments (everything nmfmu_gemm validates); NMFMU_OK or the error nmfmu_gemm returns
static int gemm_prepare(const nmfmu_gemm_desc* d, int epilogue, GemmArgs& a, int& x3, int& kind, int& f16) {
  if (!d || !d->a_hi || !d->b_hi) return NMFMU_ERR_ARG;
  // tiles are 128 x 128, except: the window-operand GEMM has narrow-N tiles (N = rank), and implicit operands with several
  // shift axes take a 64-row (B_HU / B_HUT) resp. 64-column (A_HU) tile when the channel side is exactly 64
  const bool nd_ops = d->ops != NMFMU_OPS_PLANES && d->ops != NMFMU_OPS_A_WIN && d->win_nd > 1;
  const int n_mult = d->ops == NMFMU_OPS_A_WIN ? 32 : (nd_ops && d->ops == NMFMU_OPS_A_HU && d->n_pad == 64) ? 64 : 128;
  const int m_mult = (nd_ops && d->ops != NMFMU_OPS_A_HU && d->m_pad == 64) ? 64 : 128;
  if (d->m_pad <= 0 || d->n_pad <= 0 || d->k_pad <= 0 || d->m_pad % m_mult || d->n_pad % n_mult || d->k_pad % 128)
    return NMFMU_ERR_ARG;
  x3 = d->precision == NMFMU_PREC_BF16X3, f16 = d->precision == NMFMU_PREC_F16;
  if (x3 && (!d->a_lo || !d->b_lo)) return NMFMU_ERR_ARG;
  kind = nmfmu_beta_kind(d->beta);
  a = GemmArgs{};
  a.a_hi = (const uint16_t*)d->a_hi, a.a_lo = (const uint16_t*)d->a_lo;
  a.b_hi = (const uint16_t*)d->b_hi, a.b_lo = (const uint16_t*)d->b_lo;
  a.m_pad = d->m_pad, a.n_pad = d->n_pad, a.k_pad = d->k_pad;
  a.k_len = d->k_len ? d->k_len : d->k_pad;
  if (a.k_len <= 0 || a.k_len > d->k_pad || a.k_len % 64) return NMFMU_ERR_ARG;
  a.k_split = d->k_split > 1 ? d->k_split : 1;
  a.tail_rows = 0;
  if (epilogue == NMFMU_EPI_FOLD) {
    a.tail_rows = a.k_split > 1 ? d->tail_rows : 0;
    if (a.tail_rows < 0 || a.tail_rows > d->m_pad / 128 || (a.k_split > 1 && a.tail_rows == 0)) return NMFMU_ERR_ARG;
    if (a.tail_rows > 0 && a.k_split > a.k_len / 64) return NMFMU_ERR_ARG;   // more parts than k-tiles
    if (a.tail_rows == 0) a.k_split = 1;
  } else if (a.k_split > 1 && (epilogue != NMFMU_EPI_F32 || (a.k_len / 64) % a.k_split)) {
    return NMFMU_ERR_ARG;
  }
  a.x = d->x;
  a.gn_hi = (uint16_t*)d->gn_hi, a.gn_lo = (uint16_t*)d->gn_lo, a.gp_hi = (uint16_t*)d->gp_hi, a.gp_lo = (uint16_t*)d->gp_lo;
  a.out = d->out;
  a.m_valid = d->m_valid, a.n_valid = d->n_valid;
  a.beta = d->beta;
  if (d->ops < NMFMU_OPS_PLANES || d->ops > NMFMU_OPS_A_WIN) return NMFMU_ERR_ARG;
  ConvGeom geom{};
  const bool nd_geom = d->ops == NMFMU_OPS_A_WIN || (d->ops != NMFMU_OPS_PLANES && d->win_nd > 1);
  if (nd_geom) {
    if (d->t_batch <= 0 || d->t_rank <= 0 || make_geom(d->win_nd, d->win_lh, d->win_taps, &geom)) return NMFMU_ERR_ARG;
    for (int i = 0; i < 3; ++i) a.win_lh[i] = geom.lh[i], a.win_t[i] = geom.t[i], a.win_l[i] = geom.l[i];
  }
  if (d->ops == NMFMU_OPS_A_WIN) {
    // A[(b,j)][(t,c)] = P[(b, j + t)][c]: rows of the plane(s) a_hi / a_lo ([batch * prod(l)][win_pitch]) shifted by the tap
    if (epilogue != NMFMU_EPI_F32) return NMFMU_ERR_ARG;      // (k_split > 1: partial slabs as for any EPI_F32 launch)
    if (d->win_pitch <= 0 || d->win_pitch % 64 || d->win_channels <= 0 || d->win_channels > d->win_pitch) return NMFMU_ERR_ARG;
    a.win_ck = (d->win_channels + 63) / 64;
    if (a.win_ck * 64 > d->win_pitch) return NMFMU_ERR_ARG;
    a.win_pitch = (unsigned)d->win_pitch * 2u;
    // fold F: F consecutive last-axis taps share a k position and go to F columns of N; rows run over lh_last + F - 1
    const int fold = d->win_fold > 1 ? d->win_fold : 1;
    if (geom.t[2] % fold || (int64_t)d->t_rank * fold > d->n_pad) return NMFMU_ERR_ARG;
    a.win_tstep = fold;
    a.win_lh[2] = geom.lh[2] + fold - 1, a.win_t[2] = geom.t[2] / fold;
    const int64_t rows = (int64_t)d->t_batch * geom.lh[0] * geom.lh[1] * a.win_lh[2];
    if (rows > d->m_pad || (int64_t)d->t_batch * geom.l_tot * a.win_pitch >= ((int64_t)1 << 31)) return NMFMU_ERR_ARG;   // 32-bit lane offsets
    a.win_rows = (int)rows;
    if ((int64_t)a.k_len != (int64_t)(geom.t_tot / fold) * a.win_ck * 64) return NMFMU_ERR_ARG;   // k = (t_outer, q, ck, c')
  } else if (nd_geom) {
    // window tables of an H with several shift axes (nmfmu_convnd_tables); t_koff from nmfmu_convnd_koff, on the device
    if (!d->t_koff || epilogue == NMFMU_EPI_FOLD || d->rag_channels > 0) return NMFMU_ERR_ARG;
    if (geom.t[2] % 8 || geom.l[2] % 8) return NMFMU_ERR_ARG;
    a.tB = d->t_batch, a.tR = d->t_rank, a.tT = geom.t_tot, a.tLh = geom.lh_tot;
    a.koff = d->t_koff;
    if (nmfmu_convnd_table_bytes(a.tB, a.tR, d->win_nd, d->win_lh, d->win_taps) >= ((size_t)1 << 31)) return NMFMU_ERR_ARG;   // 32-bit lane offsets
    const int64_t bl = (int64_t)a.tB * geom.l_tot, rt = (int64_t)a.tR * geom.t_tot;
    const int hu_rows = d->ops == NMFMU_OPS_A_HU ? d->m_pad : d->n_pad;
    if (d->ops == NMFMU_OPS_B_HUT ? (hu_rows < rt || d->k_pad < bl) : (hu_rows < bl || d->k_pad < rt)) return NMFMU_ERR_ARG;
  } else if (d->ops != NMFMU_OPS_PLANES) {
    a.tB = d->t_batch, a.tR = d->t_rank, a.tT = d->t_taps, a.tLh = d->t_lh;
    if (a.tB <= 0 || a.tR <= 0 || a.tT <= 0 || a.tLh <= 0 || a.tT % 8 || (a.tLh + a.tT - 1) % 8) return NMFMU_ERR_ARG;
    if (nmfmu_conv_table_bytes(a.tB, a.tR, a.tLh, a.tT) >= ((size_t)1 << 31)) return NMFMU_ERR_ARG;   // 32-bit lane offsets
    const int64_t bl = (int64_t)a.tB * (a.tLh + a.tT - 1), rt = (int64_t)a.tR * a.tT;   // logical extents of Hu
    const int hu_rows = d->ops == NMFMU_OPS_A_HU ? d->m_pad : d->n_pad;
    if (d->ops == NMFMU_OPS_B_HUT ? (hu_rows < rt || d->k_pad < bl) : (hu_rows < bl || d->k_pad < rt)) return NMFMU_ERR_ARG;
  }
  if (epilogue == NMFMU_EPI_RATIO) {
    if (!a.x || !a.gn_hi || (x3 && !a.gn_lo)) return NMFMU_ERR_ARG;
    if (kind != NMFMU_BETA_KL && (!a.gp_hi || (x3 && !a.gp_lo))) return NMFMU_ERR_ARG;
  } else if (epilogue == NMFMU_EPI_F32) {
    if (!a.out) return NMFMU_ERR_ARG;
  } else if (epilogue == NMFMU_EPI_LOSS) {
    if (!a.x || !a.out) return NMFMU_ERR_ARG;
  } else if (epilogue == NMFMU_EPI_FOLD) {
    // rows (r,t), columns (b,l) of the H numerator GEMM; out = nmfmu_fold_part_bytes(m_pad, n_pad)
    if (!a.out || d->ops != NMFMU_OPS_PLANES) return NMFMU_ERR_ARG;
    if (!nmfmu_fold_parts_supported(d->t_batch, d->t_rank, d->t_lh, d->t_taps)) return NMFMU_ERR_UNSUPPORTED;
    a.tB = d->t_batch, a.tR = d->t_rank, a.tT = d->t_taps, a.tLh = d->t_lh;
    if (d->m_pad < a.tR * a.tT || (int64_t)d->n_pad < (int64_t)a.tB * (a.tLh + a.tT - 1)) return NMFMU_ERR_ARG;
  } else {
    return NMFMU_ERR_ARG;
  }
  if (d->rag_channels > 0) {   // ragged channels as an extra MFMA block inside this launch's grid
    if (epilogue != NMFMU_EPI_RATIO || (d->ops != NMFMU_OPS_B_HU && d->ops != NMFMU_OPS_A_HU)) return NMFMU_ERR_ARG;
    const int own = d->ops == NMFMU_OPS_B_HU ? d->m_pad : d->n_pad;    // channels the GEMM's own tiles cover
    if (d->rag_c0 != own || d->rag_channels <= d->rag_c0) return NMFMU_ERR_ARG;
    if (d->ops == NMFMU_OPS_A_HU && d->rag_channels > (d->n_ld ? d->n_ld : d->n_pad)) return NMFMU_ERR_ARG;
    if (!nmfmu_gemm_ragged_supported(d->ops, d->m_pad, d->n_pad, d->rag_channels - d->rag_c0)) return NMFMU_ERR_UNSUPPORTED;
    a.rag_c0 = d->rag_c0, a.rag_C = d->rag_channels;
  }
  if (d->tile_rows != 0 && d->tile_rows != 128) return NMFMU_ERR_UNSUPPORTED;   // (the 256 x 256 tile of ABI 3 is gone)
  a.ldn = d->n_ld ? d->n_ld : d->n_pad;
  if (a.ldn < d->n_pad || (a.ldn != d->n_pad && epilogue == NMFMU_EPI_FOLD)) return NMFMU_ERR_ARG;
  if (d->stage_mode != 0 && d->stage_mode != 1) return NMFMU_ERR_ARG;
  return NMFMU_OK;
}

// window staging of the implicit operand (nmfmu_gemm.h: WS): automatic where the shape allows it
static bool gemm_takes_window(const nmfmu_gemm_desc* d, int epilogue, const GemmArgs& a) {
  return d->stage_mode == 0 && epilogue != NMFMU_EPI_FOLD && d->ops != NMFMU_OPS_PLANES && d->ops != NMFMU_OPS_A_WIN &&
         gemm_window_stageable(d->ops, a);
}

int nmfmu_gemm_window_staged(const nmfmu_gemm_desc* d, int epilogue) {
  GemmArgs a;
  int x3, kind, f16;
  const int rc = gemm_prepare(d, epilogue, a, x3, kind, f16);
  return rc ? rc : (gemm_takes_window(d, epilogue, a) ? 1 : 0);
}

int nmfmu_gemm(const nmfmu_gemm_desc* d, int epilogue, void* stream) {
  GemmArgs a;
  int x3, kind, f16;
  int rc = gemm_prepare(d, epilogue, a, x3, kind, f16);
  if (rc) return rc;
  rc = gemm_takes_window(d, epilogue, a) ? launch_gemm_ws(x3, epilogue, kind, d->ops, f16, a, S(stream))
                                         : launch_gemm(x3, epilogue, kind, d->ops, f16, a, S(stream));
  return rc == -2 ? NMFMU_ERR_UNSUPPORTED : rc;
}

int nmfmu_pack2d(const float* src, int rows, int cols, int row_inner, int64_t row_outer_stride, int64_t row_inner_stride,
                 int col_inner, int64_t col_outer_stride, int64_t col_inner_stride, int rows_pad, int cols_pad,
                 float* dst_f32, void* dst_hi, void* dst_lo, uint32_t* flags, void* stream) {
  if (!src || rows <= 0 || cols <= 0 || rows_pad < rows || cols_pad < cols || cols_pad % 8 || row_inner <= 0 ||
      col_inner <= 0)
    return NMFMU_ERR_ARG;
  if (!dst_f32 && !dst_hi) return NMFMU_ERR_ARG;
  Pack2D p{src, rows, cols, row_inner, col_inner, row_outer_stride, row_inner_stride, col_outer_stride,
           col_inner_stride, rows_pad, cols_pad};
  const int grid = grid_for((int64_t)rows_pad * (cols_pad / 8));
  if (dst_f32) {
    hipLaunchKernelGGL(pack2d_kernel<false>, dim3(grid), dim3(256), 0, S(stream), p, dst_f32, nullptr, nullptr, flags);
    flags = nullptr;
  }
  if (dst_hi)
    hipLaunchKernelGGL(pack2d_kernel<true>, dim3(grid), dim3(256), 0, S(stream), p, nullptr, (uint16_t*)dst_hi,
                       (uint16_t*)dst_lo, flags);
  return (int)hipGetLastError();
}

int nmfmu_conv_unfold(const float* h, int batch, int rank, int lh, int taps, void* hu_hi, void* hu_lo, void* hut_hi,
                      void* hut_lo, int bl_pad, int rp_pad, void* stream) {
  if (!h || !hu_hi || !hut_hi || batch <= 0 || rank <= 0 || lh <= 0 || taps <= 0) return NMFMU_ERR_ARG;
  if (bl_pad < batch * (lh + taps - 1) || rp_pad < rank * taps || bl_pad % 8 || rp_pad % 8) return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)bl_pad * (rp_pad / 8) * 2;
  hipLaunchKernelGGL(conv_unfold_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), h, batch, rank, lh, taps,
                     (uint16_t*)hu_hi, (uint16_t*)hu_lo, (uint16_t*)hut_hi, (uint16_t*)hut_lo, bl_pad, rp_pad);
  return (int)hipGetLastError();
}

size_t nmfmu_conv_table_bytes(int batch, int rank, int lh, int taps) {
  if (batch <= 0 || rank <= 0 || lh <= 0 || taps <= 0) return 0;
  return (1 + (size_t)batch * rank * (lh + 2 * (size_t)taps - 2)) * 16;
}

int nmfmu_conv_tables(const float* h, int batch, int rank, int lh, int taps, void* rev_hi, void* rev_lo, void* fwd_hi,
                      void* fwd_lo, void* stream) {
  if (!h || !rev_hi || !fwd_hi || batch <= 0 || rank <= 0 || lh <= 0 || taps <= 0) return NMFMU_ERR_ARG;
  if (taps % 8 || (lh + taps - 1) % 8 || (rev_lo == nullptr) != (fwd_lo == nullptr)) return NMFMU_ERR_ARG;
  const int64_t n = 1 + (int64_t)batch * rank * (lh + 2 * taps - 2);
  hipLaunchKernelGGL(conv_tables_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), h, batch, rank, lh, taps,
                     (u32x4*)rev_hi, (u32x4*)rev_lo, (u32x4*)fwd_hi, (u32x4*)fwd_lo, 0);
  return (int)hipGetLastError();
}

int nmfmu_conv_tables_f16(const float* h, int batch, int rank, int lh, int taps, void* rev, void* fwd, void* stream) {
  if (!h || !rev || !fwd || batch <= 0 || rank <= 0 || lh <= 0 || taps <= 0 || taps % 8 || (lh + taps - 1) % 8) return NMFMU_ERR_ARG;
  const int64_t n = 1 + (int64_t)batch * rank * (lh + 2 * taps - 2);
  hipLaunchKernelGGL(conv_tables_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), h, batch, rank, lh, taps,
                     (u32x4*)rev, nullptr, (u32x4*)fwd, nullptr, 1);
  return (int)hipGetLastError();
}

int nmfmu_rank_sums(const float* src, int outer, int rank, int inner, float* part, float* out, void* stream) {
  if (!src || !out || !part || outer <= 0 || rank <= 0 || inner <= 0) return NMFMU_ERR_ARG;
  if ((int64_t)outer * inner <= (1 << 14)) {     // small (W of a short-kernel model): both stages in one launch
    hipLaunchKernelGGL(rank_sums_one_kernel, dim3(rank), dim3(1024), 0, S(stream), src, outer, rank, inner, out);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(rank_sums_partial_kernel, dim3(rank, kRankChunks), dim3(256), 0, S(stream), src, outer, rank, inner,
                     part);
  hipLaunchKernelGGL(rank_sums_final_kernel, dim3(rank), dim3(kRankChunks), 0, S(stream), part, out);
  return (int)hipGetLastError();
}

int nmfmu_conv_apply_w(float* w, int channels, int rank, int taps, const float* num, const float* den,
                       const float* kl_den, int rp_pad, float l1, float l2, float gamma, void* stream) {
  if (!w || !num || (!den && !kl_den) || rp_pad < rank * taps) return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)channels * rank * taps;
  hipLaunchKernelGGL(conv_apply_w_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), w, channels, rank * taps, taps, num,
                     den, kl_den, rp_pad, l1, l2, gamma);
  return (int)hipGetLastError();
}

static int conv_pack_w(float* w, int channels, int rank, int taps, const float* num, const float* den,
                       const float* kl_den, int c_pad, int rp_pad, float l1, float l2, float gamma, int update,
                       void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, const float* scale, void* stream,
                       const float* kl_hpart = nullptr, int n_hparts = 0, float* wcol = nullptr, int num_slabs = 1,
                       int f16 = 0, WkOut wk = WkOut{}) {
  if (!w || !wm_hi || !wmt_hi || channels <= 0 || rank <= 0 || taps <= 0) return NMFMU_ERR_ARG;
  if (update && (!num || (!den && !kl_den && !kl_hpart))) return NMFMU_ERR_ARG;
  if ((kl_hpart && n_hparts <= 0) || (wcol && scale) || ((kl_hpart || wcol) && taps < 64)) return NMFMU_ERR_ARG;
  if (c_pad < channels || rp_pad < (int64_t)rank * taps || c_pad % 64 || rp_pad % 64 || (wm_lo == nullptr) != (wmt_lo == nullptr))
    return NMFMU_ERR_ARG;
  if (f16 && wm_lo) return NMFMU_ERR_ARG;
  const dim3 grid(rp_pad / 64, c_pad / 64);
  if (wm_lo)
    hipLaunchKernelGGL(conv_apply_pack_w_kernel<true>, grid, dim3(256), 0, S(stream), w, channels, rank * taps, taps, num,
                       den, kl_den, c_pad, rp_pad, l1, l2, gamma, update, (uint16_t*)wm_hi, (uint16_t*)wm_lo,
                       (uint16_t*)wmt_hi, (uint16_t*)wmt_lo, scale, kl_hpart, n_hparts, wcol, num_slabs, 0, wk);
  else
    hipLaunchKernelGGL(conv_apply_pack_w_kernel<false>, grid, dim3(256), 0, S(stream), w, channels, rank * taps, taps, num,
                       den, kl_den, c_pad, rp_pad, l1, l2, gamma, update, (uint16_t*)wm_hi, nullptr, (uint16_t*)wmt_hi,
                       nullptr, scale, kl_hpart, n_hparts, wcol, num_slabs, f16, wk);
  return (int)hipGetLastError();
}

int nmfmu_conv_apply_pack_w(float* w, int channels, int rank, int taps, const float* num, const float* den,
                            const float* kl_den, int c_pad, int rp_pad, float l1, float l2, float gamma, int update,
                            void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, void* stream) {
  return conv_pack_w(w, channels, rank, taps, num, den, kl_den, c_pad, rp_pad, l1, l2, gamma, update, wm_hi, wm_lo, wmt_hi,
                     wmt_lo, nullptr, stream);
}

int nmfmu_conv_apply_pack_w_sums(float* w, int channels, int rank, int taps, const float* num, const float* den,
                                 const float* kl_den, const float* kl_hpart, int n_hparts, float* wcol, int num_slabs,
                                 int c_pad, int rp_pad, float l1, float l2, float gamma, int update, int precision,
                                 void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, void* stream) {
  if (num_slabs < 1 || num_slabs > 64) return NMFMU_ERR_ARG;
  if (precision != NMFMU_PREC_BF16 && precision != NMFMU_PREC_BF16X3 && precision != NMFMU_PREC_F16) return NMFMU_ERR_ARG;
  if ((precision == NMFMU_PREC_BF16X3) != (wm_lo != nullptr)) return NMFMU_ERR_ARG;
  return conv_pack_w(w, channels, rank, taps, num, den, kl_den, c_pad, rp_pad, l1, l2, gamma, update, wm_hi, wm_lo, wmt_hi,
                     wmt_lo, nullptr, stream, kl_hpart, n_hparts, wcol, num_slabs, precision == NMFMU_PREC_F16);
}

int nmfmu_conv_apply_pack_w_wk(float* w, int channels, int rank, int taps, const float* num, const float* den,
                               const float* kl_den, const float* kl_hpart, int n_hparts, float* wcol, int num_slabs,
                               int c_pad, int rp_pad, float l1, float l2, float gamma, int update, int precision,
                               void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, int taps_last, int fold, int wk_rows_pad,
                               int wk_k_pad, void* wk_hi, void* wk_lo, void* stream) {
  if (num_slabs < 1 || num_slabs > 64) return NMFMU_ERR_ARG;
  if (precision != NMFMU_PREC_BF16 && precision != NMFMU_PREC_BF16X3 && precision != NMFMU_PREC_F16) return NMFMU_ERR_ARG;
  if ((precision == NMFMU_PREC_BF16X3) != (wm_lo != nullptr)) return NMFMU_ERR_ARG;
  // (the arguments nmfmu_conv_pack_wk checks)
  if (!wk_hi || taps_last <= 0 || taps % taps_last || fold < 1 || taps_last % fold || wk_rows_pad < rank * fold || wk_k_pad % 8)
    return NMFMU_ERR_ARG;
  const int ck = (channels + 63) / 64;
  if ((int64_t)wk_k_pad < (int64_t)(taps / fold) * ck * 64) return NMFMU_ERR_ARG;
  if (precision == NMFMU_PREC_BF16X3 && !wk_lo) return NMFMU_ERR_ARG;
  WkOut wk;
  wk.hi = (uint16_t*)wk_hi, wk.lo = (uint16_t*)(precision == NMFMU_PREC_BF16X3 ? wk_lo : nullptr);
  wk.t_last = taps_last, wk.fold = fold, wk.ck = ck, wk.k_pad = wk_k_pad;
  return conv_pack_w(w, channels, rank, taps, num, den, kl_den, c_pad, rp_pad, l1, l2, gamma, update, wm_hi, wm_lo, wmt_hi,
                     wmt_lo, nullptr, stream, kl_hpart, n_hparts, wcol, num_slabs, precision == NMFMU_PREC_F16, wk);
}

int nmfmu_conv_pack_w_scaled(float* w, int channels, int rank, int taps, const float* scale, int c_pad, int rp_pad,
                             void* wm_hi, void* wm_lo, void* wmt_hi, void* wmt_lo, void* stream) {
  if (!scale) return NMFMU_ERR_ARG;
  return conv_pack_w(w, channels, rank, taps, nullptr, nullptr, nullptr, c_pad, rp_pad, 0.f, 0.f, 1.f, 0, wm_hi, wm_lo, wmt_hi,
                     wmt_lo, scale, stream);
}

int nmfmu_conv_fold_apply_h(float* h, int batch, int rank, int lh, int taps, const float* y_num, const float* y_den,
                            const float* kl_den, int bl_pad, float l1, float l2, float gamma, void* stream) {
  if (!h || !y_num || (!y_den && !kl_den) || bl_pad < batch * (lh + taps - 1)) return NMFMU_ERR_ARG;
  const int grid = batch * rank * ((lh + 63) / 64);
  hipLaunchKernelGGL(conv_fold_apply_h_kernel, dim3(grid), dim3(256), 0, S(stream), h, batch, rank, lh, taps,
                     y_num, y_den, kl_den, bl_pad, l1, l2, gamma);
  return (int)hipGetLastError();
}

int nmfmu_conv_ragged_supported(int rank, int taps) {
  return rank > 0 && taps > 0 && (size_t)8 * (2 * (size_t)taps + 63 + 8) * sizeof(float) <= 64 * 1024;
}

int nmfmu_conv_ragged_blocks(int batch, int lh, int taps) { return batch * ((lh + taps - 1 + 63) / 64); }

int nmfmu_gemm_ragged_supported(int ops, int m_pad, int n_pad, int extra) {
  // eight workgroups share out the 128 frames of an implicit-operand tile: >= 8 tiles along the explicit operand
  if (extra < 1 || extra > 16) return 0;
  if (ops == NMFMU_OPS_B_HU) return m_pad >= 8 * 128 && m_pad % 128 == 0;
  if (ops == NMFMU_OPS_A_HU) return n_pad >= 8 * 128 && n_pad % 128 == 0;
  return 0;
}

int nmfmu_conv_ragged_rows(const float* w, int channels, int rank, int taps, const float* h, int batch, int lh, int c0,
                           int precision, float beta, int mode, const float* x, int64_t ld, void* gn_hi, void* gn_lo,
                           void* gp_hi, void* gp_lo, float* loss_part, void* stream) {
  if (!w || !h || !x || channels <= 0 || batch <= 0 || lh <= 0 || c0 < 0 || c0 >= channels || mode < 0 || mode > 2)
    return NMFMU_ERR_ARG;
  if (!nmfmu_conv_ragged_supported(rank, taps)) return NMFMU_ERR_UNSUPPORTED;
  const int x3 = precision == NMFMU_PREC_BF16X3, f16 = precision == NMFMU_PREC_F16;
  if (precision != NMFMU_PREC_BF16 && !x3 && !f16) return NMFMU_ERR_UNSUPPORTED;
  const int kind = nmfmu_beta_kind(beta);
  if (mode == 2) {
    if (!loss_part) return NMFMU_ERR_ARG;
  } else {
    if (!gn_hi || (x3 && !gn_lo)) return NMFMU_ERR_ARG;
    if (kind != NMFMU_BETA_KL && (!gp_hi || (x3 && !gp_lo))) return NMFMU_ERR_ARG;
  }
  RaggedArgs a{w, h, channels, rank, taps, batch, lh, c0, x3, mode, f16, beta, x, ld, (uint16_t*)gn_hi, (uint16_t*)gn_lo,
               (uint16_t*)gp_hi, (uint16_t*)gp_lo, loss_part};
  const dim3 grid(nmfmu_conv_ragged_blocks(batch, lh, taps), channels - c0);
  const size_t lds = std::max<size_t>((size_t)8 * (2 * (size_t)taps + 63 + 8), 512) * sizeof(float);
  switch (kind) {
    case kKL: hipLaunchKernelGGL(conv_ragged_rows_kernel<kKL>, grid, dim3(512), lds, S(stream), a); break;
    case kEuc: hipLaunchKernelGGL(conv_ragged_rows_kernel<kEuc>, grid, dim3(512), lds, S(stream), a); break;
    case kIS: hipLaunchKernelGGL(conv_ragged_rows_kernel<kIS>, grid, dim3(512), lds, S(stream), a); break;
    default: hipLaunchKernelGGL(conv_ragged_rows_kernel<kGen>, grid, dim3(512), lds, S(stream), a); break;
  }
  return (int)hipGetLastError();
}

size_t nmfmu_fold_part_bytes(int m_pad, int n_pad) { return (size_t)(m_pad / 128) * (n_pad / 128) * 4 * 256 * sizeof(float); }

int nmfmu_fold_parts_supported(int batch, int rank, int lh, int taps) {
  return batch > 0 && rank > 0 && lh > 0 && taps >= 128 && lh + taps - 1 >= 128;
}

static int fold_parts_apply(float* h, int batch, int rank, int lh, int taps, const float* p_num, const float* p_den,
                            const float* kl_den, const float* kl_wcol, int c_tiles, int rp_pad, float* hsum_part, int bl_pad,
                            float l1, float l2, float gamma, void* stream, int m_pad = 0, int tail_rows = 0, int k_split = 1,
                            const FoldTables* tb = nullptr) {
  if (!h || !p_num || (!p_den && !kl_den && !kl_wcol) || bl_pad % 128 || bl_pad < batch * (lh + taps - 1)) return NMFMU_ERR_ARG;
  if (kl_wcol && (c_tiles <= 0 || rp_pad < rank * taps)) return NMFMU_ERR_ARG;
  if (!nmfmu_fold_parts_supported(batch, rank, lh, taps)) return NMFMU_ERR_UNSUPPORTED;
  const int grid = batch * rank * (tb ? (lh + 14 + kTabOwn - 1) / kTabOwn : (lh + 255) / 256);
  // the GEMM's tail-round split: tile rows >= m_pad / 128 - tail_rows come as k_split partial slabs
  int tail_tm0 = 1 << 30;
  size_t slab = 0;
  if (tail_rows > 0 && k_split > 1) {
    if (m_pad % 128 || tail_rows > m_pad / 128) return NMFMU_ERR_ARG;
    tail_tm0 = m_pad / 128 - tail_rows;
    slab = nmfmu_fold_part_bytes(m_pad, bl_pad) / sizeof(float);
  }
  if (tb)
    hipLaunchKernelGGL(conv_fold_parts_apply_h_kernel<true>, dim3(grid), dim3(256), 0, S(stream), h, batch, rank, lh, taps,
                       p_num, p_den, kl_den, kl_wcol, c_tiles, rp_pad, hsum_part, bl_pad / 128, l1, l2, gamma, tail_tm0,
                       k_split, slab, *tb);
  else
    hipLaunchKernelGGL(conv_fold_parts_apply_h_kernel<false>, dim3(grid), dim3(256), 0, S(stream), h, batch, rank, lh, taps,
                       p_num, p_den, kl_den, kl_wcol, c_tiles, rp_pad, hsum_part, bl_pad / 128, l1, l2, gamma, tail_tm0,
                       k_split, slab, FoldTables{});
  return (int)hipGetLastError();
}

int nmfmu_conv_fold_parts_apply_h(float* h, int batch, int rank, int lh, int taps, const float* p_num, const float* p_den,
                                  const float* kl_den, int bl_pad, float l1, float l2, float gamma, void* stream) {
  return fold_parts_apply(h, batch, rank, lh, taps, p_num, p_den, kl_den, nullptr, 0, 0, nullptr, bl_pad, l1, l2, gamma, stream);
}

int nmfmu_fold_hsum_parts(int batch, int lh) { return batch * ((lh + 255) / 256); }

int nmfmu_conv_fold_parts_apply_h_tail(float* h, int batch, int rank, int lh, int taps, const float* p_num,
                                       const float* p_den, const float* kl_den, const float* kl_wcol, int c_tiles,
                                       int rp_pad, float* hsum_part, int bl_pad, float l1, float l2, float gamma, int m_pad,
                                       int tail_rows, int k_split, void* stream) {
  return fold_parts_apply(h, batch, rank, lh, taps, p_num, p_den, kl_den, kl_wcol, c_tiles, rp_pad, hsum_part, bl_pad, l1, l2,
                          gamma, stream, m_pad, tail_rows, k_split);
}

int nmfmu_fold_hsum_parts_tables(int batch, int lh) { return batch * ((lh + 14 + kTabOwn - 1) / kTabOwn); }

int nmfmu_conv_fold_parts_apply_h_tables(float* h, const float* h_old, float* h_next, int batch, int rank, int lh, int taps,
                                         const float* p_num, const float* p_den, const float* kl_den, const float* kl_wcol,
                                         int c_tiles, int rp_pad, float* hsum_part, int bl_pad, float l1, float l2,
                                         float gamma, int m_pad, int tail_rows, int k_split, int precision, void* rev_hi,
                                         void* rev_lo, void* fwd_hi, void* fwd_lo, void* stream) {
  if (!h_old || !h_next || h_old == h_next || h_old == h || h_next == h || !rev_hi || !fwd_hi) return NMFMU_ERR_ARG;
  if (taps % 8 || (lh + taps - 1) % 8) return NMFMU_ERR_ARG;                       // (what nmfmu_conv_tables asks for)
  const bool x3 = precision == NMFMU_PREC_BF16X3;
  if (precision != NMFMU_PREC_BF16 && precision != NMFMU_PREC_F16 && !x3) return NMFMU_ERR_UNSUPPORTED;
  if (x3 != (rev_lo != nullptr) || x3 != (fwd_lo != nullptr)) return NMFMU_ERR_ARG;
  const FoldTables tb{h_old, h_next, (u32x4*)rev_hi, (u32x4*)rev_lo, (u32x4*)fwd_hi, (u32x4*)fwd_lo, precision == NMFMU_PREC_F16};
  return fold_parts_apply(h, batch, rank, lh, taps, p_num, p_den, kl_den, kl_wcol, c_tiles, rp_pad, hsum_part, bl_pad, l1, l2,
                          gamma, stream, m_pad, tail_rows, k_split, &tb);
}

int nmfmu_conv_fold_parts_apply_h_sums(float* h, int batch, int rank, int lh, int taps, const float* p_num,
                                       const float* p_den, const float* kl_den, const float* kl_wcol, int c_tiles,
                                       int rp_pad, float* hsum_part, int bl_pad, float l1, float l2, float gamma,
                                       void* stream) {
  return fold_parts_apply(h, batch, rank, lh, taps, p_num, p_den, kl_den, kl_wcol, c_tiles, rp_pad, hsum_part, bl_pad, l1, l2,
                          gamma, stream);
}

int nmfmu_convnd_unfold(const float* h, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps,
                        void* hu_hi, void* hu_lo, void* hut_hi, void* hut_lo, int bl_pad, int rp_pad, void* stream) {
  ConvGeom g;
  if (!h || !hu_hi || !hut_hi || batch <= 0 || rank <= 0 || make_geom(ndim, lh, taps, &g)) return NMFMU_ERR_ARG;
  if (bl_pad < (int64_t)batch * g.l_tot || rp_pad < (int64_t)rank * g.t_tot || bl_pad % 8 || rp_pad % 8 ||
      (hu_lo == nullptr) != (hut_lo == nullptr))
    return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)bl_pad * (rp_pad / 8) * 2;
  hipLaunchKernelGGL(convnd_unfold_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), h, batch, rank, g,
                     (uint16_t*)hu_hi, (uint16_t*)hu_lo, (uint16_t*)hut_hi, (uint16_t*)hut_lo, bl_pad, rp_pad);
  return (int)hipGetLastError();
}

int nmfmu_convnd_fold_apply_h(float* h, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps,
                              const float* y_num, const float* y_den, const float* kl_den, int bl_pad, float l1,
                              float l2, float gamma, void* stream) {
  ConvGeom g;
  if (!h || !y_num || (!y_den && !kl_den) || batch <= 0 || rank <= 0 || make_geom(ndim, lh, taps, &g))
    return NMFMU_ERR_ARG;
  if (bl_pad < (int64_t)batch * g.l_tot) return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)batch * rank * g.lh_tot;
  hipLaunchKernelGGL(convnd_fold_apply_h_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), h, batch, rank, g, y_num,
                     y_den, kl_den, bl_pad, l1, l2, gamma, (float*)nullptr);
  return (int)hipGetLastError();
}

int nmfmu_convnd_fold(float* out, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps, const float* y,
                      int bl_pad, void* stream) {
  ConvGeom g;
  if (!out || !y || batch <= 0 || rank <= 0 || make_geom(ndim, lh, taps, &g)) return NMFMU_ERR_ARG;
  if (bl_pad < (int64_t)batch * g.l_tot) return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)batch * rank * g.lh_tot;
  hipLaunchKernelGGL(convnd_fold_apply_h_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), out, batch, rank, g, y,
                     (const float*)nullptr, y /* non-null: skips the den loop */, bl_pad, 0.f, 0.f, 1.f, out);
  return (int)hipGetLastError();
}

int nmfmu_conv_pack_wk(const float* w, int channels, int rank, int taps, int taps_last, int fold, int rows_pad, int k_pad,
                       int precision, void* wk_hi, void* wk_lo, void* stream) {
  if (!w || !wk_hi || channels <= 0 || rank <= 0 || taps <= 0 || taps_last <= 0 || taps % taps_last || fold < 1 ||
      taps_last % fold || rows_pad < rank * fold || k_pad % 8)
    return NMFMU_ERR_ARG;
  const int ck = (channels + 63) / 64;
  if ((int64_t)k_pad < (int64_t)(taps / fold) * ck * 64) return NMFMU_ERR_ARG;
  if (precision == NMFMU_PREC_BF16X3 && !wk_lo) return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)rows_pad * (k_pad / 8);
  hipLaunchKernelGGL(conv_pack_wk_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), w, channels, rank, taps, taps_last, fold,
                     ck, rows_pad, k_pad, (uint16_t*)wk_hi, (uint16_t*)(precision == NMFMU_PREC_BF16X3 ? wk_lo : nullptr),
                     (int)(precision == NMFMU_PREC_F16));
  return (int)hipGetLastError();
}

int nmfmu_conv_apply_h_rows(float* h, int batch, int rank, int lh_outer, int lh_last, int fold, const float* num,
                            const float* den, const float* kl_den, int ld, float l1, float l2, float gamma, void* stream) {
  if (!h || !num || (!den && !kl_den) || batch <= 0 || rank <= 0 || lh_outer <= 0 || lh_last <= 0 || fold < 1 ||
      ld < rank * fold)
    return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)batch * lh_outer * lh_last * rank;
  hipLaunchKernelGGL(conv_apply_h_rows_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), h, batch, rank, lh_outer, lh_last,
                     fold, num, den, kl_den, ld, l1, l2, gamma, (float*)nullptr);
  return (int)hipGetLastError();
}

int nmfmu_conv_h_rows_parts(int batch, int rank, int lh_outer, int lh_last) {
  if (batch <= 0 || rank <= 0 || rank > 256 || lh_outer <= 0 || lh_last <= 0) return 0;
  const int64_t npos = (int64_t)batch * lh_outer * lh_last;
  const int p = 256 / rank;
  return (int)std::max<int64_t>(1, std::min<int64_t>((npos + p - 1) / p, 2048));
}

int nmfmu_conv_apply_h_rows_sums(float* h, int batch, int rank, int lh_outer, int lh_last, int fold, const float* num,
                                 const float* kl_wcol, int c_tiles, int rp_pad, int taps, int ld, float l1, float l2, float gamma,
                                 float* hsum_part, void* stream) {
  if (!h || !num || !kl_wcol || !hsum_part || batch <= 0 || rank <= 0 || rank > 256 || lh_outer <= 0 || lh_last <= 0 || fold < 1 ||
      ld < rank * fold || c_tiles <= 0 || taps < 64 || rp_pad % 64 || rp_pad < rank * taps)
    return NMFMU_ERR_ARG;
  const int grid = nmfmu_conv_h_rows_parts(batch, rank, lh_outer, lh_last);
  hipLaunchKernelGGL(conv_apply_h_rows_sums_kernel, dim3(grid), dim3(256), 0, S(stream), h, batch, rank, lh_outer, lh_last, fold,
                     num, kl_wcol, c_tiles, rp_pad / 64, taps, ld, l1, l2, gamma, hsum_part);
  return (int)hipGetLastError();
}

size_t nmfmu_convnd_table_bytes(int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps) {
  ConvGeom g;
  if (batch <= 0 || rank <= 0 || make_geom(ndim, lh, taps, &g)) return 0;
  size_t n = (size_t)batch * rank;
  for (int d = 0; d < 3; ++d) n *= (size_t)(g.l[d] + g.t[d] - 1);
  return (1 + n) * 16;
}

int nmfmu_convnd_tables(const float* h, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps, int precision,
                        void* rev_hi, void* rev_lo, void* fwd_hi, void* fwd_lo, void* stream) {
  ConvGeom g;
  if (!h || !rev_hi || !fwd_hi || batch <= 0 || rank <= 0 || make_geom(ndim, lh, taps, &g)) return NMFMU_ERR_ARG;
  if (g.t[2] % 8 || g.l[2] % 8 || (rev_lo == nullptr) != (fwd_lo == nullptr)) return NMFMU_ERR_ARG;
  if ((precision == NMFMU_PREC_BF16X3) != (rev_lo != nullptr) || precision == NMFMU_PREC_F16X) return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)(nmfmu_convnd_table_bytes(batch, rank, ndim, lh, taps) / 16);
  if (n > INT32_MAX) return NMFMU_ERR_ARG;     // chunk indices are 32-bit in the GEMM
  hipLaunchKernelGGL(convnd_tables_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), h, batch, rank, g, (u32x4*)rev_hi,
                     (u32x4*)rev_lo, (u32x4*)fwd_hi, (u32x4*)fwd_lo, (int)(precision == NMFMU_PREC_F16));
  return (int)hipGetLastError();
}

int nmfmu_convnd_koff(int ops, int batch, int rank, int ndim, const int32_t* lh, const int32_t* taps, int k_pad,
                      int32_t* koff_host) {
  ConvGeom g;
  if (!koff_host || batch <= 0 || rank <= 0 || k_pad <= 0 || k_pad % 8 || make_geom(ndim, lh, taps, &g)) return NMFMU_ERR_ARG;
  if (g.t[2] % 8 || g.l[2] % 8) return NMFMU_ERR_ARG;
  const int64_t jj2 = g.l[2] + g.t[2] - 1, jj1 = g.l[1] + g.t[1] - 1, jj0 = g.l[0] + g.t[0] - 1, jjt = jj0 * jj1 * jj2;
  if (1 + (int64_t)batch * rank * jjt > INT32_MAX) return NMFMU_ERR_ARG;
  for (int kc = k_pad / 8; kc < k_pad / 8 + 8; ++kc) koff_host[kc] = INT32_MIN;   // spare: the kernel reads one k-tile ahead
  if (ops == NMFMU_OPS_B_HU || ops == NMFMU_OPS_A_HU) {           // k = (r, t0, t1, t2): chunk = 8 consecutive t2
    for (int kc = 0; kc < k_pad / 8; ++kc) {
      const int k0 = kc * 8, r = k0 / g.t_tot, tf = k0 - r * g.t_tot;
      const int t2 = tf % g.t[2], t01 = tf / g.t[2], t1 = t01 % g.t[1], t0 = t01 / g.t[1];
      koff_host[kc] = r < rank ? (int32_t)(1 + r * jjt - ((int64_t)t0 * jj1 + t1) * jj2 - t2) : INT32_MIN;
    }
  } else if (ops == NMFMU_OPS_B_HUT) {                            // k = (b, l0, l1, l2): chunk = 8 consecutive l2
    for (int kc = 0; kc < k_pad / 8; ++kc) {
      const int k0 = kc * 8, b = k0 / g.l_tot, lf = k0 - b * g.l_tot;
      const int l2 = lf % g.l[2], l01 = lf / g.l[2], l1 = l01 % g.l[1], l0 = l01 / g.l[1];
      koff_host[kc] = b < batch ? (int32_t)(1 + (int64_t)b * rank * jjt + ((int64_t)l0 * jj1 + l1) * jj2 + l2) : INT32_MIN;
    }
  } else {
    return NMFMU_ERR_ARG;
  }
  return NMFMU_OK;
}

int nmfmu_slab_sum(float* slabs, int64_t slab_elems, int nslab, void* stream) {
  if (!slabs || slab_elems <= 0 || slab_elems % 4 || nslab < 1) return NMFMU_ERR_ARG;
  if (nslab == 1) return NMFMU_OK;
  hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for(slab_elems / 4)), dim3(256), 0, S(stream), slabs, slab_elems / 4, nslab);
  return (int)hipGetLastError();
}

int nmfmu_conv_rows_fold(float* out, int batch, int rank, int lh_outer, int lh_last, int fold, const float* num, int ld,
                         void* stream) {
  if (!out || !num || batch <= 0 || rank <= 0 || lh_outer <= 0 || lh_last <= 0 || fold < 1 || ld < rank * fold)
    return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)batch * lh_outer * lh_last * rank;
  hipLaunchKernelGGL(conv_apply_h_rows_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), out, batch, rank, lh_outer, lh_last,
                     fold, num, (const float*)nullptr, num /* non-null: closed form, unused */, ld, 0.f, 0.f, 1.f, out);
  return (int)hipGetLastError();
}

}  // extern "C"
