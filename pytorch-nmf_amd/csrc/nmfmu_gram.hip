// Gram matrix of a factor from its 16-bit operand image, for the beta == 2 path without reconstruction (kModeXB of
// nmfmu_fused.h; reference: nmf.py:61-63 -- no eps inside the grad_outputs, so the denominator of the update is
// owner @ (panel^T panel) exactly and the N x C reconstruction is never needed).
//
//   gram_partial_kernel : G_chunk = sum over the chunk's 64-row tiles of P2_tile P2_tile^T  (MFMA 32x32x16, fp32 accumulate).
//                         Both operands come from the SAME transposed image tile ([R_PAD ranks][64 rows], nmfmu_layout.h):
//                         lane (j, hl) of tile-row ta reads the 16-byte slot (rank 32 ta + j, rows 16 kk + 8 hl ..) -- the
//                         fragment the second GEMM of the fused kernels reads from LDS; here straight from L2 / HBM
//                         (every byte of the image is used once per wave).
//   gram_finalize_kernel: fixed-order sum of the chunk partials (deterministic), the fp32 matrix for the apply kernel, and
//                         the 16-bit hi / lo images + per-row power-of-two scales for the fused-apply epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nmfmu.h"
#include "nmfmu_fused.h"

namespace nmfmu {

constexpr int kGramMaxChunks = 256;   // one chunk per CU for long panels (4 tiles each at 65536 rows)
constexpr int kGramTilesPerChunk = 4;

template <int R_PAD, int OPT>
__global__ void __launch_bounds__(256) gram_partial_kernel(const char* __restrict__ p2, int ktiles, int nchunk,
                                                           float* __restrict__ part) {
  constexpr int RT = R_PAD / 32;
  constexpr int NTR = (RT + 3) / 4;   // tile rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, hl = lane >> 5;
  const int ch = blockIdx.x;
  const int per = (ktiles + nchunk - 1) / nchunk;
  const int t0 = ch * per, t1 = min(ktiles, t0 + per);
  f32x16 acc[NTR][RT];
#pragma unroll
  for (int a = 0; a < NTR; ++a)
#pragma unroll
    for (int b = 0; b < RT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  // byte offset of this lane's slot for k-step kk inside rank row r of a tile: r * 128 + ((2 kk + hl) ^ ((r >> 1) & 7)) * 16
  auto slot = [&](int r, int kk) { return r * 128 + (((2 * kk + hl) ^ ((r >> 1) & 7)) << 4); };
  if constexpr (R_PAD <= 128) {
    // (round 5) one tile-row per wave (NTR == 1): the 4 * (RT + 1) fragment loads of a tile are issued together and the
    // next tile's go out before this tile's MFMAs (two register sets) -- a chunk of four tiles was four HBM latencies in
    // a row (10 us per launch, twice per beta = 2 iteration); the MFMA order per accumulator is unchanged
    static_assert(NTR == 1, "one tile row per wave");
    u32x4 fa[2][4], fb[2][4][RT];
    const bool live = wave < RT;
    auto ld = [&](int t, auto bc) {
      constexpr int b = decltype(bc)::value;
      const char* tile = p2 + (size_t)t * (R_PAD * 128);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        fa[b][kk] = ld16(tile + slot(32 * (live ? wave : 0) + j, kk));
#pragma unroll
        for (int q = 0; q < RT; ++q) fb[b][kk][q] = ld16(tile + slot(32 * q + j, kk));
      }
    };
    auto mm = [&](auto bc) {
      constexpr int b = decltype(bc)::value;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int q = 0; q < RT; ++q) acc[0][q] = mfma_op<OPT>(fa[b][kk], fb[b][kk][q], acc[0][q]);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    if (live && t0 < t1) {
      ld(t0, B0{});
      int t = t0;
      for (; t + 2 <= t1; t += 2) {
        ld(t + 1, B1{});
        mm(B0{});
        if (t + 2 < t1) ld(t + 2, B0{});
        mm(B1{});
      }
      if (t < t1) mm(B0{});
    }
  } else
  for (int t = t0; t < t1; ++t) {
    const char* tile = p2 + (size_t)t * (R_PAD * 128);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4 fb[RT];
#pragma unroll
      for (int b = 0; b < RT; ++b) fb[b] = ld16(tile + slot(32 * b + j, kk));
#pragma unroll
      for (int a = 0; a < NTR; ++a) {
        const int ta = wave + 4 * a;
        if (ta < RT) {
          const u32x4 fa = ld16(tile + slot(32 * ta + j, kk));
#pragma unroll
          for (int b = 0; b < RT; ++b) acc[a][b] = mfma_op<OPT>(fa, fb[b], acc[a][b]);
        }
      }
    }
  }
  // accumulator register e of lane (j, hl): row 32 ta + (e & 3) + 8 (e >> 2) + 4 hl, column 32 b + j
  float* out = part + (size_t)ch * R_PAD * R_PAD;
#pragma unroll
  for (int a = 0; a < NTR; ++a) {
    const int ta = wave + 4 * a;
    if (ta < RT) {
#pragma unroll
      for (int b = 0; b < RT; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          out[(size_t)(32 * ta + (e & 3) + 8 * (e >> 2) + 4 * hl) * R_PAD + 32 * b + j] = acc[a][b][e];
    }
  }
}

// one workgroup per matrix row r (= column r: the matrix is symmetric).  The chunk partials of the row are summed by
// 256 / (r_pad / 4) thread groups in parallel (float4 per thread, eight independent loads in flight), the groups' sums are
// combined in a fixed order: deterministic, and latency-bound on ~nchunk / 64 rounds instead of nchunk loads in a row.
// Afterwards thread q owns G[r][q].
__global__ void __launch_bounds__(256) gram_finalize_kernel(const float* __restrict__ part, int nchunk, int r_pad,
                                                            float* __restrict__ gram, uint16_t* __restrict__ g_hi,
                                                            uint16_t* __restrict__ g_lo, float* __restrict__ g_scale,
                                                            int f16) {
  __shared__ float red[4];
  __shared__ float4 gsum[256];
  __shared__ float rowv[256];
  const int r = blockIdx.x, q = threadIdx.x;
  {
    const int r4 = r_pad / 4, ngrp = 256 / r4;            // r_pad = 32 .. 256 -> 32 .. 4 groups
    const int c4 = q % r4, grp = q / r4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* src = reinterpret_cast<const float4*>(part + (size_t)r * r_pad) + c4;
    const size_t cstride = (size_t)r_pad * r_pad / 4;
    int ch = grp;
    for (; ch + 15 * ngrp < nchunk; ch += 16 * ngrp) {   // (same order of additions as the rounds of eight below)
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(ch + u * ngrp) * cstride];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
    }
    for (; ch + 7 * ngrp < nchunk; ch += 8 * ngrp) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(ch + u * ngrp) * cstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
    }
    for (; ch < nchunk; ch += ngrp) {
      const float4 v = src[(size_t)ch * cstride];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    gsum[q] = acc;
    __syncthreads();
    if (q < r4) {
      float4 t = gsum[q];
      for (int g = 1; g < ngrp; ++g) {
        const float4 v = gsum[g * r4 + q];
        t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
      }
      rowv[4 * q] = t.x, rowv[4 * q + 1] = t.y, rowv[4 * q + 2] = t.z, rowv[4 * q + 3] = t.w;
    }
    __syncthreads();
  }
  const float s = q < r_pad ? rowv[q] : 0.f;
  if (q < r_pad) gram[(size_t)r * r_pad + q] = s;
  // row maximum -> power-of-two scale that puts the row's largest entry at ~2^10 (fp16: normal range with headroom for the
  // MFMA's fp32 accumulation of <= 256 terms is not an issue -- the accumulator is fp32; bf16 needs no scale, gets one anyway)
  float m = s;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((q & 63) == 0) red[q >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int ex = 0;
  if (m > 0.f && m < 3.0e38f) {
    (void)frexpf(m, &ex);      // m = f * 2^ex, f in [0.5, 1)
    ex -= 10;                   // scaled maximum in [2^9, 2^10)
  }
  const float down = ldexpf(1.f, -ex);
  if (q == 0) g_scale[r] = ldexpf(1.f, ex);
  if (q < r_pad) {
    const float v = s * down;
    uint16_t h, l;
    if (f16) {
      const _Float16 hh = (_Float16)v;
      const _Float16 ll = (_Float16)(v - (float)hh);
      h = __builtin_bit_cast(uint16_t, hh), l = __builtin_bit_cast(uint16_t, ll);
    } else {
      const uint32_t hp = pack_bf16(v, 0.f);
      const uint32_t lp = pack_bf16(v - bf16_lo(hp), 0.f);
      h = (uint16_t)(hp & 0xffffu), l = (uint16_t)(lp & 0xffffu);
    }
    g_hi[(size_t)r * r_pad + q] = h;
    g_lo[(size_t)r * r_pad + q] = l;
  }
}

template <int R_PAD>
int launch_gram_partial(int f16, const char* p2, int ktiles, int nchunk, float* part, hipStream_t s) {
  if (f16) hipLaunchKernelGGL((gram_partial_kernel<R_PAD, kOpF16>), dim3(nchunk), dim3(256), 0, s, p2, ktiles, nchunk, part);
  else hipLaunchKernelGGL((gram_partial_kernel<R_PAD, kOpBf16>), dim3(nchunk), dim3(256), 0, s, p2, ktiles, nchunk, part);
  return (int)hipGetLastError();
}

}  // namespace nmfmu

using namespace nmfmu;

extern "C" {

size_t nmfmu_gram_ws_bytes(int r_pad) { return r_pad > 0 ? (size_t)kGramMaxChunks * r_pad * r_pad * 4 : 0; }

int nmfmu_gram_panel(const nmfmu_factor* panel, int r_pad, int precision, void* ws, float* gram, void* g_hi, void* g_lo,
                     float* g_scale, void* stream) {
  if (!panel || !panel->p2_hi || !ws || !gram || !g_hi || !g_lo || !g_scale) return NMFMU_ERR_ARG;
  if (panel->rows_pad <= 0 || panel->rows_pad % 64) return NMFMU_ERR_ARG;
  if (precision != NMFMU_PREC_BF16 && precision != NMFMU_PREC_F16 && precision != NMFMU_PREC_F16X) return NMFMU_ERR_UNSUPPORTED;
  const int f16 = precision != NMFMU_PREC_BF16;
  const int ktiles = panel->rows_pad / 64;
  // one tile per chunk while that stays within kGramMaxChunks (a 4096-row factor: 64 workgroups and one HBM latency
  // instead of 16 workgroups and four), four tiles per chunk at 65536 rows
  int nchunk = ktiles <= kGramMaxChunks ? ktiles : (ktiles + kGramTilesPerChunk - 1) / kGramTilesPerChunk;
  if (nchunk > kGramMaxChunks) nchunk = kGramMaxChunks;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const char* p2 = static_cast<const char*>(panel->p2_hi);
  float* part = static_cast<float*>(ws);
  int e;
  switch (r_pad) {
    case 32: e = launch_gram_partial<32>(f16, p2, ktiles, nchunk, part, s); break;
    case 64: e = launch_gram_partial<64>(f16, p2, ktiles, nchunk, part, s); break;
    case 128: e = launch_gram_partial<128>(f16, p2, ktiles, nchunk, part, s); break;
    case 256: e = launch_gram_partial<256>(f16, p2, ktiles, nchunk, part, s); break;
    default: return NMFMU_ERR_UNSUPPORTED;
  }
  if (e) return e;
  hipLaunchKernelGGL(gram_finalize_kernel, dim3(r_pad), dim3(256), 0, s, part, nchunk, r_pad, gram,
                     static_cast<uint16_t*>(g_hi), static_cast<uint16_t*>(g_lo), g_scale, f16);
  return (int)hipGetLastError();
}

}  // extern "C"
