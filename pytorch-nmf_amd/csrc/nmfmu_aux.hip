// Memory-bound companions of the fused MU kernel: packing, the MU "apply" step, reductions, probes.
// All of them are HBM/latency bound; they are written for coalesced 16-byte traffic and deterministic
// (fixed-order) reductions, not for MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "nmfmu_aux.h"
#include "nmfmu_pp.h"

namespace nmfmu {

// ------------------------------------------------------------------------------------------------------------
// pack_x: V fp32 (row-major, ld) -> fragment-order X (bf16 or fp32), zero padded.  One thread = one 16-byte chunk.
// Fused with the validation passes of nmf.py:329-336 (any(v < 0 or NaN), min(v)).
// ------------------------------------------------------------------------------------------------------------
// NMFMU_PREC_F16R: the target rounded (to nearest, ties to even) to the top 24 bits of its fp32 word -- sign, 8 exponent bits,
// 15 mantissa bits: 16 significant bits, relative error <= 2^-16, fp32's whole range (subnormals included).  A value that
// would round up to infinity is truncated instead.  The kernels put the word back together with ONE v_perm_b32 per element.
__device__ __forceinline__ uint32_t round24(float x) {
  const uint32_t b = __builtin_bit_cast(uint32_t, x);
  uint32_t r = b + 0x7fu + ((b >> 8) & 1u);
  if ((r & 0x7f800000u) == 0x7f800000u && (b & 0x7f800000u) != 0x7f800000u) r = b;
  return r & 0xffffff00u;
}

template <int FMT, bool TRANSPOSE>
__global__ void __launch_bounds__(256) pack_x_kernel(const float* __restrict__ v, int64_t ld, int rows, int cols,
                                                     void* __restrict__ xp, int ktiles, int64_t nchunks,
                                                     uint32_t* flags, int G) {
  constexpr bool FP32 = FMT == 1;
  constexpr bool F16R = FMT == 3;   // the fp32 rounded to its top 24 bits: bits 31..16 in chunks 0..3, bits 15..8 in chunks 4, 5
  constexpr int NQ = FP32 ? 8 : (F16R ? 6 : 4);
  constexpr int EPC = FP32 ? 4 : 8;
  const int M = TRANSPOSE ? cols : rows;  // owner axis length
  const int K = TRANSPOSE ? rows : cols;
  uint32_t bad = 0, mn = 0x7f800000u;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * 256) {
    const int lane = (int)(c & 63);
    int64_t r = c >> 6;
    const int q = (int)(r % NQ);
    r /= NQ;
    const int g = (int)(r % G);
    r /= G;
    const int w = (int)(r & 3);
    r >>= 2;
    const int64_t kt = r % ktiles;
    const int64_t mb = r / ktiles;
    const int64_t m = mb * (128 * G) + w * (32 * G) + g * 32 + (lane & 31);
    if constexpr (F16R) {
      if (q >= 4) {
        // third bytes (bits 15..8 of the rounded word) of elements 16 (q - 4) .. + 15 of this lane's 32 columns
        u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t k = kt * 64 + 32 * (lane >> 5) + 16 * (q - 4) + i;
          float x = 0.f;
          if (m < M && k < K) x = TRANSPOSE ? v[k * ld + m] : v[m * ld + k];
          o[i >> 2] |= ((round24(x) >> 8) & 0xffu) << (8 * (i & 3));
        }
        reinterpret_cast<u32x4*>(xp)[c] = o;
        continue;    // (validation flags are taken by the head chunks, which see every element once)
      }
    }
    const int64_t k0 = kt * 64 + 32 * (lane >> 5) + (int64_t)q * EPC;
    float e[EPC];
#pragma unroll
    for (int i = 0; i < EPC; ++i) {
      const int64_t k = k0 + i;
      float x = 0.f;
      if (m < M && k < K) {
        x = TRANSPOSE ? v[k * ld + m] : v[m * ld + k];
        bad |= !(x >= 0.f) ? 1u : 0u;
        mn = min(mn, __builtin_bit_cast(uint32_t, x) & 0x7fffffffu);
      }
      e[i] = x;
    }
    u32x4 o;
    if constexpr (FP32) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = __builtin_bit_cast(uint32_t, e[i]);
    } else if constexpr (F16R) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (round24(e[2 * i]) >> 16) | (round24(e[2 * i + 1]) & 0xffff0000u);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = pack_img(e[2 * i], e[2 * i + 1], FMT == 2);
    }
    reinterpret_cast<u32x4*>(xp)[c] = o;
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

int launch_pack_x(const float* v, int64_t ld, int rows, int cols, bool transpose, int fmt, void* xp, int m_pad,
                  int k_pad, uint32_t* flags, int G, hipStream_t s) {
  const bool fp32 = fmt == 1;
  const int ktiles = k_pad / kBK;
  const int64_t nchunks = (int64_t)m_pad * k_pad * (fp32 ? 4 : (fmt == 3 ? 3 : 2)) / 16;
  const int grid = (int)std::min<int64_t>((nchunks + 255) / 256, 256 * 32);
#define L(F, T) hipLaunchKernelGGL((pack_x_kernel<F, T>), dim3(grid), dim3(256), 0, s, v, ld, rows, cols, xp, ktiles, nchunks, flags, G)
  if (fmt == 3 && transpose) L(3, true);
  else if (fmt == 3) L(3, false);
  else if (fp32 && transpose) L(1, true);
  else if (fp32) L(1, false);
  else if (fmt == 2 && transpose) L(2, true);
  else if (fmt == 2) L(2, false);
  else if (transpose) L(0, true);
  else L(0, false);
#undef L
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// apply: nmf.py:78-92 on a 64-row stripe of the owner factor, then re-emit that stripe's bf16 images (P1 rows,
// one whole P2 tile) and its partial column sums.  PACK_ONLY skips the update (initial packing of W0 / H0).
// ------------------------------------------------------------------------------------------------------------
// (round 5) NT threads per block: 512 when the update phase has at least 512 (row, four-rank) items, so that every thread
// has ONE item and all of a block's slab / master loads are in flight together -- with 256 threads the short-factor
// instance (16 rows x 128 ranks) ran two items per thread one after the other, i.e. two HBM round trips per launch.
template <int ROWS, int R_PAD>
constexpr int apply_threads() { return ROWS * (R_PAD / 4) >= 512 ? 512 : 256; }

template <int R_PAD, bool X3, bool PACK_ONLY, int ROWS, int NT>
__global__ void __launch_bounds__(NT) apply_kernel(ApplyArgs a) {
  // ROWS = 64 (one whole P2 tile per block) for tall factors, 16 for short ones so that the grid still fills the chip.
  constexpr int LDT = R_PAD + 1;  // odd leading dimension: the P2 column reads below stay <= 2-way bank conflicted
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);  // [ROWS][LDT]
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * ROWS;
  const size_t plane = (size_t)a.rows_pad * R_PAD;
  constexpr int R4 = R_PAD / 4;

  // trainer mode with an orthogonality penalty needs p.sum(1) of the OLD factor (trainer.py:105-106): stage the old
  // rows in the tile first, reduce each row in a fixed order
  float* rowsum = tile + ROWS * LDT;  // [ROWS]
  if constexpr (!PACK_ONLY) {
    if (a.trainer && a.ortho > 0.f) {
      for (int idx = tid; idx < ROWS * R_PAD; idx += NT) {
        const int rl = idx / R_PAD, r = idx - rl * R_PAD;
        const int row = row0 + rl;
        tile[rl * LDT + r] = (row < a.rows && r < a.rank) ? a.f[(size_t)row * a.rank + r] : 0.f;
      }
      __syncthreads();
      for (int rl = tid; rl < ROWS; rl += NT) {
        float s = 0.f;
        for (int r = 0; r < a.rank; ++r) s += tile[rl * LDT + r];
        rowsum[rl] = s;
      }
      __syncthreads();
    }
  }
  for (int idx = tid; idx < ROWS * R4; idx += NT) {
    const int rl = idx / R4, r = (idx - rl * R4) * 4;
    const int row = row0 + rl;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < a.rows && r < a.rank) {
      const float* fp = a.f + (size_t)row * a.rank + r;
      const int nv = min(4, a.rank - r);
      for (int i = 0; i < nv; ++i) f[i] = fp[i];
      if constexpr (PACK_ONLY) {
        if (a.scale)
          for (int i = 0; i < nv; ++i) f[i] *= a.scale[r + i];
      }
      if constexpr (!PACK_ONLY) {
        const size_t e = (size_t)row * R_PAD + r;
        // k-split partials: eight independent loads in flight per lane (the slabs are read exactly once: HBM latency,
        // not bandwidth, bounds a dependent chain), combined in a fixed order
        float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
        int s0 = 0;
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        for (; s0 + 16 <= a.nslab; s0 += 16) {   // (same order of additions as the rounds of eight)
          f32x4_t v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(a.num + (size_t)(s0 + u) * plane + e));
#pragma unroll
          for (int u = 0; u < 16; ++u) n4.x += v[u].x, n4.y += v[u].y, n4.z += v[u].z, n4.w += v[u].w;
        }
        for (; s0 + 8 <= a.nslab; s0 += 8) {
          f32x4_t v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(a.num + (size_t)(s0 + u) * plane + e));
#pragma unroll
          for (int u = 0; u < 8; ++u) n4.x += v[u].x, n4.y += v[u].y, n4.z += v[u].z, n4.w += v[u].w;
        }
        for (; s0 < a.nslab; ++s0) {
          const float4 v = *reinterpret_cast<const float4*>(a.num + (size_t)s0 * plane + e);
          n4.x += v.x, n4.y += v.y, n4.z += v.z, n4.w += v.w;
        }
        float neg[4] = {n4.x, n4.y, n4.z, n4.w};
        float pos[4];
        const float den_eps = a.trainer ? 0.f : kEps;  // the trainer adds eps after the penalties (trainer.py:108)
        if (a.kl_den) {  // closed form, no relu / eps (nmf.py:80 branch skipped; trainer.py:84 ones-backward)
          const float4 d4 = *reinterpret_cast<const float4*>(a.kl_den + r);
          pos[0] = d4.x, pos[1] = d4.y, pos[2] = d4.z, pos[3] = d4.w;
        } else {
          float4 d4 = *reinterpret_cast<const float4*>(a.den + e);
          const int dslab = a.den_nslab > 0 ? a.den_nslab : a.nslab;
          for (int s = 1; s < dslab; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(a.den + s * plane + e);
            d4.x += v.x, d4.y += v.y, d4.z += v.z, d4.w += v.w;
          }
          pos[0] = fmaxf(d4.x, 0.f) + den_eps, pos[1] = fmaxf(d4.y, 0.f) + den_eps;  // nmf.py:83
          pos[2] = fmaxf(d4.z, 0.f) + den_eps, pos[3] = fmaxf(d4.w, 0.f) + den_eps;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < nv) {
            const float ngr = fmaxf(neg[i], 0.f);
            const float ng = ngr + kEps;        // nmf.py:78, trainer.py:109
            float ps = pos[i];
            if (a.trainer && a.grad) a.grad[(size_t)row * a.rank + r + i] = ps - ngr;  // trainer.py:98
            if (a.l1 > 0.f) ps += a.l1;         // nmf.py:85-86, trainer.py:100-101
            if (a.l2 > 0.f) ps += a.l2 * f[i];  // nmf.py:87-88, trainer.py:102-103
            if (a.trainer) {
              if (a.ortho > 0.f) ps += a.ortho * (rowsum[rl] - f[i]);  // trainer.py:105-106
              ps += kEps;                                              // trainer.py:108
            }
            float mult = ng / ps;
            if (a.gamma != 1.f) mult = powf(mult, a.gamma);
            f[i] *= mult;
            a.f[(size_t)row * a.rank + r + i] = f[i];
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[rl * LDT + r + i] = f[i];
  }
  __syncthreads();

  // P1: ROWS rows x (R_PAD/8) sixteen-byte slots, swizzled inside each row
  constexpr int SP = R_PAD / 8;
  bool clamped = false;   // fp16 images: a factor value above 65504 is stored as 65504 -- tell the caller (status word)
  for (int idx = tid; idx < ROWS * SP; idx += NT) {
    const int rl = idx / SP, slot = idx - rl * SP;
    const float* src = tile + rl * LDT + slot * 8;
    u32x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x0 = src[2 * i], x1 = src[2 * i + 1];
      clamped |= fmaxf(x0, x1) > 65504.f;
      const uint32_t h = pack_img(x0, x1, a.f16);
      hi[i] = h;
      lo[i] = pack_bf16(x0 - bf16_lo(h), x1 - bf16_hi(h));
    }
    const int64_t off = p1_offset(row0 + rl, slot * 8, R_PAD);
    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.p1_hi) + off) = hi;
    if constexpr (X3) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.p1_lo) + off) = lo;
  }
  if (a.f16 && a.status && __any(clamped) && (tid & 63) == 0) atomicOr(a.status, 1u);
  // P2: [R_PAD][64] tiles; this block owns ROWS/8 of the 8 slots (8 consecutive factor rows each) of every rank row
  constexpr int NS = ROWS / 8;
  for (int idx = tid; idx < (a.p2_hi ? R_PAD * NS : 0); idx += NT) {   // (p2_hi == nullptr: NMFMU_STAGE_DMA_NOP2, nobody reads it)
    const int r = idx / NS, sl = idx - r * NS;
    u32x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x0 = tile[(sl * 8 + 2 * i) * LDT + r], x1 = tile[(sl * 8 + 2 * i + 1) * LDT + r];
      const uint32_t h = pack_img(x0, x1, a.f16);
      hi[i] = h;
      lo[i] = pack_bf16(x0 - bf16_lo(h), x1 - bf16_hi(h));
    }
    const int64_t off = p2_offset(row0 + sl * 8, r, R_PAD);
    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.p2_hi) + off) = hi;
    if constexpr (X3) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.p2_lo) + off) = lo;
  }
  // partial column sums of this stripe (fixed order -> deterministic)
  for (int r = tid; r < R_PAD; r += NT) {
    float s = 0.f;
#pragma unroll 8
    for (int rl = 0; rl < ROWS; ++rl) s += tile[rl * LDT + r];
    a.colsum_part[(size_t)blockIdx.x * R_PAD + r] = s;
  }
}

// colsum[r] = sum_b part[b][r].  One block per 32 columns: 8 float4 column groups x 32 row groups, combined in a
// fixed order (deterministic), each thread striding over the partials so a 1024-stripe factor needs 32 loads per thread.
__global__ void __launch_bounds__(256) colsum_finalize_kernel(const float* __restrict__ part, int nblk, int r_pad,
                                                              float* __restrict__ out) {
  __shared__ float4 red[32][8];
  const int c4 = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int col = blockIdx.x * 32 + c4 * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int b = g;
  // eight independent loads in flight per thread (a dependent chain made this launch latency-bound: 24 us for the 2 048
  // partials of PLCA's EM pass); same summation order as the plain loop
  for (; b + 7 * 32 < nblk; b += 8 * 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (size_t)(b + 32 * u) * r_pad + col);
#pragma unroll
    for (int u = 0; u < 8; ++u) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
  }
  for (; b < nblk; b += 32) {
    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)b * r_pad + col);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  red[g][c4] = s;
  __syncthreads();
  if (g == 0) {
    float4 t = red[0][c4];
#pragma unroll
    for (int i = 1; i < 32; ++i) {
      const float4 v = red[i][c4];
      t.x += v.x, t.y += v.y, t.z += v.z, t.w += v.w;
    }
    *reinterpret_cast<float4*>(out + col) = t;
  }
}

template <int R_PAD, int ROWS>
int launch_apply_rr(const ApplyArgs& a, bool x3, bool pack_only, hipStream_t s) {
  const int grid = a.rows_pad / ROWS;
  const size_t lds = (size_t)ROWS * (R_PAD + 1) * sizeof(float) + ROWS * sizeof(float);   // tile + row sums
  constexpr int NT = apply_threads<ROWS, R_PAD>();
#define L(X, P)                                                                                                      \
  {                                                                                                                  \
    auto k = apply_kernel<R_PAD, X, P, ROWS, NT>;                                                                      \
    if (lds > 64 * 1024) {                                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds);                                                                  \
      if (e != hipSuccess) return (int)e;                                                                            \
    }                                                                                                                \
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, s, a);                                                          \
  }
  if (x3 && pack_only) L(true, true)
  else if (x3) L(true, false)
  else if (pack_only) L(false, true)
  else L(false, false)
#undef L
  int e = (int)hipGetLastError();
  if (e || a.skip_colsum) return e;
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(R_PAD / 32), dim3(256), 0, s, a.colsum_part, grid, R_PAD, a.colsum);
  return (int)hipGetLastError();
}

int apply_stripe_rows(int rows_pad) {
  // short factors (few 64-row stripes) get 16-row stripes so that >= ~256 workgroups exist
  return rows_pad / 64 < 512 ? 16 : 64;
}

template <int R_PAD>
int launch_apply_r(const ApplyArgs& a, bool x3, bool pack_only, hipStream_t s) {
  if (apply_stripe_rows(a.rows_pad) == 16) return launch_apply_rr<R_PAD, 16>(a, x3, pack_only, s);
  return launch_apply_rr<R_PAD, 64>(a, x3, pack_only, s);
}

int launch_colsum_finalize(const float* part, int nblk, int r_pad, float* out, hipStream_t s) {
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(r_pad / 32), dim3(256), 0, s, part, nblk, r_pad, out);
  return (int)hipGetLastError();
}

int launch_apply(int r_pad, const ApplyArgs& a, bool x3, bool pack_only, hipStream_t s) {
  switch (r_pad) {
    case 32: return launch_apply_r<32>(a, x3, pack_only, s);
    case 64: return launch_apply_r<64>(a, x3, pack_only, s);
    case 128: return launch_apply_r<128>(a, x3, pack_only, s);
    case 256: return launch_apply_r<256>(a, x3, pack_only, s);
  }
  return -2;
}

// ------------------------------------------------------------------------------------------------------------
// slab_reduce: out = sum_s slab[s]   (float4 streams)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float4* __restrict__ slab, int nslab, int64_t plane4,
                                                          float4* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < plane4; i += (int64_t)gridDim.x * 256) {
    float4 acc = slab[i];
    for (int s = 1; s < nslab; ++s) {
      const float4 v = slab[s * plane4 + i];
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    out[i] = acc;
  }
}

int launch_slab_reduce(const float* slab, int nslab, int64_t plane, float* out, hipStream_t s) {
  const int64_t plane4 = plane / 4;
  const int grid = (int)std::min<int64_t>((plane4 + 255) / 256, 2048);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(slab), nslab, plane4,
                     reinterpret_cast<float4*>(out));
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// loss finalize: double-precision, fixed-order sum of the per-workgroup partials
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) sum_finalize_kernel(const T* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

int launch_sum_finalize_f32(const float* part, int n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(sum_finalize_kernel<float>, dim3(1), dim3(256), 0, s, part, n, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// loss checkpoint of fit() in ONE launch behind the loss kernel (round 6): block 0 = sum_finalize_kernel's fixed-order sum
// of the partials (the same value, bit for bit) into out[0] and the fp16-range flag (bit 0 of the step's status word)
// into out[1]; every block copies its share of the two factors into their snapshots (what rollback() restores when the
// stop rule of nmf.py:405 turns out to have fired at this checkpoint).  Replaces the finalize launch, five small torch
// launches and two copies per checkpoint.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) checkpoint_kernel(const float* __restrict__ part, int n, const uint32_t* __restrict__ status,
                                                         double* __restrict__ out, const float4* __restrict__ a,
                                                         float4* __restrict__ a_snap, int64_t na4, const float4* __restrict__ b,
                                                         float4* __restrict__ b_snap, int64_t nb4) {
  if (blockIdx.x == 0) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      out[0] = red[0];
      out[1] = (status && (*status & 1u)) ? 1.0 : 0.0;
    }
  }
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na4; i += stride) a_snap[i] = a[i];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb4; i += stride) b_snap[i] = b[i];
}

// ------------------------------------------------------------------------------------------------------------
// "riding loss" (round 6): fit()'s periodic KL loss without its own pass over V (reference: nmf.py:400-401 evaluates
// kl_div(H W^T, V), metrics.py:22 = target @ (log(target + eps) - log(input + eps)) - target.sum() + input.sum()).
//   A = sum x ln(x + eps), C = sum x        target_sums_kernel, once per fit (V does not change; the same pass answers the
//                                           fp16 modes' admission questions: max x, any x fp16 does not hold)
//   B = ln 2 * sum x log2(s + eps), D = sum s - (count) eps
//                                           accumulated by the NEXT W half-step's kernel (its s is this iteration's H W^T + eps,
//                                           from the same operand images the loss pass would read)
//   loss = A - B - C + D                    riding_finish_kernel
// (sum(H W^T) from the masters' column sums instead of D costs one VALU less per element, but differs from the images' sum by
// their rounding: 2e-5 of the loss at 300 x 1000 -- measured, and too close to the stop rule's 1e-4.)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) target_sums_kernel(const float* __restrict__ v, int64_t ld, int rows, int cols,
                                                          double* __restrict__ part) {
  // per block: { sum x ln(x + eps), sum x, max x, any(x != fp16(x)) } -- the last two are what the admission test of the
  // fp16 modes asks about the target (engine.DenseMU.f16_stats), in the same single pass
  __shared__ double red[4][4];
  double a = 0.0, c = 0.0;
  float mx = 0.f;
  int inexact = 0;
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = ld == cols ? v[i] : v[(i / cols) * ld + (i % cols)];
    a += (double)(x * (__builtin_amdgcn_logf(x + kEps) * 0.6931471805599453f));
    c += (double)x;
    mx = fmaxf(mx, x);
    inexact |= ((float)(_Float16)x != x) ? 1 : 0;      // (NaN, and anything beyond 65504: inexact)
  }
  double dm = (double)mx, di = (double)inexact;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64), c += __shfl_xor(c, o, 64);
    dm = fmax(dm, __shfl_xor(dm, o, 64)), di = fmax(di, __shfl_xor(di, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red[0][w] = a, red[1][w] = c, red[2][w] = dm, red[3][w] = di;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[4 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[4 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    part[4 * blockIdx.x + 2] = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
    part[4 * blockIdx.x + 3] = fmax(fmax(red[3][0], red[3][1]), fmax(red[3][2], red[3][3]));
  }
}
__global__ void __launch_bounds__(256) target_sums_finalize_kernel(const double* __restrict__ part, int n, double* __restrict__ out4) {
  __shared__ double red[4][256];
  double a = 0.0, c = 0.0, m = 0.0, x = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += part[4 * i], c += part[4 * i + 1], m = fmax(m, part[4 * i + 2]), x = fmax(x, part[4 * i + 3]);
  red[0][threadIdx.x] = a, red[1][threadIdx.x] = c, red[2][threadIdx.x] = m, red[3][threadIdx.x] = x;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o], red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] = fmax(red[2][threadIdx.x], red[2][threadIdx.x + o]);
      red[3][threadIdx.x] = fmax(red[3][threadIdx.x], red[3][threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out4[0] = red[0][0], out4[1] = red[1][0], out4[2] = red[2][0], out4[3] = red[3][0];
}
int launch_target_sums(const float* v, int64_t ld, int rows, int cols, double* part, int nparts, double* out4, hipStream_t s) {
  hipLaunchKernelGGL(target_sums_kernel, dim3(nparts), dim3(256), 0, s, v, ld, rows, cols, part);
  hipLaunchKernelGGL(target_sums_finalize_kernel, dim3(1), dim3(256), 0, s, part, nparts, out4);
  return (int)hipGetLastError();
}

__global__ void __launch_bounds__(256) riding_finish_kernel(const float* __restrict__ part, int npairs, const double* __restrict__ ac,
                                                            double n_all, const uint32_t* __restrict__ status,
                                                            double* __restrict__ out2) {
  // part: pairs { sum x log2(s), sum s } per wave (s = reconstruction + eps, over EVERY processed element: n_all of them,
  // padding rows / columns hold x = 0 and s = eps)
  __shared__ double red[2][256];
  double b = 0.0, d = 0.0;
  for (int i = threadIdx.x; i < npairs; i += 256) b += (double)part[2 * i], d += (double)part[2 * i + 1];
  red[0][threadIdx.x] = b, red[1][threadIdx.x] = d;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[0][threadIdx.x] += red[0][threadIdx.x + o], red[1][threadIdx.x] += red[1][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = ((ac[0] - red[0][0] * 0.6931471805599453) - ac[1]) + (red[1][0] - n_all * (double)kEps);
    out2[1] = (status && (*status & 1u)) ? 1.0 : 0.0;
  }
}
int launch_riding_finish(const float* part, int npairs, const double* ac, double n_all, const uint32_t* status, double* out2,
                         hipStream_t s) {
  hipLaunchKernelGGL(riding_finish_kernel, dim3(1), dim3(256), 0, s, part, npairs, ac, n_all, status, out2);
  return (int)hipGetLastError();
}

int launch_checkpoint(const float* part, int n, const uint32_t* status, double* out, const float* a, float* a_snap, int64_t na,
                      const float* b, float* b_snap, int64_t nb, hipStream_t s) {
  const int64_t work = (na + nb) / 4;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((work + 256 * 8 - 1) / (256 * 8), 2048));
  hipLaunchKernelGGL(checkpoint_kernel, dim3(grid), dim3(256), 0, s, part, n, status, out, reinterpret_cast<const float4*>(a),
                     reinterpret_cast<float4*>(a_snap), na / 4, reinterpret_cast<const float4*>(b),
                     reinterpret_cast<float4*>(b_snap), nb / 4);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// metrics.beta_div(x, y, beta) on plain arrays (metrics.py:60-96): the public metric, not the fit hot loop.
// ------------------------------------------------------------------------------------------------------------
template <int BETA>
__global__ void __launch_bounds__(256) beta_div_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       int64_t n, float beta, double* __restrict__ part) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float s = (BETA == kEuc) ? x[i] : x[i] + kEps;
    acc += (double)loss_elem<BETA>(s, y[i], beta);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

int launch_beta_div(const float* x, const float* y, int64_t n, float beta, int kind, double* part, double* out,
                    hipStream_t s) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
  switch (kind) {
    case kKL: hipLaunchKernelGGL(beta_div_kernel<kKL>, dim3(grid), dim3(256), 0, s, x, y, n, beta, part); break;
    case kEuc: hipLaunchKernelGGL(beta_div_kernel<kEuc>, dim3(grid), dim3(256), 0, s, x, y, n, beta, part); break;
    case kIS: hipLaunchKernelGGL(beta_div_kernel<kIS>, dim3(grid), dim3(256), 0, s, x, y, n, beta, part); break;
    default: hipLaunchKernelGGL(beta_div_kernel<kGen>, dim3(grid), dim3(256), 0, s, x, y, n, beta, part); break;
  }
  int e = (int)hipGetLastError();
  if (e) return e;
  hipLaunchKernelGGL(sum_finalize_kernel<double>, dim3(1), dim3(256), 0, s, part, grid, out);
  return (int)hipGetLastError();
}

// trainer.BetaMu on a chain of layers (trainer.py:75-112): the two pieces that are not matrix products.
// (1) the back-propagated seeds of trainer.py:75-91 on plain arrays:  gn, gp = f(V, WH, beta)
template <int BETA>
__global__ void __launch_bounds__(256) mu_terms_kernel(const float* __restrict__ s, const float* __restrict__ v, int64_t n,
                                                       float beta, float* __restrict__ gn, float* __restrict__ gp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float a, b;
    mu_elem<BETA>(BETA == kEuc ? s[i] : s[i] + kEps, v[i], beta, a, b);
    gn[i] = a;
    gp[i] = BETA == kKL ? 1.f : b;      // beta == 1 back-propagates ones (trainer.py:84)
  }
}

int launch_mu_terms(const float* s, const float* v, int64_t n, float beta, int kind, float* gn, float* gp, hipStream_t st) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096));
  switch (kind) {
    case kKL: hipLaunchKernelGGL(mu_terms_kernel<kKL>, dim3(grid), dim3(256), 0, st, s, v, n, beta, gn, gp); break;
    case kEuc: hipLaunchKernelGGL(mu_terms_kernel<kEuc>, dim3(grid), dim3(256), 0, st, s, v, n, beta, gn, gp); break;
    case kIS: hipLaunchKernelGGL(mu_terms_kernel<kIS>, dim3(grid), dim3(256), 0, st, s, v, n, beta, gn, gp); break;
    default: hipLaunchKernelGGL(mu_terms_kernel<kGen>, dim3(grid), dim3(256), 0, st, s, v, n, beta, gn, gp); break;
  }
  return (int)hipGetLastError();
}

// (2) trainer.py:93-112 on a plain row-major parameter [rows][cols]; one workgroup per row (row sum for the
// orthogonality penalty in a fixed order)
__global__ void __launch_bounds__(256) trainer_update_kernel(float* __restrict__ f, int cols, const float* __restrict__ neg,
                                                             const float* __restrict__ pos, float l1, float l2, float ortho,
                                                             float gamma, float* __restrict__ grad) {
  __shared__ float red[256];
  const size_t base = (size_t)blockIdx.x * cols;
  float rs = 0.f;
  if (ortho > 0.f) {
    for (int c = threadIdx.x; c < cols; c += 256) rs += f[base + c];
    red[threadIdx.x] = rs;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    rs = red[0];
  }
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float x = f[base + c];
    const float ng = fmaxf(neg[base + c], 0.f);
    float ps = fmaxf(pos[base + c], 0.f);
    if (grad) grad[base + c] = ps - ng;             // trainer.py:98
    if (l1 > 0.f) ps += l1;
    if (l2 > 0.f) ps += l2 * x;
    if (ortho > 0.f) ps += ortho * (rs - x);        // trainer.py:105-106
    float mult = (ng + kEps) / (ps + kEps);         // trainer.py:108-110
    if (gamma != 1.f) mult = powf(mult, gamma);
    f[base + c] = x * mult;
  }
}

int launch_trainer_update(float* f, int rows, int cols, const float* neg, const float* pos, float l1, float l2, float ortho,
                          float gamma, float* grad, hipStream_t st) {
  hipLaunchKernelGGL(trainer_update_kernel, dim3(rows), dim3(256), 0, st, f, cols, neg, pos, l1, l2, ortho, gamma, grad);
  return (int)hipGetLastError();
}

// metrics.sparseness (metrics.py:99-115) needs ||x||_1 and ||x||_2: out[0] = sum |x|, out[1] = sum x^2
__global__ void __launch_bounds__(256) norms_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ part) {
  __shared__ double red[2][4];
  double a1 = 0.0, a2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = (double)x[i];
    a1 += fabs(v), a2 += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a1 += __shfl_xor(a1, o, 64), a2 += __shfl_xor(a2, o, 64);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = a1, red[1][threadIdx.x >> 6] = a2;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[gridDim.x + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

int launch_norms(const float* x, int64_t n, double* part, double* out, hipStream_t s) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 512));   // part holds 2 * 512 doubles
  hipLaunchKernelGGL(norms_kernel, dim3(grid), dim3(256), 0, s, x, n, part);
  hipLaunchKernelGGL(sum_finalize_kernel<double>, dim3(1), dim3(256), 0, s, part, grid, out);
  hipLaunchKernelGGL(sum_finalize_kernel<double>, dim3(1), dim3(256), 0, s, part + grid, grid, out + 1);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// reconstruct: out[m][k] = sum_r A[m][r] B[k][r], fp32 in / fp32 out (NMF.reconstruct, nmf.py:691-693).
// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  128x128 output tile per workgroup (4 waves, 64x64 = 2x2 MFMA tiles
// each); the rank axis is staged through LDS 32 columns at a time with coalesced float4 loads (row stride 33 floats:
// the per-lane operand reads A[row j][r] are then conflict free).  Bound by the N x C fp32 store and the fp32 MFMA
// rate (1/16 of bf16); it is the materialising forward(), not part of the fit loop.
// ------------------------------------------------------------------------------------------------------------
constexpr int kRecBK = 32, kRecLD = kRecBK + 1;

__global__ void __launch_bounds__(256) reconstruct_kernel(const float* __restrict__ A, int M, const float* __restrict__ B,
                                                          int K, int R, float* __restrict__ out, int64_t ld) {
  __shared__ float sa[128 * kRecLD], sb[128 * kRecLD];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, hl = lane >> 5;
  const int wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const bool vec = (R & 3) == 0;   // float4 loads need 16-byte aligned rows
  for (int r0 = 0; r0 < R; r0 += kRecBK) {
    // stage A[m0 .. m0+127][r0 .. r0+31] and B likewise (zero fill outside the matrix)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = p * 256 + tid, row = idx >> 3, c4 = (idx & 7) * 4;
      float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
      const int r = r0 + c4;
      if (m0 + row < M && r < R) {
        const float* ap = A + (size_t)(m0 + row) * R + r;
        if (vec) {
          const float4 v = *reinterpret_cast<const float4*>(ap);
          av[0] = v.x, av[1] = v.y, av[2] = v.z, av[3] = v.w;
        } else {
          for (int i = 0; i < 4 && r + i < R; ++i) av[i] = ap[i];
        }
      }
      if (k0 + row < K && r < R) {
        const float* bp = B + (size_t)(k0 + row) * R + r;
        if (vec) {
          const float4 v = *reinterpret_cast<const float4*>(bp);
          bv[0] = v.x, bv[1] = v.y, bv[2] = v.z, bv[3] = v.w;
        } else {
          for (int i = 0; i < 4 && r + i < R; ++i) bv[i] = bp[i];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) sa[row * kRecLD + c4 + i] = av[i], sb[row * kRecLD + c4 + i] = bv[i];
    }
    __syncthreads();
    const float* pa = sa + (wm * 64 + j) * kRecLD + hl;
    const float* pb = sb + (wn * 64 + j) * kRecLD + hl;
#pragma unroll
    for (int s2 = 0; s2 < kRecBK; s2 += 2) {
      const float a0 = pa[s2], a1 = pa[32 * kRecLD + s2];
      const float b0 = pb[s2], b1 = pb[32 * kRecLD + s2];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int k = k0 + wn * 64 + b * 32 + j;
      if (k < K) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
          if (row < M) __builtin_nontemporal_store(acc[a][b][e], out + (size_t)row * ld + k);
        }
      }
    }
}

int launch_reconstruct(const float* A, int M, const float* B, int K, int R, float* out, int64_t ld, hipStream_t s) {
  dim3 grid((K + 127) / 128, (M + 127) / 128);
  hipLaunchKernelGGL(reconstruct_kernel, grid, dim3(256), 0, s, A, M, B, K, R, out, ld);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// probes: validate the two hardware assumptions everything rests on.
// ------------------------------------------------------------------------------------------------------------
// (1) v_mfma_f32_32x32x16_bf16 operand / result lane maps, exactly as the fused kernel uses them.
__global__ void __launch_bounds__(64) probe_mfma_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                        float* __restrict__ d) {
  const int lane = threadIdx.x, j = lane & 31, hl = lane >> 5;
  u32x4 av, bv;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // A[i = j][k = 8*hl + e], row-major 32x16 ; B[k][j], row-major 16x32
    av[i] = (uint32_t)a[j * 16 + 8 * hl + 2 * i] | ((uint32_t)a[j * 16 + 8 * hl + 2 * i + 1] << 16);
    bv[i] = (uint32_t)b[(8 * hl + 2 * i) * 32 + j] | ((uint32_t)b[(8 * hl + 2 * i + 1) * 32 + j] << 16);
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = mfma_bf16(av, bv, acc);
#pragma unroll
  for (int e = 0; e < 16; ++e) d[((e & 3) + 8 * (e >> 2) + 4 * hl) * 32 + j] = acc[e];
}

int launch_probe_mfma(const uint16_t* a, const uint16_t* b, float* d, hipStream_t s) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, s, a, b, d);
  return (int)hipGetLastError();
}

// (2) global_load_lds: LDS destination = wave-uniform base + lane * 16, source per lane.
__global__ void __launch_bounds__(256) probe_lds_dma_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                            int n_dwords) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int passes = n_dwords / 1024;  // 4 KiB per pass
  for (int p = 0; p < passes; ++p) {
    const char* g = reinterpret_cast<const char*>(src) + (size_t)p * 4096 + tid * 16;
    char* l = smem + p * 4096 + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  }
  __syncthreads();
  for (int i = tid; i < n_dwords; i += 256) dst[i] = reinterpret_cast<const uint32_t*>(smem)[i];
}

int launch_probe_lds_dma(const uint32_t* src, uint32_t* dst, int n_dwords, hipStream_t s) {
  hipLaunchKernelGGL(probe_lds_dma_kernel, dim3(1), dim3(256), (size_t)n_dwords * 4, s, src, dst, n_dwords);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// (3) the zero-overhead ceiling of the fused MU step: an MFMA loop beside an
// INDEPENDENT HBM stream at the MU step's flop-per-byte ratio.  Every wave issues 32 MFMAs (32x32x16, bf16 or fp16) per
// "tile" on operand fragments that never change -- taken from a REAL factor image, so that the matrix pipe sees the value
// distribution of the MU step's operands (its power draw, and with it the clock the socket's limit leaves, follow the data)
// -- and streams NLOAD KiB per tile from HBM by LDS-DMA (non-temporal, never read back, at most three tiles in flight).
// No LDS reads, no VALU work, no barriers, nothing depends on the stream.  NLOAD = 4 is the rank-128 MU step: 4 KiB of
// 16-bit X per wave and 32 MFMAs = 256 flop per X byte; grid = CUs x 8 waves x 64 tiles = exactly the MFMA count and the
// X bytes of one configs[1] half-step.  What this loop sustains is what a kernel with NO overhead at all could reach on
// this part under its power limit; bench.py prices the shipped kernel against it next to the nominal 2.5 PFLOP/s.
template <int NLOAD, bool F16>
__global__ void __launch_bounds__(512) ubench_mfma_hbm_kernel(const char* __restrict__ operands, size_t operand_bytes,
                                                              const char* __restrict__ src_all, int tiles, float* __restrict__ out,
                                                              int wrap_tiles, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nwaves = (int)blockDim.x >> 6;
  // 2 x 8 fragments of 1 KiB per wave: A from the first half of the image, B from the second (16-byte chunks = 8 consecutive
  // ranks of one factor row, the unit the fused kernels feed to the MFMA as well)
  const size_t half = (operand_bytes / 2) & ~(size_t)16383;
  const size_t woff = (((size_t)blockIdx.x * nwaves + wave) * 8192) % (half - 8192);
  u32x4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const u32x4*>(operands + woff + (size_t)i * 1024 + lane * 16);
    b[i] = *reinterpret_cast<const u32x4*>(operands + half + woff + (size_t)i * 1024 + lane * 16);
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 8192u;
  // (round 6) a launch may run more tiles than the stream holds: it then wraps after wrap_tiles tiles per wave (the ceiling
  // wants a LONG loop -- 64 tiles are 88 us of which ramp, operand fetch and drain are a fifth); wave 0 stamps the loop
  const int wt = wrap_tiles > 0 ? wrap_tiles : tiles;
  const char* src = src_all + ((size_t)blockIdx.x * nwaves + wave) * ((size_t)wt * (NLOAD ? NLOAD : 1) * 1024) + lane * 16;
  unsigned long long c0 = 0, r0 = 0;
  if (stamps) {      // every wave stamps its own loop: the two waves of a SIMD do not share the pipe evenly (the older one wins)
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  int tw = 0;
  for (int t = 0; t < tiles; ++t) {
    if constexpr (NLOAD > 0) {
#pragma unroll
      for (int p = 0; p < NLOAD; ++p) {
        const unsigned la = lds_base + (unsigned)((p & 7) * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
                     :
                     : "v"(src + (size_t)(tw * NLOAD + p) * 1024), "s"(la)
                     : "memory", "m0");
      }
      tw = tw + 1 == wt ? 0 : tw + 1;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLOAD > 63 ? 63 : 3 * NLOAD) : "memory");
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[0] = mfma_op<F16 ? kOpF16 : kOpBf16>(a[i], b[i], acc[0]);
      acc[1] = mfma_op<F16 ? kOpF16 : kOpBf16>(a[i], b[(i + 1) & 7], acc[1]);
      acc[2] = mfma_op<F16 ? kOpF16 : kOpBf16>(a[(i + 2) & 7], b[i], acc[2]);
      acc[3] = mfma_op<F16 ? kOpF16 : kOpBf16>(a[(i + 3) & 7], b[(i + 5) & 7], acc[3]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float tt = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) tt += acc[i][e];   // (reads every accumulator: the loop's MFMAs have retired)
  if (stamps) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      stamps[4 * ((size_t)blockIdx.x * nwaves + wave)] = c1 - c0;       // shader cycles of this wave's loop
      stamps[4 * ((size_t)blockIdx.x * nwaves + wave) + 1] = r1 - r0;   // 100 MHz ticks of it
      stamps[4 * ((size_t)blockIdx.x * nwaves + wave) + 2] = r0;        // its start and end on the constant clock: the
      stamps[4 * ((size_t)blockIdx.x * nwaves + wave) + 3] = r1;        // workgroup's span is max(end) - min(start)
    }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = tt;
}

int launch_ubench_mfma_hbm(const void* operands, size_t operand_bytes, int f16, const void* stream_src, int kib_per_tile,
                           int waves, int tiles, int grid, float* out, hipStream_t s, int wrap_tiles, unsigned long long* stamps) {
  auto go = [&](auto kern) {
    static bool done[64] = {};
    (void)done;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), (size_t)waves * 8192, s, (const char*)operands, operand_bytes,
                       (const char*)stream_src, tiles, out, wrap_tiles, stamps);
    return (int)hipGetLastError();
  };
#define UB(N) \
  if (kib_per_tile == N) return f16 ? go(ubench_mfma_hbm_kernel<N, true>) : go(ubench_mfma_hbm_kernel<N, false>);
  UB(0) UB(1) UB(2) UB(4) UB(8)
#undef UB
  return -2;
}

}  // namespace nmfmu
