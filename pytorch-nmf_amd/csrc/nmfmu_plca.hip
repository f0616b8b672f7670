// PLCA's EM update (reference: plca.py:248-290) around the fused kernel.
//
// The O(N C R) work of an EM iteration -- G = Vn / (H diag(Z) W^T + eps), G^T H and G W -- runs on the fused MU kernel
// with a split panel (first GEMM reads the image of the Z-scaled factor, second GEMM the unscaled one).  What is left
// is O((N + C) R) per factor, the three small kernels here:
//   plca_em        f *= relu(num * z_old)          + column sums of the result + Z.grad partials  sum_rows f_old * num
//   plca_normalize f /= divider ; Dirichlet prior: f += alpha - 1, clamp below at eps ; column sums of the result
//   plca_scale     f /= colsum                     (renormalisation after a prior)
// Column sums are two-stage and order-fixed (deterministic), like everywhere else in this library.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nmfmu.h"
#include "nmfmu_aux.h"
#include "nmfmu_fused.h"

namespace nmfmu {

constexpr int kPlcaRows = 32;   // factor rows per workgroup (128 measured slower: too few workgroups for the short factor)

// threads: column r = tid % r_pad, row group g = tid / r_pad (256 / r_pad groups); rows g, g + groups, ... of the block
template <int MODE>   // 0: em, 1: normalize
__global__ void __launch_bounds__(256) plca_kernel(float* __restrict__ f, int rows, int rank, int r_pad,
                                                   const float* __restrict__ num, int nslab, size_t plane,
                                                   const float* __restrict__ vec, float alpha, int update,
                                                   float* __restrict__ cs_part, float* __restrict__ zg_part) {
  __shared__ float red[2][256];
  const int tid = threadIdx.x, r = tid % r_pad, g = tid / r_pad, groups = 256 / r_pad;
  const int row0 = blockIdx.x * kPlcaRows;
  float cs = 0.f, zg = 0.f;
  if (r < rank) {
    const float v = vec[r];   // MODE 0: z_old[r];  MODE 1: divider[r]
#pragma unroll 4
    for (int rl = g; rl < kPlcaRows; rl += groups) {   // (unrolled: several rows' loads in flight; f and num do not alias)
      const int row = row0 + rl;
      if (row >= rows) break;
      const size_t i = (size_t)row * rank + r;
      float x = f[i];
      if constexpr (MODE == 0) {
        float n = 0.f;
        for (int s = 0; s < nslab; ++s) n += num[s * plane + (size_t)row * r_pad + r];   // unscaled numerator
        zg += x * n;                                   // Z.grad[r] = sum f_old * num   (plca.py:250)
        x *= fmaxf(n * v, 0.f);                        // f *= relu(f.grad), f.grad = num * z_old  (plca.py:263, 277)
      } else {
        x /= v;                                        // plca.py:270, 284
        if (alpha != 1.f) {
          x += alpha - 1.f;                            // plca.py:272-274, 286-288
          x = x > kEps ? x : kEps;                     // F.threshold(x, eps, eps)
        }
      }
      if (update) f[i] = x;
      cs += x;
    }
  }
  red[0][tid] = cs;
  red[1][tid] = zg;
  __syncthreads();
  if (g == 0) {
    for (int k = 1; k < groups; ++k) cs += red[0][k * r_pad + r], zg += red[1][k * r_pad + r];
    cs_part[(size_t)blockIdx.x * r_pad + r] = cs;
    if (zg_part) zg_part[(size_t)blockIdx.x * r_pad + r] = zg;
  }
}

__global__ void __launch_bounds__(256) plca_scale_kernel(float* __restrict__ f, int64_t n, int rank,
                                                         const float* __restrict__ colsum) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) f[i] /= colsum[i % rank];
}

// The latent update of plca.py:253-260 in one launch (round 4; it was a dozen R-element torch ops per EM iteration):
// prior[r] = z[r] * relu(zgrad[r]) (what the factors' normalisation divides by); z <- prior (+ Dirichlet prior: + alpha - 1,
// clamped below at eps); z /= sum(z).  One workgroup, rank <= 256, fixed-order sum.
__global__ void __launch_bounds__(256) plca_z_kernel(float* __restrict__ z, const float* __restrict__ zgrad, int rank,
                                                     float alpha, float* __restrict__ prior) {
  __shared__ float red[256];
  const int r = threadIdx.x;
  float z1 = 0.f;
  if (r < rank) {
    z1 = z[r] * fmaxf(zgrad[r], 0.f);
    prior[r] = z1;
    if (alpha != 1.f) {
      z1 += alpha - 1.f;
      z1 = z1 > kEps ? z1 : kEps;
    }
  }
  red[r] = z1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (r < o) red[r] += red[r + o];
    __syncthreads();
  }
  if (r < rank) z[r] = z1 / red[0];
}

// ------------------------------------------------------------------------------------------------------------
// Shift-invariant PLCA (plca.py:376-606): W is (C, R, *T) and H is (B, R, *Lh) -- the rank axis sits in the middle, so
// a factor is addressed as [outer][R][inner] (W: outer = C, inner = prod T; H: outer = B, inner = prod Lh) and the
// unscaled numerator as num[o * num_pitch + r * inner + i] (W: a row of the [c_pad][rp_pad] GEMM output; H: the folded
// [B][R][Lh] buffer).  Workgroup (r, chunk) walks its slice of the (outer x inner) index space of rank r; the chunk
// partials are combined in order (deterministic).
// ------------------------------------------------------------------------------------------------------------
constexpr int kPlca3Chunks = 64;

template <int MODE>   // 0: em, 1: normalize, 2: scale
__global__ void __launch_bounds__(256) plca3_kernel(float* __restrict__ f, int outer, int R, int inner,
                                                    const float* __restrict__ num, int64_t num_pitch,
                                                    const float* __restrict__ vec, float alpha, int update,
                                                    float* __restrict__ part) {
  __shared__ float red[2][256];
  const int r = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const int64_t n = (int64_t)outer * inner;
  const int64_t per = (n + kPlca3Chunks - 1) / kPlca3Chunks, e0 = ch * per, e1 = min(n, e0 + per);
  const float v = vec[r];
  float cs = 0.f, zg = 0.f;
  for (int64_t e = e0 + tid; e < e1; e += 256) {
    const int64_t o = e / inner, i = e - o * inner;
    const size_t idx = ((size_t)o * R + r) * inner + i;
    float x = f[idx];
    if constexpr (MODE == 0) {
      const float nv = num[(size_t)o * num_pitch + (size_t)r * inner + i];
      zg += x * nv;
      x *= fmaxf(nv * v, 0.f);
    } else if constexpr (MODE == 1) {
      x /= v;
      if (alpha != 1.f) {
        x += alpha - 1.f;
        x = x > kEps ? x : kEps;
      }
    } else {
      x /= v;
    }
    if (update) f[idx] = x;
    cs += x;
  }
  red[0][tid] = cs;
  red[1][tid] = zg;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[0][tid] += red[0][tid + o], red[1][tid] += red[1][tid + o];
    __syncthreads();
  }
  if (tid == 0 && part) {
    part[((size_t)r * kPlca3Chunks + ch) * 2] = red[0][0];
    part[((size_t)r * kPlca3Chunks + ch) * 2 + 1] = red[1][0];
  }
}

__global__ void __launch_bounds__(64) plca3_final_kernel(const float* __restrict__ part, int R, float* __restrict__ cs_out,
                                                         float* __restrict__ zg_out) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= R) return;
  float cs = 0.f, zg = 0.f;
  for (int ch = 0; ch < kPlca3Chunks; ++ch) cs += part[((size_t)r * kPlca3Chunks + ch) * 2], zg += part[((size_t)r * kPlca3Chunks + ch) * 2 + 1];
  cs_out[r] = cs;
  if (zg_out) zg_out[r] = zg;
}

}  // namespace nmfmu

using namespace nmfmu;

namespace {
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int nblk_of(int rows) { return (rows + kPlcaRows - 1) / kPlcaRows; }
}  // namespace

extern "C" {

size_t nmfmu_plca_part_bytes(int rows, int r_pad) {
  return rows > 0 && r_pad > 0 ? (size_t)2 * nblk_of(rows) * r_pad * 4 : 0;
}

int nmfmu_plca_em(float* f, int rows, int rank, int r_pad, const float* num, int nslab, int rows_pad, const float* z_old,
                  int update, float* part, float* colsum_out, float* zgrad_out, void* stream) {
  if (!f || !num || !z_old || !part || !colsum_out || rows <= 0 || rank <= 0 || nslab < 1) return NMFMU_ERR_ARG;
  if (r_pad != nmfmu_pad_rank(rank) || rows_pad < rows) return NMFMU_ERR_ARG;
  const int nb = nblk_of(rows);
  float* zg_part = zgrad_out ? part + (size_t)nb * r_pad : nullptr;
  hipLaunchKernelGGL(plca_kernel<0>, dim3(nb), dim3(256), 0, S(stream), f, rows, rank, r_pad, num, nslab,
                     (size_t)rows_pad * r_pad, z_old, 1.f, update, part, zg_part);
  int e = launch_colsum_finalize(part, nb, r_pad, colsum_out, S(stream));   // two-stage, order-fixed (nmfmu_aux.hip)
  if (!e && zgrad_out) e = launch_colsum_finalize(zg_part, nb, r_pad, zgrad_out, S(stream));
  return e;
}

int nmfmu_plca_normalize(float* f, int rows, int rank, int r_pad, const float* divider, float alpha, float* part,
                         float* colsum_out, void* stream) {
  if (!f || !divider || !part || !colsum_out || rows <= 0 || rank <= 0 || r_pad != nmfmu_pad_rank(rank)) return NMFMU_ERR_ARG;
  const int nb = nblk_of(rows);
  hipLaunchKernelGGL(plca_kernel<1>, dim3(nb), dim3(256), 0, S(stream), f, rows, rank, r_pad, nullptr, 0, (size_t)0, divider,
                     alpha, 1, part, nullptr);
  return launch_colsum_finalize(part, nb, r_pad, colsum_out, S(stream));
}

int nmfmu_plca_scale(float* f, int rows, int rank, const float* colsum, void* stream) {
  if (!f || !colsum || rows <= 0 || rank <= 0) return NMFMU_ERR_ARG;
  const int64_t n = (int64_t)rows * rank;
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(plca_scale_kernel, dim3(grid), dim3(256), 0, S(stream), f, n, rank, colsum);
  return (int)hipGetLastError();
}

int nmfmu_plca_z(float* z, const float* zgrad, int rank, float alpha, float* prior, void* stream) {
  if (!z || !zgrad || !prior || rank <= 0 || rank > 256) return NMFMU_ERR_ARG;
  hipLaunchKernelGGL(plca_z_kernel, dim3(1), dim3(256), 0, S(stream), z, zgrad, rank, alpha, prior);
  return (int)hipGetLastError();
}

size_t nmfmu_plca3_part_bytes(int rank) { return rank > 0 ? (size_t)rank * kPlca3Chunks * 2 * 4 : 0; }

/* mode 0: em (num, z_old), 1: normalize (divider, alpha), 2: scale (colsum) -- see nmfmu_plca_* for the semantics */
int nmfmu_plca3(int mode, float* f, int outer, int rank, int inner, const float* num, int64_t num_pitch, const float* vec,
                float alpha, int update, float* part, float* colsum_out, float* zgrad_out, void* stream) {
  if (!f || !vec || outer <= 0 || rank <= 0 || inner <= 0 || mode < 0 || mode > 2) return NMFMU_ERR_ARG;
  if (mode == 0 && (!num || num_pitch < (int64_t)rank * inner)) return NMFMU_ERR_ARG;
  if (mode != 2 && (!part || !colsum_out)) return NMFMU_ERR_ARG;
  const dim3 grid(rank, kPlca3Chunks);
  if (mode == 0)
    hipLaunchKernelGGL(plca3_kernel<0>, grid, dim3(256), 0, S(stream), f, outer, rank, inner, num, num_pitch, vec, 1.f, update, part);
  else if (mode == 1)
    hipLaunchKernelGGL(plca3_kernel<1>, grid, dim3(256), 0, S(stream), f, outer, rank, inner, nullptr, (int64_t)0, vec, alpha, 1, part);
  else
    hipLaunchKernelGGL(plca3_kernel<2>, grid, dim3(256), 0, S(stream), f, outer, rank, inner, nullptr, (int64_t)0, vec, 1.f, 1,
                       (float*)nullptr);
  if (mode != 2)
    hipLaunchKernelGGL(plca3_final_kernel, dim3((rank + 63) / 64), dim3(64), 0, S(stream), part, rank, colsum_out,
                       mode == 0 ? zgrad_out : nullptr);
  return (int)hipGetLastError();
}

}  // extern "C"
