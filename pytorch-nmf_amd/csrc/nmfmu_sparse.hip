// Sparse-COO targets (reference: nmf.py:351-398, 602-638), beta in {1, 2}.
//
// The target is stored as CSR over the OWNER axis of a half-step (rows of V for the H half-step, rows of V^T for the
// W half-step).  One wave per owner row; the lanes span the rank, so every panel row is one coalesced read:
//   s   = <owner[row], panel[col]>                       wave reduction (fixed order -> deterministic)
//   g   = v / (s + eps)   (beta == 1)    |    v          (beta == 2)
//   num[row][:] += g * panel[col][:]
// HBM / L2-gather bound by design (2 R flops per stored entry and panel element); no MFMA, no reshaping into GEMMs.
// The denominators are the dense closed forms: column sums of the panel (beta == 1, nmf.py:122-131) or
// owner @ (panel^T panel) (beta == 2: the gradient of the reference's pos = 1/2 <H W^T W, H>, nmf.py:616-617).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nmfmu.h"
#include "nmfmu_fused.h"

namespace nmfmu {

__device__ __forceinline__ float wave_sum(float v) {   // xor butterfly: every lane gets the same, order-fixed sum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// RL = rank slots per lane (r_pad / 64 rounded up; r_pad 32 uses half a wave's lanes with zeros beyond the rank)
// KIND: kKL g = v / (s + eps); kEuc g = v; kGen g = v (s + eps)^(beta - 2)
template <int RL, int KIND>
__global__ void __launch_bounds__(256) sp_partial_kernel(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colidx,
                                                         const float* __restrict__ vals, int rows,
                                                         const float* __restrict__ owner,
                                                         const float* __restrict__ panel, int rank,
                                                         float* __restrict__ num, int r_pad, float beta) {
  constexpr bool KL = KIND != kEuc;   // needs the dot product
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float a[RL], acc[RL];
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    const int r = lane + 64 * q;
    a[q] = (KL && r < rank) ? owner[(size_t)row * rank + r] : 0.f;
    acc[q] = 0.f;
  }
  const int p0 = rowptr[row], p1 = rowptr[row + 1];
  // U stored entries per trip: their index / value / panel-row loads are independent, so U gathers are in flight per
  // wave instead of one (the loop is otherwise a chain of two dependent loads per entry)
  constexpr int U = 4;
  for (int p = p0; p < p1; p += U) {
    int col[U];
    float v[U], b[U][RL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = p + u < p1;
      col[u] = ok ? colidx[p + u] : 0;
      v[u] = ok ? vals[p + u] : 0.f;      // v = 0 contributes nothing (s + eps > 0)
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const int r = lane + 64 * q;
        b[u][q] = r < rank ? panel[(size_t)col[u] * rank + r] : 0.f;
      }
    float g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) g[u] = v[u];
    if constexpr (KL) {
      float part[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        part[u] = 0.f;
#pragma unroll
        for (int q = 0; q < RL; ++q) part[u] += a[q] * b[u][q];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)     // U butterflies interleaved; each is the fixed-order wave_sum
#pragma unroll
        for (int u = 0; u < U; ++u) part[u] += __shfl_xor(part[u], o, 64);
#pragma unroll
      for (int u = 0; u < U; ++u) {       // nmf.py:65 / 72 restricted to the stored entries
        const float se = part[u] + kEps;
        g[u] = KIND == kKL ? v[u] / se : v[u] * exp2f((beta - 2.f) * log2f(se));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)           // entries accumulate in storage order: deterministic
#pragma unroll
      for (int q = 0; q < RL; ++q) acc[q] += g[u] * b[u][q];
  }
#pragma unroll
  for (int q = 0; q < RL; ++q) {
    const int r = lane + 64 * q;
    if (r < r_pad) num[(size_t)row * r_pad + r] = acc[q];
  }
}

// neg term of the tracked loss: sum over stored entries of v log(s + eps) (beta == 1) or v s (beta == 2); one
// double partial per workgroup (4 rows), summed on the host side of the ABI in a fixed order by sp_reduce_kernel.
template <int RL, int KIND>
__global__ void __launch_bounds__(256) sp_loss_kernel(const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ colidx,
                                                      const float* __restrict__ vals, int rows,
                                                      const float* __restrict__ owner,
                                                      const float* __restrict__ panel, int rank,
                                                      double* __restrict__ part, float beta) {
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  double tot = 0.0;
  if (row < rows) {
    float a[RL];
#pragma unroll
    for (int q = 0; q < RL; ++q) {
      const int r = lane + 64 * q;
      a[q] = r < rank ? owner[(size_t)row * rank + r] : 0.f;
    }
    for (int p = rowptr[row]; p < rowptr[row + 1]; ++p) {
      const int col = colidx[p];
      float partial = 0.f;
#pragma unroll
      for (int q = 0; q < RL; ++q) {
        const int r = lane + 64 * q;
        partial += r < rank ? a[q] * panel[(size_t)col * rank + r] : 0.f;
      }
      const float s = wave_sum(partial);
      tot += KIND == kKL ? (double)(vals[p] * logf(s + kEps))
           : KIND == kEuc ? (double)(vals[p] * s)
                          : (double)(vals[p] * exp2f((beta - 1.f) * log2f(s + kEps)) / (beta - 1.f));   // nmf.py:636
    }
  }
  if (lane == 0) red[w] = tot;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) sp_reduce_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

// gram[a][b] = sum_rows f[row][a] f[row][b]   (rank <= 256).  Two deterministic stages: workgroup (a, chunk) sums
// its slice of rows for every column b (coalesced along b, four independent accumulators), then the chunk partials
// are added in order.  part: [kGramChunks][rank][rank] floats of scratch.
constexpr int kGramChunks = 64;

__global__ void __launch_bounds__(256) gram_partial_kernel(const float* __restrict__ f, int rows, int rank,
                                                           float* __restrict__ part) {
  const int a = blockIdx.x, ch = blockIdx.y, b = threadIdx.x;
  if (b >= rank) return;
  const int per = (rows + kGramChunks - 1) / kGramChunks;
  const int i0 = ch * per, i1 = min(rows, i0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = i0;
  for (; i + 3 < i1; i += 4) {
    s0 += f[(size_t)i * rank + a] * f[(size_t)i * rank + b];
    s1 += f[(size_t)(i + 1) * rank + a] * f[(size_t)(i + 1) * rank + b];
    s2 += f[(size_t)(i + 2) * rank + a] * f[(size_t)(i + 2) * rank + b];
    s3 += f[(size_t)(i + 3) * rank + a] * f[(size_t)(i + 3) * rank + b];
  }
  for (; i < i1; ++i) s0 += f[(size_t)i * rank + a] * f[(size_t)i * rank + b];
  part[((size_t)ch * rank + a) * rank + b] = (s0 + s1) + (s2 + s3);
}

__global__ void __launch_bounds__(256) gram_final_kernel(const float* __restrict__ part, int rank, float* __restrict__ gram) {
  const int a = blockIdx.x, b = threadIdx.x;
  if (b >= rank) return;
  float s = 0.f;
  for (int ch = 0; ch < kGramChunks; ++ch) s += part[((size_t)ch * rank + a) * rank + b];
  gram[a * rank + b] = s;
}

// den[row][r] = sum_q owner[row][q] gram[q][r]; one wave per row, gram rows streamed from L2
__global__ void __launch_bounds__(256) rowmat_kernel(const float* __restrict__ owner, int rows, int rank,
                                                     const float* __restrict__ gram, float* __restrict__ den, int r_pad) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  for (int r = lane; r < r_pad; r += 64) {
    float s = 0.f;
    if (r < rank)
      for (int q = 0; q < rank; ++q) s += owner[(size_t)row * rank + q] * gram[q * rank + r];
    den[(size_t)row * r_pad + r] = s;
  }
}

}  // namespace nmfmu

using namespace nmfmu;

namespace {
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
}

extern "C" {

int nmfmu_sp_partial(const int32_t* rowptr, const int32_t* colidx, const float* vals, int owner_rows, const float* owner,
                     const float* panel, int rank, float beta, float* num, int r_pad, void* stream) {
  if (!rowptr || !colidx || !vals || !owner || !panel || !num || owner_rows <= 0 || rank <= 0) return NMFMU_ERR_ARG;
  if (r_pad != nmfmu_pad_rank(rank)) return NMFMU_ERR_ARG;
  const int kind = nmfmu_beta_kind(beta);
  if (kind == NMFMU_BETA_IS) return NMFMU_ERR_UNSUPPORTED;   // beta == 0 is rejected for sparse targets (nmf.py:332-336)
  const dim3 grid((owner_rows + 3) / 4), block(256);
#define L2(RLV, K)                                                                                                   \
  hipLaunchKernelGGL((sp_partial_kernel<RLV, K>), grid, block, 0, S(stream), rowptr, colidx, vals, owner_rows, owner,  \
                     panel, rank, num, r_pad, beta);
#define L(RLV)                                                                         \
  if (kind == NMFMU_BETA_KL) { L2(RLV, kKL) } else if (kind == NMFMU_BETA_EUC) { L2(RLV, kEuc) } else { L2(RLV, kGen) }
  if (r_pad <= 64) { L(1) } else if (r_pad == 128) { L(2) } else { L(4) }
#undef L
#undef L2
  return (int)hipGetLastError();
}

int nmfmu_sp_loss_neg(const int32_t* rowptr, const int32_t* colidx, const float* vals, int owner_rows, const float* owner,
                      const float* panel, int rank, float beta, double* part, double* out, void* stream) {
  if (!rowptr || !colidx || !vals || !owner || !panel || !part || !out || owner_rows <= 0 || rank <= 0 || rank > 256)
    return NMFMU_ERR_ARG;
  const int kind = nmfmu_beta_kind(beta);
  if (kind == NMFMU_BETA_IS) return NMFMU_ERR_UNSUPPORTED;
  const int nblk = (owner_rows + 3) / 4;
  const int r_pad = nmfmu_pad_rank(rank);
#define L2(RLV, K)                                                                                                   \
  hipLaunchKernelGGL((sp_loss_kernel<RLV, K>), dim3(nblk), dim3(256), 0, S(stream), rowptr, colidx, vals, owner_rows,  \
                     owner, panel, rank, part, beta);
#define L(RLV)                                                                         \
  if (kind == NMFMU_BETA_KL) { L2(RLV, kKL) } else if (kind == NMFMU_BETA_EUC) { L2(RLV, kEuc) } else { L2(RLV, kGen) }
  if (r_pad <= 64) { L(1) } else if (r_pad == 128) { L(2) } else { L(4) }
#undef L
#undef L2
  hipLaunchKernelGGL(sp_reduce_kernel, dim3(1), dim3(256), 0, S(stream), part, nblk, out);
  return (int)hipGetLastError();
}

size_t nmfmu_gram_part_bytes(int rank) { return rank > 0 ? (size_t)kGramChunks * rank * rank * 4 : 0; }

int nmfmu_gram(const float* f, int rows, int rank, float* part, float* gram, void* stream) {
  if (!f || !gram || !part || rows <= 0 || rank <= 0 || rank > 256) return NMFMU_ERR_ARG;
  hipLaunchKernelGGL(gram_partial_kernel, dim3(rank, kGramChunks), dim3(256), 0, S(stream), f, rows, rank, part);
  hipLaunchKernelGGL(gram_final_kernel, dim3(rank), dim3(256), 0, S(stream), part, rank, gram);
  return (int)hipGetLastError();
}

int nmfmu_rowmat(const float* owner, int rows, int rank, const float* gram, float* den, int r_pad, void* stream) {
  if (!owner || !gram || !den || rows <= 0 || rank <= 0 || r_pad != nmfmu_pad_rank(rank)) return NMFMU_ERR_ARG;
  hipLaunchKernelGGL(rowmat_kernel, dim3((rows + 3) / 4), dim3(256), 0, S(stream), owner, rows, rank, gram, den, r_pad);
  return (int)hipGetLastError();
}

}  // extern "C"
