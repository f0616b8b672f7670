// Model of a ROLE-SPLIT ("warp specialised") tile pipeline: 8 waves per workgroup, two per SIMD.
//   producer waves 0-3: GEMM1(t) (16 MFMA, panel operand from LDS) -> elementwise(t) -> Gn(t) written to LDS
//   consumer waves 4-7: GEMM2(t) (16 MFMA; A operand = Gn(t) read back from LDS, B operand = panel from LDS)
// One workgroup barrier per tile; the Gn exchange buffer is double buffered, so the consumer works on tile t while
// the producer is already on tile t+1: the elementwise stage of one wave runs beside the other wave's MFMAs
// without any intra-wave interleaving.  MEM: 0 none, 1 panel LDS-DMA (L2 resident) drained per tile,
// 2 + HBM X stream (producers only) one tile ahead, 3 X two tiles ahead with a counted vmcnt.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
__device__ __forceinline__ uint32_t pk(float a, float b) { f32x2 v = {a, b}; bf16x2 r = __builtin_convertvector(v, bf16x2); return __builtin_bit_cast(uint32_t, r); }

template <int MEM, int PAIRS>
__global__ void __launch_bounds__(128 * PAIRS) model(float* out, const u32x4* xin, int iters, const char* big) {
  // LDS: [0,32K) panel stage 0, [32K,64K) panel stage 1, [64K, 64K+2*PAIRS*4K) Gn exchange (2 slots x PAIRS waves x 4 KB)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((uint32_t*)lds)[i] = 0x3f803f80u;
  __syncthreads();
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
  constexpr int NDMA = 32768 / (128 * PAIRS * 16);   // DMA instructions per thread per tile (32 KB panel stage)
  auto dma = [&](int t) {
    if (MEM) {
#pragma unroll
      for (int p = 0; p < NDMA; ++p) {
        const unsigned la = lds_base + (unsigned)((t & 1) * 32768 + p * (128 * PAIRS * 16)) + (unsigned)wave * 1024u;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(big + (size_t)(p * (128 * PAIRS * 16) + (t & 63) * 32768) + threadIdx.x * 16), "s"(la) : "memory", "m0");
      }
    }
  };
  const bool producer = wave < PAIRS;
  const int pw = producer ? wave : wave - PAIRS;
  char* xch = lds + 65536 + pw * 4096 + lane * 32;   // + slot * PAIRS * 4096
  float acc = 0.f;
  if (producer) {
    u32x4 q[8]; for (int i = 0; i < 8; ++i) q[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u + (uint32_t)lane, 0x3f803f80u};
    f32x16 eps; for (int e = 0; e < 16; ++e) eps[e] = 1.2e-7f;
    u32x4 x[4], xn[4], xf[4];
    for (int i = 0; i < 4; ++i) { x[i] = xin[(blockIdx.x * 4 + i) * 64 + lane]; xn[i] = x[i]; xf[i] = x[i]; }
    const char* gsrc = big + ((size_t)blockIdx.x * 4 + wave) * (1 << 20) + lane * 16;
    for (int t = 0; t < iters; ++t) {
      dma(t + 1);
      if (MEM == 2) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { x[qq] = xn[qq]; xn[qq] = __builtin_nontemporal_load((const u32x4*)(gsrc + (size_t)(t & 63) * 16384 + qq * 1024)); }
      }
      if (MEM == 3) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { x[qq] = xn[qq]; xn[qq] = xf[qq]; }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) xf[qq] = __builtin_nontemporal_load((const u32x4*)(gsrc + (size_t)(t & 63) * 16384 + qq * 1024));
      }
      const char* sb = lds + (t & 1) * 32768;
      f32x16 s[2];
      u32x4 ring[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) ring[p] = *(const u32x4*)(sb + ((lane * 16 + p * 1024) & 16383));
#pragma unroll
      for (int step = 0; step < 16; ++step) {
        const int tt = step & 1, kk = step >> 1;
        const u32x4 ah = ring[step & 3];
        if (step + 4 < 16) ring[step & 3] = *(const u32x4*)(sb + ((lane * 16 + (step + 4) * 1024) & 16383));
        s[tt] = mf(ah, q[kk], kk == 0 ? eps : s[tt]);
      }
      uint32_t g[2][8];
#pragma unroll
      for (int step = 0; step < 16; ++step) {
        const int t2 = step >> 3, d = step & 7;
        const uint32_t w = x[2 * t2 + (d >> 2)][d & 3];
        const float x0 = __builtin_bit_cast(float, w << 16), x1 = __builtin_bit_cast(float, w & 0xffff0000u);
        g[t2][d] = pk(x0 * __builtin_amdgcn_rcpf(s[t2][2 * d]), x1 * __builtin_amdgcn_rcpf(s[t2][2 * d + 1]));
      }
      char* dst = xch + (t & 1) * PAIRS * 4096;
      *(u32x4*)(dst) = u32x4{g[0][0], g[0][1], g[0][2], g[0][3]};
      *(u32x4*)(dst + 16) = u32x4{g[0][4], g[0][5], g[0][6], g[0][7]};
      *(u32x4*)(dst + 2048) = u32x4{g[1][0], g[1][1], g[1][2], g[1][3]};
      *(u32x4*)(dst + 2048 + 16) = u32x4{g[1][4], g[1][5], g[1][6], g[1][7]};
      if (MEM == 3) { __builtin_amdgcn_s_waitcnt(4 | (7 << 4) | (0 << 8)); __builtin_amdgcn_s_barrier(); }
      else { __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8)); __builtin_amdgcn_s_barrier(); }
    }
    for (int i = 0; i < 4; ++i) acc += __builtin_bit_cast(float, x[i][0]);
  } else {
    f32x16 on[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) on[i][e] = 0.f;
    for (int t = 0; t < iters; ++t) {
      dma(t + 1);
      if (t > 0) {
        const char* src = xch + ((t - 1) & 1) * PAIRS * 4096;
        const u32x4 g00 = *(const u32x4*)(src), g01 = *(const u32x4*)(src + 16), g10 = *(const u32x4*)(src + 2048), g11 = *(const u32x4*)(src + 2048 + 16);
        const char* sb = lds + ((t - 1) & 1) * 32768 + 16384;
        u32x4 ring[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) ring[p] = *(const u32x4*)(sb + ((lane * 16 + p * 1024) & 16383));
#pragma unroll
        for (int step = 0; step < 16; ++step) {
          const int rt = step & 3, c = step >> 2;
          const u32x4 bh = ring[step & 3];
          if (step + 4 < 16) ring[step & 3] = *(const u32x4*)(sb + ((lane * 16 + (step + 4) * 1024) & 16383));
          const u32x4 nh = c == 0 ? g00 : c == 1 ? g01 : c == 2 ? g10 : g11;
          on[rt] = mf(nh, bh, on[rt]);
        }
      }
      __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (0 << 8));
      __builtin_amdgcn_s_barrier();
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc += on[i][e];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MEM, int PAIRS>
void run(float* out, const u32x4* xin, const char* big) {
  const int iters = 2000, blocks = 256;
  const size_t lds = 65536 + 2 * PAIRS * 4096;
  auto k = model<MEM, PAIRS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(128 * PAIRS), lds, 0, out, xin, iters, big);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(128 * PAIRS), lds, 0, out, xin, iters, big);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 16 * 2 * PAIRS;   // per workgroup (= per CU)
  printf("role-split model, %d producer + %d consumer waves/CU, mem=%d: %.1f ns/tile, err=%d -> %.0f TF chip-wide\n", PAIRS, PAIRS, MEM,
         ms * 1e6 / iters, (int)hipGetLastError(), 256.0 * mfmas * 32768 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; u32x4* xin;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&xin, 256 * 4 * 64 * 16); hipMemset(xin, 0x3f, 256 * 4 * 64 * 16);
  char* big; hipMalloc(&big, (size_t)2100 << 20); hipMemset(big, 0x3f, (size_t)2100 << 20);
  run<0, 4>(out, xin, big); run<1, 4>(out, xin, big); run<2, 4>(out, xin, big); run<3, 4>(out, xin, big);
  run<0, 8>(out, xin, big); run<1, 8>(out, xin, big); run<2, 8>(out, xin, big); run<3, 8>(out, xin, big);
  return 0;
}
