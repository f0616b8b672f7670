// Model of the cross-tile software-pipelined tile body (one wave per SIMD or two), with the real dependencies:
//   phase A: 16 MFMA of GEMM1(t+1) into s_next  ||  elementwise(t) on s_cur (2 and/shl, 2 rcp, 2 mul, 1 cvt_pk per MFMA) + 1 ds_read/MFMA
//   phase B: 16 MFMA of GEMM2(t) (A operand = packed P from phase A) + 1 ds_read/MFMA
// S ping-pongs by unrolling x2.  Reports wall ns per MFMA and the implied chip TF.
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

template <int WAVES, bool FENCE, int MEM>
__global__ void __launch_bounds__(64 * WAVES) model(float* out, const u32x4* xin, int iters, const char* big) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 64 KiB panel area + X ring (MEM >= 4)
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((uint32_t*)lds)[i] = 0x3f803f80u;
  __syncthreads();
  u32x4 q[8]; for (int i = 0; i < 8; ++i) q[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u + (uint32_t)lane, 0x3f803f80u};
  f32x16 sA[2], sB[2], on[4];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) { sA[i][e] = 1.f + e; sB[i][e] = 2.f + e; }
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) on[i][e] = 0.f;
  u32x4 x[4]; for (int i = 0; i < 4; ++i) x[i] = xin[(blockIdx.x * 4 + i) * 64 + lane];
  uint32_t g[2][8];
  u32x4 xn[4], xf[4]; for (int i = 0; i < 4; ++i) { xn[i] = x[i]; xf[i] = x[i]; }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* gsrc = big + ((size_t)blockIdx.x * 4 + wave) * (1 << 20) + lane * 16;   // each wave streams its own MiB-strided region
  int tilecount = 0;
  auto tile = [&](f32x16(&sc)[2], f32x16(&sn)[2]) {
    if (MEM) {   // 8 LDS-DMA pieces (panel from a small, cache-resident region) + 4 X loads (streamed) per tile
#pragma unroll
      for (int p = 0; p < (WAVES == 8 ? 4 : 8); ++p) {   // 32 KiB of panel per tile and workgroup
        const unsigned la = __builtin_amdgcn_readfirstlane(32768u + (unsigned)((tilecount & 1) * 8192 + p * 1024 + (wave & 0) ));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(big + (size_t)(p * 4096 + (tilecount & 63) * 32768) + threadIdx.x * 16), "s"(la) : "memory", "m0");
      }
      if (MEM == 2) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) x[qq] = __builtin_nontemporal_load((const u32x4*)(gsrc + (size_t)(tilecount & 63) * 16384 + qq * 1024));
      }
      if (MEM >= 4) {   // X through an LDS ring by DMA, D = MEM - 3 tiles ahead; this wave's 4 KiB piece of slot (t + D) % (D + 1)
        constexpr int D = MEM - 3;
        const unsigned slot = (unsigned)((tilecount + D) % (D + 1));
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const unsigned la = __builtin_amdgcn_readfirstlane(65536u + slot * (WAVES * 4096u) + (unsigned)wave * 4096u + qq * 1024u);
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc + (size_t)(tilecount & 63) * 16384 + qq * 1024), "s"(la) : "memory", "m0");
        }
        const unsigned rs = (unsigned)(tilecount % (D + 1));
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) x[qq] = *(const u32x4*)(lds + 65536 + rs * (WAVES * 4096) + wave * 4096 + qq * 1024 + lane * 16);
      }
      if (MEM == 3) {   // X two tiles ahead: rotate (moves are free beside MFMAs), load the far set
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { x[qq] = xn[qq]; xn[qq] = xf[qq]; }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) xf[qq] = __builtin_nontemporal_load((const u32x4*)(gsrc + (size_t)(tilecount & 63) * 16384 + qq * 1024));
      }
      ++tilecount;
    }
    // phase A
    u32x4 ring[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) ring[p] = *(const u32x4*)(lds + ((lane * 16 + p * 1024) & 32767));
#pragma unroll
    for (int step = 0; step < 16; ++step) {
      const int tt = step & 1, kk = step >> 1;
      const u32x4 ah = ring[step & 3];
      if (step + 4 < 16) ring[step & 3] = *(const u32x4*)(lds + ((lane * 16 + (step + 4) * 1024) & 32767));
      sn[tt] = mf(ah, q[kk], sn[tt]);
      // elementwise pair `step` of the CURRENT tile
      const int t2 = step >> 3, d = step & 7;
      const uint32_t w = x[2 * t2 + (d >> 2)][d & 3];
      const float x0 = __builtin_bit_cast(float, w << 16), x1 = __builtin_bit_cast(float, w & 0xffff0000u);
      const float n0 = x0 * __builtin_amdgcn_rcpf(sc[t2][2 * d]), n1 = x1 * __builtin_amdgcn_rcpf(sc[t2][2 * d + 1]);
      g[t2][d] = pk(n0, n1);
      if (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
    // phase B
#pragma unroll
    for (int p = 0; p < 4; ++p) ring[p] = *(const u32x4*)(lds + ((lane * 16 + p * 1024 + 16384) & 32767));
#pragma unroll
    for (int step = 0; step < 16; ++step) {
      const int rt = step & 3, c = step >> 2, t2 = c >> 1, m2 = c & 1;
      const u32x4 bh = ring[step & 3];
      if (step + 4 < 16) ring[step & 3] = *(const u32x4*)(lds + ((lane * 16 + (step + 4) * 1024 + 16384) & 32767));
      const u32x4 nh = {g[t2][4 * m2], g[t2][4 * m2 + 1], g[t2][4 * m2 + 2], g[t2][4 * m2 + 3]};
      on[rt] = mf(nh, bh, on[rt]);
      if (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
    if (MEM >= 5) { __builtin_amdgcn_s_waitcnt(4 | (7 << 4) | (15 << 8)); __builtin_amdgcn_s_barrier(); }
    else if (MEM == 3) { __builtin_amdgcn_s_waitcnt(4 | (7 << 4) | (15 << 8)); __builtin_amdgcn_s_barrier(); }
    else if (MEM) { __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8)); __syncthreads(); }
  };
  for (int it = 0; it < iters; it += 2) {
    tile(sA, sB);
#pragma unroll
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) sA[i][e] = 1.2e-7f;
    tile(sB, sA);
#pragma unroll
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) sB[i][e] = 1.2e-7f;
  }
  float s = 0; for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += on[i][e];
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) s += sA[i][e] + sB[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES, bool FENCE, int MEM>
void run(float* out, const u32x4* xin, const char* big) {
  const int iters = 2000, blocks = 256;
  const size_t ldsb = 65536 + (MEM >= 4 ? (MEM - 2) * WAVES * 4096 : 0);
  hipFuncSetAttribute(reinterpret_cast<const void*>(model<WAVES, FENCE, MEM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL((model<WAVES, FENCE, MEM>), dim3(blocks), dim3(64 * WAVES), ldsb, 0, out, xin, iters, big);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((model<WAVES, FENCE, MEM>), dim3(blocks), dim3(64 * WAVES), ldsb, 0, out, xin, iters, big);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 32;
  printf("software-pipelined model, waves/CU=%d fence=%d mem=%d: %.1f ns/tile/wave, %.2f ns per MFMA per SIMD -> %.0f TF chip-wide\n", WAVES, (int)FENCE, MEM,
         ms * 1e6 / iters, ms * 1e6 / mfmas / (WAVES / 4), 256.0 * WAVES * mfmas * 32768 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; u32x4* xin;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&xin, 256 * 4 * 64 * 16); hipMemset(xin, 0x3f, 256 * 4 * 64 * 16);
  char* big; hipMalloc(&big, (size_t)2100 << 20); hipMemset(big, 0x3f, (size_t)2100 << 20);
  run<4, true, 0>(out, xin, big); run<4, true, 1>(out, xin, big); run<4, true, 2>(out, xin, big);
  run<8, true, 0>(out, xin, big); run<8, true, 1>(out, xin, big); run<8, true, 2>(out, xin, big);
  run<4, true, 3>(out, xin, big); run<8, true, 3>(out, xin, big);
  run<8, true, 4>(out, xin, big); run<8, true, 5>(out, xin, big); run<8, true, 6>(out, xin, big);
  run<4, true, 4>(out, xin, big); run<4, true, 5>(out, xin, big); run<4, true, 6>(out, xin, big);
  return 0;
}
