// Microbenchmark: how many cycles does one v_mfma_f32_32x32x16_bf16 cost when F filler instructions of a given kind
// sit between consecutive MFMAs of ONE wave per SIMD?   hipcc --offload-arch=gfx950 -O3 mfma_fill.hip -o mfma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

enum { K_NONE, K_VALU, K_RCP, K_CVT, K_DSREAD, K_MIX, K_PKMUL, K_BITOP, K_MOV, K_SALU };

template <int KIND, int F, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) bench(float* out, uint64_t* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int lane = threadIdx.x & 63;
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v[8]; for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 0.001f + i;
  u32x4 r[4]; for (int i = 0; i < 4; ++i) r[i] = a;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((uint32_t*)lds)[i] = 0x3f803f80u;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const int k = (m * F + f) & 7;
        if constexpr (KIND == K_VALU) v[k] = v[k] * 1.0001f + 0.5f;
        if constexpr (KIND == K_RCP) v[k] = __builtin_amdgcn_rcpf(v[k]);
        if constexpr (KIND == K_CVT) { typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2; f2 x = {v[k], v[(k + 1) & 7]}; b2 y = __builtin_convertvector(x, b2); v[k] = __builtin_bit_cast(float, y) ; }
        if constexpr (KIND == K_DSREAD) r[k & 3] = *(const u32x4*)(lds + ((lane * 16 + k * 1024 + (int)r[k & 3][0] * 0) & 16383));
        if constexpr (KIND == K_PKMUL) { typedef __attribute__((ext_vector_type(2))) float f2; f2 x = {v[k], v[(k + 1) & 7]}, y = {1.0001f, 0.9999f}; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y)); v[k] = x[0]; v[(k + 1) & 7] = x[1]; }
        if constexpr (KIND == K_BITOP) { uint32_t u = __builtin_bit_cast(uint32_t, v[k]); asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u) : "v"(u)); v[k] = __builtin_bit_cast(float, u); }
        if constexpr (KIND == K_MOV) { float u; asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(v[k])); v[k] = u; }
        if constexpr (KIND == K_SALU) { int u; asm volatile("s_add_i32 %0, %1, 1" : "=s"(u) : "s"(it) : "scc"); asm volatile("" :: "s"(u)); }
        if constexpr (KIND == K_MIX) { if ((f & 3) == 0) v[k] = __builtin_amdgcn_rcpf(v[k]); else if ((f & 3) == 3) r[k & 3] = *(const u32x4*)(lds + ((lane * 16 + k * 1024) & 16383)); else v[k] = v[k] * 1.0001f + 0.5f; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += (float)r[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int F, int WAVES>
void run(const char* name, float* out, uint64_t* cyc) {
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL((bench<KIND, F, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((bench<KIND, F, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i]; c /= 256;
  const double mf = (double)iters * 16;
  printf("%-8s F=%d waves/CU=%d: %.1f cycles(readcyclecounter)/MFMA  wall %.3f ms -> %.1f ns/MFMA/wave  (%.0f TF chip-wide)\n", name, F, WAVES,
         c / mf, ms, ms * 1e6 / mf, 256.0 * WAVES * mf * 32768 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  run<K_NONE, 0, 4>("none", out, cyc);
  run<K_NONE, 0, 8>("none", out, cyc);
  run<K_VALU, 2, 4>("valu", out, cyc); run<K_VALU, 4, 4>("valu", out, cyc); run<K_VALU, 6, 4>("valu", out, cyc); run<K_VALU, 8, 4>("valu", out, cyc);
  run<K_VALU, 8, 8>("valu", out, cyc);
  run<K_RCP, 1, 4>("rcp", out, cyc); run<K_RCP, 2, 4>("rcp", out, cyc); run<K_RCP, 4, 4>("rcp", out, cyc);
  run<K_RCP, 2, 8>("rcp", out, cyc);
  run<K_CVT, 1, 4>("cvt_pk", out, cyc); run<K_CVT, 2, 4>("cvt_pk", out, cyc); run<K_CVT, 4, 4>("cvt_pk", out, cyc);
  run<K_DSREAD, 1, 4>("ds_b128", out, cyc); run<K_DSREAD, 2, 4>("ds_b128", out, cyc);
  run<K_DSREAD, 1, 8>("ds_b128", out, cyc);
  run<K_PKMUL, 1, 4>("pk_mul", out, cyc); run<K_PKMUL, 2, 4>("pk_mul", out, cyc); run<K_PKMUL, 4, 4>("pk_mul", out, cyc);
  run<K_BITOP, 2, 4>("v_and", out, cyc); run<K_BITOP, 4, 4>("v_and", out, cyc);
  run<K_MOV, 2, 4>("v_mov", out, cyc); run<K_MOV, 4, 4>("v_mov", out, cyc);
  run<K_SALU, 4, 4>("s_add", out, cyc); run<K_SALU, 8, 4>("s_add", out, cyc);
  run<K_MIX, 4, 4>("mix", out, cyc); run<K_MIX, 8, 4>("mix", out, cyc); run<K_MIX, 8, 8>("mix", out, cyc);
  return 0;
}
