// MFMA-only ceiling as a function of operand DATA: the sustained clock of the MI355X under v_mfma_f32_32x32x16_bf16
// depends on how many operand bits toggle.  Constant operands (what most micro-benchmarks use) vs random bf16 in
// [0, 1) (what the MU kernel multiplies).  4 independent accumulator chains per wave, operands in registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
__device__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ uint32_t bf16pair01(uint32_t& s) {   // two bf16 values uniform in [0, 1): exponent 0x3f (x - 1 of [1,2) not needed for power purposes)
  const uint32_t r = rnd(s);
  return (0x3f00u | ((r >> 3) & 0x7fu)) | ((0x3f00u | ((r >> 13) & 0x7fu) | ((r >> 20) & 0x80u ? 0x0080u : 0)) << 16);
}
template <int DATA, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, int iters) {
  uint32_t s = blockIdx.x * 1024 + threadIdx.x + 12345u;
  u32x4 a[8], b[8];
  for (int i = 0; i < 8; ++i)
    for (int c = 0; c < 4; ++c) {
      a[i][c] = DATA == 0 ? 0x3f803f80u : DATA == 1 ? bf16pair01(s) : rnd(s) & 0x7fff7fffu & ~0x40004000u;   // 2: random mantissa+exponent (<2)
      b[i][c] = DATA == 0 ? 0x3f803f80u : DATA == 1 ? bf16pair01(s) : rnd(s) & 0x7fff7fffu & ~0x40004000u;
    }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[0] = mf(a[i], b[i], acc[0]); acc[1] = mf(a[i], b[(i + 1) & 7], acc[1]);
      acc[2] = mf(a[(i + 2) & 7], b[i], acc[2]); acc[3] = mf(a[(i + 3) & 7], b[(i + 5) & 7], acc[3]);
    }
  }
  float t = 0; for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) t += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <int DATA, int WAVES>
void run(float* out) {
  const int iters = 4000, blocks = 256 * (8 / WAVES);
  hipLaunchKernelGGL((k<DATA, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<DATA, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double fl = (double)blocks * WAVES * iters * 32 * 32768.0;
  printf("MFMA only, data=%s, %d waves/workgroup x %d workgroups: %.3f ms -> %.0f TFLOP/s\n",
         DATA == 0 ? "constant 1.0" : DATA == 1 ? "random bf16 in [0.5,1)" : "random bits", WAVES, blocks, ms, fl / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; hipMalloc(&out, 512 * 512 * 4);
  run<0, 4>(out); run<1, 4>(out); run<2, 4>(out);
  run<0, 8>(out); run<1, 8>(out); run<2, 8>(out);
  return 0;
}
