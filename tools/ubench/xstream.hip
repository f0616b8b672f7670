// Ceiling of the X stream alone: the fused kernel's exact access pattern (per workgroup a contiguous run of 16 KiB
// tiles, wave w reads 4 x 1 KiB pieces at w*4096 + q*1024, non-temporal), no compute.  Prints TB/s and the MU
// TFLOP/s that rate would correspond to at rank 128 (256 flop per X byte).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
template <int DEPTH>
__global__ void __launch_bounds__(256, 2) xs(const char* xp, int tiles, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = xp + (size_t)blockIdx.x * tiles * 16384 + wave * 4096 + lane * 16;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[DEPTH][4];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int q = 0; q < 4; ++q) buf[d][q] = __builtin_nontemporal_load((const u32x4*)(base + (size_t)d * 16384 + q * 1024));
  for (int t = 0; t < tiles; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc ^= buf[d][q];
      const int tn = t + d + DEPTH < tiles ? t + d + DEPTH : t + d;
#pragma unroll
      for (int q = 0; q < 4; ++q) buf[d][q] = __builtin_nontemporal_load((const u32x4*)(base + (size_t)tn * 16384 + q * 1024));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}
template <int DEPTH>
void run(const char* xp, uint32_t* out, int blocks, int tiles) {
  hipLaunchKernelGGL(xs<DEPTH>, dim3(blocks), dim3(256), 0, 0, xp, tiles, out);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(xs<DEPTH>, dim3(blocks), dim3(256), 0, 0, xp, tiles, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)blocks * tiles * 16384;
  printf("X stream only, %d WGs x %d tiles, %d tile(s) in flight per wave: %.3f ms, %.2f TB/s -> %.0f TF-equivalent at rank 128\n", blocks, tiles, DEPTH, ms,
         bytes / (ms * 1e-3) / 1e12, bytes * 256 / (ms * 1e-3) / 1e12);
}
int main() {
  const int blocks = 512, tiles = 64;   // = BASELINE configs[1]: 32 owner blocks x 16 splits, 64 tiles each (512 MiB)
  char* xp; hipMalloc(&xp, (size_t)blocks * tiles * 16384 + (1 << 20)); hipMemset(xp, 1, (size_t)blocks * tiles * 16384);
  uint32_t* out; hipMalloc(&out, blocks * 256 * 4);
  run<1>(xp, out, blocks, tiles); run<2>(xp, out, blocks, tiles); run<4>(xp, out, blocks, tiles);
  run<1>(xp, out, 2048, 16); run<2>(xp, out, 2048, 16); run<4>(xp, out, 2048, 16);
  return 0;
}
