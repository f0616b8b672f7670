// Does HBM streaming slow the MFMA pipe when nothing depends on the data?  Every wave runs the MFMA-only loop of
// mfma_peak.hip (random bf16 operands, 32 MFMAs per "tile") and additionally streams NLOAD KiB per tile from HBM
// by LDS-DMA (never read back, never waited for except to bound the number in flight to three tiles).  No barriers,
// no LDS reads, no dependency between the stream and the matrix work.  NLOAD = 4 is the MU kernel's ratio
// (4 KiB of X per wave and tile of 32 MFMAs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
__device__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
template <int NLOAD, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, const char* big, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  uint32_t s = blockIdx.x * 1024 + threadIdx.x + 12345u;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32x4 a[8], b[8];
  for (int i = 0; i < 8; ++i)
    for (int c = 0; c < 4; ++c) { a[i][c] = (rnd(s) & 0x007f007fu) | 0x3f003f00u; b[i][c] = (rnd(s) & 0x007f007fu) | 0x3f003f00u; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds) + wave * 8192u;
  // each wave streams its own contiguous region (tiles x NLOAD KiB), regions packed back to back: 1 GiB-scale total
  const char* src = big + ((size_t)blockIdx.x * WAVES + wave) * ((size_t)tiles * (NLOAD ? NLOAD : 1) * 1024) + lane * 16;
  for (int t = 0; t < tiles; ++t) {
    if (NLOAD) {
#pragma unroll
      for (int p = 0; p < NLOAD; ++p) {
        const unsigned la = lds_base + (unsigned)((p & 7) * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(src + (size_t)(t * NLOAD + p) * 1024), "s"(la) : "memory", "m0");
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLOAD > 63 ? 63 : 3 * NLOAD) : "memory");
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[0] = mf(a[i], b[i], acc[0]); acc[1] = mf(a[i], b[(i + 1) & 7], acc[1]);
      acc[2] = mf(a[(i + 2) & 7], b[i], acc[2]); acc[3] = mf(a[(i + 3) & 7], b[(i + 5) & 7], acc[3]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float tt = 0; for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) tt += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = tt;
}
template <int NLOAD, int WAVES>
void run(float* out, const char* big) {
  const int tiles = 512, blocks = 256 * (8 / WAVES);   // 8 waves per CU; NLOAD=4: 256*8*512*4 KiB = 4 GiB streamed
  auto kern = k<NLOAD, WAVES>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * 8192);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WAVES), WAVES * 8192, 0, out, big, tiles);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WAVES), WAVES * 8192, 0, out, big, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)blocks * WAVES * tiles * 32 * 32768.0;
  const double by = (double)blocks * WAVES * tiles * NLOAD * 1024.0;
  printf("MFMA (random bf16) + %d KiB/tile/wave HBM stream, %d waves/WG: %.3f ms -> %.0f TFLOP/s and %.2f TB/s (err %d)\n", NLOAD, WAVES, ms,
         fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e12, (int)hipGetLastError());
}
int main() {
  float* out; hipMalloc(&out, 512 * 512 * 4);
  char* big; hipMalloc(&big, (size_t)8200 << 20); hipMemset(big, 0x3c, (size_t)8200 << 20);
  run<0, 4>(out, big); run<1, 4>(out, big); run<2, 4>(out, big); run<4, 4>(out, big); run<8, 4>(out, big);
  run<0, 8>(out, big); run<4, 8>(out, big);
  return 0;
}
