// How fast can a CU ingest operand tiles?  Every workgroup streams `iters` stages of STAGE bytes from an L2- or
// MALL-resident buffer, either by LDS-DMA (global_load_lds dwordx4) or into registers (global_load_dwordx4), keeping
// DEPTH stages in flight, with nothing else going on.  Prints GB/s per CU and TB/s chip-wide for 1 and 2 workgroups
// per CU.  (Round 2: the NT GEMM of the NMFD path and the ping-pong MU kernel both settle at ~36 GB/s per CU.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int STAGE, bool DMA>
__global__ void __launch_bounds__(THREADS) stream_kernel(const char* __restrict__ src, size_t footprint, int iters,
                                                         uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PER = STAGE / (THREADS * 16);   // 16-byte chunks per thread per stage
  const int tid = threadIdx.x, wave = tid >> 6;
  // every workgroup walks its own window of the buffer (wraps inside the footprint), stages contiguous
  size_t off = ((size_t)blockIdx.x * 7919 * STAGE) % footprint;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const char* base = src + off;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const char* g = base + (size_t)(p * THREADS + tid) * 16;
      if constexpr (DMA) {
        char* dst = smem + (it & 1) * STAGE + p * THREADS * 16 + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(g);
        acc ^= v;
      }
    }
    if constexpr (DMA) {
      // double buffer: wait for the previous stage only
      if (it > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      __syncthreads();
    }
    off += STAGE;
    if (off + STAGE > footprint) off = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 0x12345678u) sink[tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int THREADS, int STAGE, bool DMA>
void run(const char* name, const char* src, size_t footprint, int wg_per_cu, uint32_t* sink) {
  const int iters = 2000, grid = 256 * wg_per_cu;
  auto k = stream_kernel<THREADS, STAGE, DMA>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), DMA ? 2 * STAGE : 0, 0, src, footprint, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * iters * STAGE;
  printf("%-26s footprint %5zu MB  %d WG/CU x %d thr  stage %2d KB: %7.1f GB/s per CU, %6.2f TB/s chip  (%.3f ms)\n", name,
         footprint >> 20, wg_per_cu, THREADS, STAGE >> 10, bytes / ms / 1e6 / 256, bytes / ms / 1e9, ms);
}

int main() {
  const size_t cap = 512ull << 20;
  char* src; uint32_t* sink;
  hipMalloc(&src, cap); hipMemset(src, 1, cap); hipMalloc(&sink, 4096);
  for (size_t fp : {(size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, cap}) {
    for (int w = 1; w <= 2; ++w) {
      run<256, 32768, true>("LDS-DMA", src, fp, w, sink);
      run<256, 32768, false>("registers", src, fp, w, sink);
    }
    run<512, 65536, true>("LDS-DMA", src, fp, 1, sink);
    run<512, 65536, false>("registers", src, fp, 1, sink);
  }
  return 0;
}
