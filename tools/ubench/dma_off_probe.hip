// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
// (round 6: the software-pipelined rank-256 kernel issues its panel pieces with ONE M0 value per group of four pieces if it does)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm dma_off_probe.hip -o dma_off_probe && ./dma_off_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) probe(const unsigned* src, unsigned* out, int off_sel) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* l = reinterpret_cast<unsigned*>(smem);
  for (int i = threadIdx.x; i < 4096; i += 64) l[i] = 0xffffffffu;
  __syncthreads();
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
  const unsigned voff = threadIdx.x * 16u;
  if (off_sel == 0)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds) : "memory", "m0");
  else if (off_sel == 1)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(src), "s"(lds) : "memory", "m0");
  else
    asm volatile("s_add_u32 m0, %2, 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(voff), "s"(src), "s"(lds) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = l[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, 16384); hipMalloc(&o, 16384);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  for (int sel = 0; sel < 3; ++sel) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, d, o, sel);
    std::vector<unsigned> r(4096);
    hipMemcpy(r.data(), o, 16384, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 4096; ++i) if (r[i] != 0xffffffffu) { if (first < 0) first = i; last = i; }
    printf("sel %d: LDS dwords [%d..%d] written (byte %d), first value = source dword %u (byte %u)\n", sel, first, last, first * 4,
           first >= 0 ? r[first] : 0u, first >= 0 ? r[first] * 4 : 0u);
  }
  return 0;
}
