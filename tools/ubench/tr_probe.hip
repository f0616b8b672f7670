// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds b16 element e = its index; lane l reads at byte address
// addr(l); prints, per lane, the four element indices it received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
__global__ void k(uint32_t* out, int pattern) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (pattern == 0) addr = l * 8;                                  // consecutive 8-byte chunks
  else if (pattern == 1) addr = (l & 15) * 256 + (l >> 4) * 8;     // 16 rows of 256 B; lane group g reads chunk g of every row
  else addr = (l & 3) * 256 + ((l >> 2) & 3) * 8 + (l >> 4) * 1024; // 4 rows x 4 chunks per 16-lane group
  addr += (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[2 * l] = r[0];
  out[2 * l + 1] = r[1];
}
int main() {
  uint32_t* d; hipMalloc(&d, 512);
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p);
    uint32_t h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("pattern %d (element indices; own chunk of lane l starts at element addr/2)\n", p);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d: %4u %4u %4u %4u", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16);
      if (l % 2 == 1) printf("\n");
    }
  }
  return 0;
}
