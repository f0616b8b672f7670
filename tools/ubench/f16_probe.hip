// Hardware facts the fp16 mode of nmfmu_pp.h relies on (gfx950), printed as text:
//   1. v_fma_mix_f32 op_sel / op_sel_hi: which half of an fp16 pair it reads
//   2. v_cvt_pk_f16_f32: operand order, rounding, saturation under MODE.FP16_OVFL
//   3. v_mfma_f32_32x32x16_f16: operand lane map identical to the bf16 form; fp16 subnormal operands
// hipcc --offload-arch=gfx950 -O2 -std=c++17 f16_probe.hip -o f16_probe && ./f16_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <vector>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__global__ void k_mix(const uint32_t* w, const float* r, float* out, uint32_t* pk, int ovfl) {
  const int i = threadIdx.x;
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  float lo, hi;
  uint32_t p;
  asm volatile("v_fma_mix_f32 %0, %3, %4, 0 op_sel_hi:[1,0,0]\n\t"
               "v_fma_mix_f32 %1, %3, %4, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
               "s_nop 1\n\t"
               "v_cvt_pk_f16_f32 %2, %0, %1"
               : "=&v"(lo), "=&v"(hi), "=&v"(p)
               : "v"(w[i]), "v"(r[i]));
  out[2 * i] = lo, out[2 * i + 1] = hi, pk[i] = p;
}

// D = A (32x16) * B (16x32) with the lane maps of nmfmu_fused.h's probe
__global__ void k_mfma(const uint16_t* a, const uint16_t* b, float* d) {
  const int lane = threadIdx.x, j = lane & 31, hl = lane >> 5;
  u32x4 av, bv;
  for (int i = 0; i < 4; ++i) {
    av[i] = (uint32_t)a[j * 16 + 8 * hl + 2 * i] | ((uint32_t)a[j * 16 + 8 * hl + 2 * i + 1] << 16);
    bv[i] = (uint32_t)b[(8 * hl + 2 * i) * 32 + j] | ((uint32_t)b[(8 * hl + 2 * i + 1) * 32 + j] << 16);
  }
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+v"(acc) : "v"(av), "v"(bv));
  for (int e = 0; e < 16; ++e) d[((e & 3) + 8 * (e >> 2) + 4 * hl) * 32 + j] = acc[e];
}

static uint16_t h16(float x) { _Float16 h = (_Float16)x; uint16_t u; std::memcpy(&u, &h, 2); return u; }
static float f16(uint16_t u) { _Float16 h; std::memcpy(&h, &u, 2); return (float)h; }

int main() {
  // ---- 1 / 2
  const float lo_v[8] = {0.5f, 1.25f, 3e-6f, 0.f, 100.f, 60000.f, 0.1f, 7.f};
  const float hi_v[8] = {2.0f, 0.75f, 5e-5f, 1.f, 200.f, 2.f, 0.3f, 9.f};
  const float r_v[8] = {3.f, 0.5f, 1.f, 4.f, 1000.f, 4.f, 0.333333f, 1e-3f};
  uint32_t hw[64]; float hr[64];
  for (int i = 0; i < 64; ++i) { hw[i] = h16(lo_v[i & 7]) | ((uint32_t)h16(hi_v[i & 7]) << 16); hr[i] = r_v[i & 7]; }
  uint32_t *dw, *dpk; float *dr, *dout;
  hipMalloc(&dw, 256); hipMalloc(&dpk, 256); hipMalloc(&dr, 256); hipMalloc(&dout, 512);
  hipMemcpy(dw, hw, 256, hipMemcpyHostToDevice); hipMemcpy(dr, hr, 256, hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(k_mix, dim3(1), dim3(64), 0, 0, dw, dr, dout, dpk, ovfl);
    float out[128]; uint32_t pk[64];
    hipMemcpy(out, dout, 512, hipMemcpyDeviceToHost); hipMemcpy(pk, dpk, 256, hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d\n", ovfl);
    for (int i = 0; i < 8; ++i)
      printf("  lo=%g hi=%g r=%g : mix_lo=%g (want %g) mix_hi=%g (want %g) | cvt_pk lo=%g hi=%g\n", f16(h16(lo_v[i])), f16(h16(hi_v[i])), r_v[i],
             out[2 * i], f16(h16(lo_v[i])) * r_v[i], out[2 * i + 1], f16(h16(hi_v[i])) * r_v[i], f16(pk[i] & 0xffff), f16(pk[i] >> 16));
  }
  // ---- 3
  std::vector<uint16_t> a(32 * 16), b(16 * 32);
  std::vector<float> af(32 * 16), bf(16 * 32), want(32 * 32, 0.f), got(32 * 32);
  for (int i = 0; i < 32 * 16; ++i) { float v = ((i * 37 % 101) - 50) / 64.f; if (i % 7 == 0) v = 2e-6f * (i % 5 + 1); a[i] = h16(v); af[i] = f16(a[i]); }
  for (int i = 0; i < 16 * 32; ++i) { float v = ((i * 53 % 89) - 44) / 32.f; b[i] = h16(v); bf[i] = f16(b[i]); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += (double)af[i * 16 + k] * bf[k * 32 + j]; want[i * 32 + j] = (float)s; }
  uint16_t *da, *db; float* dd;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 4096);
  hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(got.data(), dd, 4096, hipMemcpyDeviceToHost);
  double maxerr = 0, maxv = 0;
  for (int i = 0; i < 1024; ++i) { maxerr = fmax(maxerr, fabs(got[i] - want[i])); maxv = fmax(maxv, fabs(want[i])); }
  printf("mfma f16 32x32x16 (incl. subnormal A entries): max |err| = %g (max |want| = %g)\n", maxerr, maxv);
  // same with A all subnormal: does the MFMA flush fp16 denormals?
  for (int i = 0; i < 32 * 16; ++i) { a[i] = h16(3e-6f); af[i] = f16(a[i]); }
  hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(got.data(), dd, 4096, hipMemcpyDeviceToHost);
  double s = 0; for (int k = 0; k < 16; ++k) s += (double)af[k] * bf[k * 32 + 0];
  printf("mfma f16 all-subnormal A: got[0][0] = %g want %g  (0 => subnormal operands are flushed)\n", got[0], s);
  return 0;
}
