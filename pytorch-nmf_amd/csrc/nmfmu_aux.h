// Launchers of the auxiliary (non-MFMA) kernels; see nmfmu_aux.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmfmu {

struct ApplyArgs {
  float* f;             // fp32 master [rows][rank]
  const float* num;     // [nslab][rows_pad][R_PAD]
  const float* den;     // same or nullptr
  const float* kl_den;  // [R_PAD] or nullptr
  int nslab;
  void* p1_hi;
  void* p1_lo;
  void* p2_hi;
  void* p2_lo;
  float* colsum_part;   // [rows_pad/16][R_PAD] (worst case)
  float* colsum;        // [R_PAD]
  int rows, rank, rows_pad;
  float l1, l2, gamma;
  // trainer.BetaMu semantics (trainer.py:93-112) instead of fit's (nmf.py:78-92): eps added after the penalties,
  // orthogonality penalty, and grad[row][r] = relu(den) - relu(num) written out (p.grad of trainer.py:98)
  const float* scale;   // PACK_ONLY: images (and column sums) of f[row][r] * scale[r]; the master is not touched
  int trainer;
  float ortho;
  float* grad;          // [rows][rank] or nullptr
  int f16;              // images in fp16 (clamped to 65504) instead of bf16; no lo planes
  uint32_t* status;     // or nullptr: bit 0 is set when an fp16 image value had to be clamped
  int den_nslab;        // slabs of `den` when it differs from nslab (kModeXB leaves ONE denominator slab); 0 = nslab
  int skip_colsum;      // do not launch the column-sum finalize (beta == 2: nothing reads the column sums)
};

int apply_stripe_rows(int rows_pad);   // rows per apply workgroup (= rows per column-sum partial): 16 or 64

// fmt: 0 = bf16, 1 = fp32, 2 = fp16 (clamped to 65504)
int launch_pack_x(const float* v, int64_t ld, int rows, int cols, bool transpose, int fmt, void* xp, int m_pad,
                  int k_pad, uint32_t* flags, int G, hipStream_t s);
int launch_apply(int r_pad, const ApplyArgs& a, bool x3, bool pack_only, hipStream_t s);
int launch_colsum_finalize(const float* part, int nblk, int r_pad, float* out, hipStream_t s);
int launch_slab_reduce(const float* slab, int nslab, int64_t plane, float* out, hipStream_t s);
int launch_sum_finalize_f32(const float* part, int n, double* out, hipStream_t s);
int launch_target_sums(const float* v, int64_t ld, int rows, int cols, double* part, int nparts, double* out4, hipStream_t s);
int launch_riding_finish(const float* part, int npairs, const double* ac, double n_all, const uint32_t* status, double* out2,
                         hipStream_t s);
int launch_checkpoint(const float* part, int n, const uint32_t* status, double* out, const float* a, float* a_snap, int64_t na,
                      const float* b, float* b_snap, int64_t nb, hipStream_t s);
int launch_beta_div(const float* x, const float* y, int64_t n, float beta, int kind, double* part, double* out,
                    hipStream_t s);
int launch_mu_terms(const float* s, const float* v, int64_t n, float beta, int kind, float* gn, float* gp, hipStream_t st);
int launch_trainer_update(float* f, int rows, int cols, const float* neg, const float* pos, float l1, float l2, float ortho,
                          float gamma, float* grad, hipStream_t st);
int launch_norms(const float* x, int64_t n, double* part, double* out, hipStream_t s);
int launch_reconstruct(const float* A, int M, const float* B, int K, int R, float* out, int64_t ld, hipStream_t s);
int launch_probe_mfma(const uint16_t* a, const uint16_t* b, float* d, hipStream_t s);
int launch_probe_lds_dma(const uint32_t* src, uint32_t* dst, int n_dwords, hipStream_t s);
int launch_ubench_mfma_hbm(const void* operands, size_t operand_bytes, int f16, const void* stream_src, int kib_per_tile,
                           int waves, int tiles, int grid, float* out, hipStream_t s, int wrap_tiles = 0,
                           unsigned long long* stamps = nullptr);

}  // namespace nmfmu
