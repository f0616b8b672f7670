// HBM / LDS data layouts of the MU engine (host + device; no HIP types here).
//
// The fused kernel sees one MU half-step as
//
//     num[m][r] = sum_k  Gn(X[m][k], S[m][k]) * B[k][r],   S[m][k] = sum_r A[m][r] B[k][r]
//
// with A = the factor being updated ("owner", M rows), B = the other factor
// ("panel", K rows) and X = V (H half-step: M=N, K=C) or V^T (W half-step:
// M=C, K=N).  Reference seam: nmf.py:376-378 / 389-391 (reconstruct +
// _double_backward_update).
//
// Everything below is *designed for the kernel*, not inherited from the
// reference's row-major fp32 tensors:
//
//  Xp  "fragment order" copy of X.  Tile (mb, kt) of 128 owner rows x 64
//      contraction columns is one contiguous block; inside it wave w, load q,
//      lane l owns one 16-byte chunk, so every wave-level load instruction
//      reads 1 KiB of consecutive bytes and lands directly in the lane that
//      holds the matching MFMA accumulator element (no LDS trip for X).
//
//      With G = 1 or 2 thirty-two-row groups per wave (workgroup tile = 128*G rows):
//      chunk index = ((((mb*ktiles + kt)*4 + w)*G + g)*NQ + q)*64 + l      (16 B each)
//      row   m = mb*128*G + w*32*G + g*32 + (l & 31)
//      bf16: NQ = 4, element e (0..7) of chunk q is column  kt*64 + 32*(l>>5) + 8*q + e
//      fp32: NQ = 8, element e (0..3) of chunk q is column  kt*64 + 32*(l>>5) + 4*q + e
//
//  P1  row-major bf16 image of a factor, [rows_pad][R_PAD], with the 16-byte
//      slots of every row XOR-swizzled by the row index so that the
//      ds_read_b128 MFMA-operand reads (same slot, 16 different rows per lane
//      group) are bank-conflict free once a 64-row tile is copied linearly
//      into LDS.
//
//  P2  transposed bf16 image, tiled [rows_pad/64][R_PAD][64]: 128-byte rows
//      (64 consecutive factor rows for one rank column), slots swizzled by the
//      rank index.  It feeds the second GEMM, whose contraction runs over
//      factor rows.
//
//  Both images exist in a "hi" plane (bf16(x)) and, for the split-precision
//  mode, a "lo" plane (bf16(x - hi)).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NMFMU_HD __host__ __device__ __forceinline__
#else
#define NMFMU_HD static inline
#endif

namespace nmfmu {

constexpr int kBM = 128;      // owner rows per workgroup for G = 1 (4 waves x 32); 256 for G = 2
constexpr int kRowPad = 256;  // every factor's row count is padded to this (covers both tile heights and 64-row k tiles)
constexpr int kBK = 64;   // contraction columns per tile
constexpr int kWaves = 4;

// swizzle of P1: physical 16-byte slot = slot ^ p1_swz(row).  The swizzle value is a bijection of the row bits above
// "rows per 256-byte LDS bank line" (1 / 2 / 4 rows per line at padded rank >= 128 / 64 / 32), which is what makes the
// ds_read_b128 operand reads of the first GEMM (16 lanes = 16 rows, one logical slot) conflict free.  Its TOP bits come
// from the LOWEST of those row bits: the transposing reads of the second GEMM (ds_read_b64_tr_b16, nmfmu_pp.h) fetch
// four consecutive rows x 64 bytes per 32-lane pass, and four consecutive rows must land in four different bank
// quarters.  Padded rank 256 (two bank lines per row) takes the rank-128 value: its rank-256 TR path (fused kernel,
// beta = 1) measured SQ_LDS_BANK_CONFLICT at 60 % of SQ_LDS_IDX_ACTIVE with the plain row & 15 it had through round 4
// (profiles/r05n_cfg5_rank256_sq_pmc_summary.txt): four consecutive rows XORed with 0..3 stay inside one quarter.
NMFMU_HD int p1_swz(int row, int r_pad) {
  const int sp = r_pad / 8;                                              // 16-byte slots per row
  if (sp >= 16) return ((row & 3) << 2) | ((row >> 2) & 3);
  if (sp == 8) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  return (row >> 2) & 3;                                                 // sp == 4
}
template <int R_PAD>
struct P1Swz {
  static NMFMU_HD int of(int row) { return p1_swz(row, R_PAD); }
};

// byte offset of element (row, r) inside a P1 plane
NMFMU_HD int64_t p1_offset(int64_t row, int r, int r_pad) {
  const int slot = r >> 3;
  return row * (int64_t)(2 * r_pad) + (int64_t)(((slot ^ p1_swz((int)(row & 63), r_pad)) << 4) + 2 * (r & 7));
}

// byte offset of element (row = contraction index k, r) inside a P2 plane
NMFMU_HD int64_t p2_offset(int64_t row, int r, int r_pad) {
  const int64_t kt = row >> 6;
  const int kl = (int)(row & 63);
  const int slot = kl >> 3;
  return kt * (int64_t)(r_pad * 128) + (int64_t)r * 128 + (int64_t)(((slot ^ ((r >> 1) & 7)) << 4) + 2 * (kl & 7));
}

// element offset (in elements, not bytes) of X[m][k] inside Xp
NMFMU_HD int64_t xp_index(int64_t m, int64_t k, int64_t ktiles, bool fp32, int G) {
  const int bm = 128 * G;
  const int64_t mb = m / bm;
  const int ml = (int)(m % bm);
  const int w = ml / (32 * G);
  const int g = (ml / 32) % G;
  const int j = ml & 31;
  const int64_t kt = k >> 6;
  const int kl = (int)(k & 63);
  const int hl = kl >> 5;
  const int kk = kl & 31;
  const int nq = fp32 ? 8 : 4;
  const int epc = fp32 ? 4 : 8;  // elements per 16-byte chunk
  const int q = kk / epc;
  const int e = kk % epc;
  const int lane = hl * 32 + j;
  return (((((mb * ktiles + kt) * 4 + w) * G + g) * nq + q) * 64 + lane) * epc + e;
}

NMFMU_HD int pad_rows(int rows) { return (rows + kRowPad - 1) / kRowPad * kRowPad; }

NMFMU_HD int pad_rank(int r) {
  if (r <= 0) return -1;
  if (r <= 32) return 32;
  if (r <= 64) return 64;
  if (r <= 128) return 128;
  if (r <= 256) return 256;
  return -1;
}

}  // namespace nmfmu
