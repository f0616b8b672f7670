// NT GEMM for the convolutive (NMFD) path:  D[m][n] = sum_k A[m][k] * B[n][k]
//
// NMFD (nmf.py:700-779 of the reference) is dense NMF on unfolded operands with an effective rank R*T (3200 at
// BASELINE configs[3]), far too wide to keep the second GEMM's accumulators in registers the way the dense fused
// kernel does.  Its MU iteration is therefore four plain GEMMs (reconstruction for each half-step, W numerator,
// H numerator before folding) whose elementwise work rides in the epilogue:
//
//   EPI_RATIO  D = S:  Gn = f(S, X) (and Gp for beta != 1) written as bf16 (hi[, lo]) planes, row-major [m][n]
//   EPI_F32    D written as fp32 row-major
//   EPI_LOSS   beta-divergence of D against X, one partial per workgroup
//   EPI_FOLD   D = Y[(r,t)][(b,l)] is never stored: the tile's diagonal sums (l - t constant) go out instead, 1 KiB per
//              (r, b) segment of the tile; nmfmu_conv_fold_parts_apply_h gathers them (the col2im sum of the conv1d
//              backward pass wrt H without the 4 * R*T * B*L bytes write + read of Y)
//
// Round 3 added two things around the k loop: the tail-round split of the EPI_FOLD launch (GemmArgs::tail_rows) and the
// ragged channels of the reconstruction launches as one extra 16 x 16 x 32 MFMA block per workgroup (GemmArgs::rag_*).
//
// Both operands are bf16 planes (hi[, lo]) with k contiguous, zero padded to multiples of 128 in every dimension.
// 128x128 block tile, 4 waves (2x2, 64x64 each = 2x2 MFMA 32x32x16 tiles),
// BK = 64, LDS double buffered by LDS-DMA.  The 128-byte LDS rows are XOR-swizzled on the DMA *source* side (linear LDS destination) and on the
// ds_read side, which makes every ds_read_b128 conflict free (same analysis as the P2 image of nmfmu_layout.h).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <type_traits>

#include "nmfmu_fused.h"

namespace nmfmu {

enum GemmEpi : int { kEpiRatio = 0, kEpiF32 = 1, kEpiLoss = 2, kEpiFold = 3 };
constexpr int kFoldLd = 129;   // LDS pitch (floats) of the EPI_FOLD tile

struct GemmArgs {
  const uint16_t* a_hi;  // [m_pad][k_pad]
  const uint16_t* a_lo;
  const uint16_t* b_hi;  // [n_pad][k_pad]
  const uint16_t* b_lo;
  int m_pad, n_pad, k_pad;
  int k_len;             // contraction length actually run (multiple of 64, <= k_pad = the planes' row pitch)
  int k_split;           // EPI_F32: gridDim.z workgroups share the contraction; slab z of out = its partial sum
  int tail_rows;         // EPI_FOLD: the last tail_rows tile rows are contraction-split k_split ways (tail-round split:
                         // a grid of 3.1 rounds then costs 3.1 + a fraction instead of 4); slab z of out = partial sums
  // epilogue operands
  const float* x;        // [m_pad][n_pad] fp32 (ratio / loss)
  uint16_t* gn_hi;       // ratio outputs, [m_pad][n_pad]
  uint16_t* gn_lo;
  uint16_t* gp_hi;
  uint16_t* gp_lo;
  float* out;            // EPI_F32: [m_pad][n_pad];  EPI_LOSS: [gridDim.x * gridDim.y] partials
  int m_valid, n_valid;  // loss masking
  int ldn;               // leading dimension of x / gn / gp / out (>= n_pad; a GEMM over the first n_pad columns of wider planes)
  float beta;
  // implicit Toeplitz operand (OPS != kOpsPlanes): the operand's *_hi / *_lo point to a window table
  int tB, tR, tT, tLh;   // H is (B, R, Lh), T taps
  // ... with more than one shift axis (NMF2D / NMF3D): koff != nullptr.  The tables are the 1-D tables of every last-axis
  // line of H zero-padded by T_d - 1 lines on both sides of every outer axis (nmfmu_convnd_tables), so the chunk index is
  // still (a term of the tile row) + (a term of the k-chunk); the latter comes precomputed, one int per k-chunk
  // (INT_MIN: padding of k), the former from win_lh / win_t / win_l below.  tT / tLh hold the products over the axes.
  const int* koff;
  // ragged channels inside the grid (EPI_RATIO with an implicit Hu operand): channels [rag_c0, rag_C), at most 16, are
  // not part of the GEMM's own tiles (one or two rows would cost a whole tile row, i.e. a second scheduling round);
  // every workgroup carries one extra 16 x 16 MFMA block for them (see nt_gemm_kernel).  rag_C <= rag_c0: off.
  int rag_c0, rag_C;
  // window operand (OPS == kOpsAWin): A[(b,j)][(t, c)] = P[(b, j + t)][c] -- the rows of A are shifted rows of the
  // row-major plane(s) a_hi / a_lo (the ratio planes of the H half-step, [(b,l)][c]); nothing is unfolded.  Shift axes
  // outermost first, missing leading axes have extent 1.
  int win_lh[3], win_t[3], win_l[3];   // H extent, taps, V extent (= lh + t - 1) per shift axis
  int win_rows;                        // B * prod(lh): rows of A that exist
  int win_ck;                          // 64-channel k-tiles per tap (k = (t * win_ck + ck) * 64 + c')
  int win_tstep;                       // rows of P per last-axis tap step (1; F when F taps are folded into N, see nmfmu.h)
  unsigned win_pitch;                  // bytes per row of P
};

// which operand is fetched from a window table of H instead of from planes (nmfmu.h: NMFMU_OPS_*)
enum GemmOps : int { kOpsPlanes = 0, kOpsBHu = 1, kOpsBHuT = 2, kOpsAHu = 3, kOpsAWin = 4 };

// Workgroup tile shape.  WM x WN waves, each MI x NI MFMA 32x32 blocks: 128 x 128, 256 threads, two workgroups per CU.
// (A 256 x 256 / eight-wave instance of this template was built and parity-tested in round 2: half the operand bytes
// per MFMA, 52 instead of 27 % MFMA-busy per CU, but at configs[3] it pads 1025 channels to 1280 and leaves 160 / 416
// workgroups for 256 CUs -- 2 210 instead of 2 650 it/s without a stream-K scheduler; removed in round 3, see the git
// history of round 2.)
template <int WM_, int WN_, int MI_, int NI_>
struct GemmShape {
  static constexpr int WM = WM_, WN = WN_, MI = MI_, NI = NI_;
  static constexpr int BM = WM * MI * 32, BN = WN * NI * 32, THREADS = 64 * WM * WN;
};
using GemmSmall = GemmShape<2, 2, 2, 2>;
// narrow-N tiles (128 x 32, 128 x 64) for the window-operand GEMM, whose N is the rank
using GemmN32 = GemmShape<4, 1, 1, 1>;
using GemmN64 = GemmShape<4, 1, 1, 2>;
// 64 x 128: the explicit operand of a reconstruction / W-numerator GEMM when the model has at most 64 channels (with
// GemmN64 for the transposed problem): half the MFMA work and explicit-operand traffic of a half-empty 128-row tile
using GemmM64 = GemmShape<1, 4, 2, 1>;
// (256 x 128 and 128 x 256 with eight 64 x 64 waves were measured too: same time as two 128 x 128 workgroups per CU.)

// WS ("window staging", round 5): the implicit Toeplitz operand of a k-tile is not staged chunk-major (8 k-chunks x 128 rows
// x 16 B = 16 KiB of table entries, most of them the SAME entries again: chunk q of row m is table entry base + m - 8 q) but
// as the window of DISTINCT entries the tile touches: 128 + 56 consecutive entries = 2.9 KiB.  Slot s of a window region
// holds table entry (region base) + s; the fragment of (row m, chunk q) is slot m + 56 - 8 q (rows (b,l), reversed windows)
// resp. (n - 1 - m) + 8 q (rows (r,t), forward windows, n rows in the segment).  A tile that crosses a boundary of the
// table's lines -- the next r inside a k-tile (rows (b,l)), the next r inside the tile's rows (rows (r,t)) -- uses a second
// region for the part behind the boundary.  Two regions of WIN_REG = 192 slots: 6 KiB per plane and stage instead of 16,
// 3 (or 6) LDS-DMA pieces per k-tile instead of 16, issued by waves 0-2.  The k loop of this kernel is bound by LDS-DMA issue
// (DESIGN.md section 3.4): 19 instead of 32 pieces per k-tile.  Requires tiles without padding rows / dead k-chunks in the
// implicit operand (nmfmu_gemm checks the shape; everything else keeps the chunk-major path, bit-identical results).
template <bool X3, class SH, bool WS = false, int TOP = 1>
struct GemmCfg {
  static constexpr int BM = SH::BM, BN = SH::BN, BK = 64;
  static constexpr int A_TILE = BM * BK * 2, B_TILE = BN * BK * 2;   // bytes of one explicit operand-plane tile
  static constexpr int NPL = X3 ? 2 : 1;
  static constexpr int WIN_REG = 192, WIN_BYTES = 2 * WIN_REG * 16;   // window regions of an implicit operand (WS)
  static constexpr int A_BYTES = (WS && TOP == 0) ? WIN_BYTES : A_TILE;   // what a stage holds per operand plane
  static constexpr int B_BYTES = (WS && TOP == 1) ? WIN_BYTES : B_TILE;
  static constexpr int STAGE = NPL * (A_BYTES + B_BYTES);             // A planes then B planes
  static constexpr int LDS_BYTES = 2 * STAGE;
  // ragged channels: 16 rows of the explicit operand per k-tile, one 2 KiB tile per plane and stage behind the stages
  static constexpr int RAG_TILE = 16 * BK * 2, RAG_BYTES = 2 * NPL * RAG_TILE;
  static constexpr int PA = BM * 8 / SH::THREADS, PB = BN * 8 / SH::THREADS;   // DMA passes per plane tile
  static_assert(PA * SH::THREADS == BM * 8 && PB * SH::THREADS == BN * 8, "whole DMA passes");
  static_assert(PA <= 4 && PB <= 4, "k-position registers of the implicit operand");
  static_assert(!WS || (BM == 128 && BN == 128 && SH::THREADS == 256), "window staging: 128 x 128 tiles, four waves");
};

template <bool X3, int EPI, int BETA, int OPS, class SH, int OPT, bool ND, bool WS = false>
__global__ void __launch_bounds__(SH::THREADS, ((X3 || SH::THREADS > 256) ? 1 : 2)) nt_gemm_kernel(const GemmArgs a) {
  using C = GemmCfg<X3, SH, WS, (OPS == kOpsAHu ? 0 : 1)>;
  static_assert(!(X3 && OPT == kOpF16), "fp16 operands are single-plane");
  static_assert(!WS || (!ND && (OPS == kOpsBHu || OPS == kOpsBHuT || OPS == kOpsAHu)), "window staging: one shift axis, an implicit operand");
  // fp16: the ratio planes are converted with saturation (a ratio above 65504 becomes 65504, not inf)
  if constexpr (OPT == kOpF16) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  constexpr int MI = SH::MI, NI = SH::NI, THREADS = SH::THREADS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hl = lane >> 5;
  const int wm = wave / SH::WN, wn = wave % SH::WN;
  int bm = blockIdx.y, zz = blockIdx.z, nsp = a.k_split;
  const int bn = blockIdx.x;
  if constexpr (EPI == kEpiFold) {
    // tail-round split: tile rows >= m_tiles - tail_rows appear k_split times along grid y (the hardware dispatches them
    // last), each instance running one part of the contraction; their diagonal sums are linear, the gather adds them
    nsp = 1;
    const int row0 = a.m_pad / C::BM - a.tail_rows;
    if (a.tail_rows > 0 && bm >= row0) {
      const int q = bm - row0;
      zz = q % a.k_split, bm = row0 + q / a.k_split, nsp = a.k_split;
    }
  }
  const int ktiles_all = a.k_len / C::BK;
  // split-K: part zz runs k-tiles [kt0, kt0 + ktiles) and stores its partial into slab zz
  const int kt_per = (ktiles_all + nsp - 1) / nsp;
  const int kt0 = zz * kt_per;
  const int ktiles = max(0, min(kt_per, ktiles_all - kt0));
  const size_t ldk = (size_t)a.k_pad * 2;  // bytes per operand row
  // Ragged channels (round 3): the reconstruction GEMMs cover whole 128-channel tiles; the 1 .. 16 channels beyond them
  // (the 1025th bin of a spectrogram) ride along as ONE extra 16 x 16 x 32 MFMA block per workgroup: the 128 frames of a
  // tile column (resp. tile row) are shared out 16 apiece over eight workgroups that hold that implicit-operand tile in
  // LDS anyway; 16 rows of the explicit operand (W planes, rows rag_c0 ..) come with every stage as a 2 KiB tile of their
  // own.  Wave 0 issues 2 (6 with split operands) small MFMAs and 4 (8) fragment reads per k-tile and runs the same
  // elementwise epilogue on its 16 x 16 block.  (A 16-frame direct-summation slice per workgroup after the tile was
  // measured first: +8.4 us per GEMM -- as much as the separate nmfmu_conv_ragged_rows launch it replaced.)
  constexpr bool RAGK = EPI == kEpiRatio && (OPS == kOpsBHu || OPS == kOpsAHu) && SH::THREADS == 256 && C::BM == 128 && C::BN == 128;
  bool rag_on = false;
  int rag_sub0 = 0;
  const char* rag_src[C::NPL] = {};
  if constexpr (RAGK) {
    const int sub = OPS == kOpsBHu ? bm : bn;      // which 16 of the tile's 128 frames this workgroup takes
    rag_on = a.rag_C > a.rag_c0 && sub < 8;
    rag_sub0 = 16 * sub;
    rag_src[0] = reinterpret_cast<const char*>(OPS == kOpsBHu ? a.a_hi : a.b_hi) + (size_t)a.rag_c0 * ldk;
    if constexpr (X3) rag_src[1] = reinterpret_cast<const char*>(OPS == kOpsBHu ? a.a_lo : a.b_lo) + (size_t)a.rag_c0 * ldk;
  }

  // DMA source pointers: thread handles chunk c = p*THREADS + tid of a tile: row = c >> 3, LDS slot = c & 7,
  // source slot = slot ^ ((row >> 1) & 7)
  const char* src[2 * C::NPL];
  {
    const char* bases[4] = {reinterpret_cast<const char*>(a.a_hi), reinterpret_cast<const char*>(a.a_lo),
                            reinterpret_cast<const char*>(a.b_hi), reinterpret_cast<const char*>(a.b_lo)};
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
      for (int pl = 0; pl < C::NPL; ++pl)
        src[op * C::NPL + pl] = bases[op * 2 + pl] + ((OPS == kOpsAWin && op == 0) ? (size_t)0 : (size_t)(op == 0 ? bm * C::BM : bn * C::BN) * ldk);
  }
  constexpr bool kToep = OPS == kOpsBHu || OPS == kOpsBHuT || OPS == kOpsAHu;   // an operand fetched from window tables
  const char* tab[2] = {nullptr, nullptr};   // window table planes of the implicit operand
  if constexpr (kToep) {
    tab[0] = reinterpret_cast<const char*>(OPS == kOpsAHu ? a.a_hi : a.b_hi);
    tab[1] = reinterpret_cast<const char*>(OPS == kOpsAHu ? a.a_lo : a.b_lo);
  }
  const int row_t = tid >> 3;                                   // + (THREADS / 8) * p
  const int sslot = (tid & 7) ^ ((row_t >> 1) & 7);             // (row >> 1) & 7 is the same for every pass (16 | THREADS/8)
  const size_t thr_off = (size_t)row_t * ldk + sslot * 16;

  // ---- implicit Toeplitz operand (see nmfmu_conv_tables for the table layout).
  // Its LDS tile is CHUNK-MAJOR, [8 k-chunks][ROWS rows] x 16 B, and the DMA lanes run along the rows: chunk
  // c = p*THREADS + tid  ->  k-chunk c / ROWS, row c % ROWS.  Consecutive rows of one k-chunk are consecutive table
  // entries, so a wave instruction reads one contiguous KiB (lanes along k instead touch four cache lines per lane quad
  // and measured 30 % slower), and the fragment reads (lane j = row) are contiguous too: no swizzle needed.
  constexpr int TOP = OPS == kOpsAHu ? 0 : 1;                 // which operand is implicit
  constexpr int TROWS = TOP == 0 ? C::BM : C::BN;             // rows of the implicit operand's tile
  constexpr int TP = TOP == 0 ? C::PA : C::PB;
  constexpr int KPP = THREADS / TROWS;                        // k-chunks of the implicit operand per DMA pass
  static_assert(!kToep || (KPP * TROWS == THREADS && KPP * TP == 8), "whole k-chunks per DMA pass");
  constexpr bool kHuRows = OPS == kOpsBHu || OPS == kOpsAHu;  // rows (b,l), k = (r,t); else rows (r,t), k = (b,l)
  int trow = -1;      // chunk-index contribution of this thread's row (the same in all passes), -1 = padding row
  int tL = 0, tJJ = 0, tT8 = 0;
  // k position of the thread's chunk in pass p (k-chunk KPP p + tid / TROWS of the k-tile), advanced by one k-tile per
  // stage_issue call, which come strictly in k order (no integer division in the loop):
  //   rows-(b,l) operand: k = r T + 8 tc  -> (kq, kr) = (r, tc);   rows-(r,t) operand: k = b L + l0 -> (kq, kr) = (b, l0)
  int kq[4] = {0, 0, 0, 0}, kr[4] = {0, 0, 0, 0};
  int nd_soff[4] = {0, 0, 0, 0};      // ND: koff of the chunks the next stage_issue call fetches
  if constexpr (kToep) {
    tL = a.tLh + a.tT - 1, tJJ = a.tLh + 2 * a.tT - 2, tT8 = a.tT / 8;
    const int row = (TOP == 0 ? bm : bn) * TROWS + (tid % TROWS);
    if constexpr (ND) {        // several shift axes: the same two forms with one term per axis
      const int jj2 = a.win_l[2] + a.win_t[2] - 1, jj1 = a.win_l[1] + a.win_t[1] - 1, jj0 = a.win_l[0] + a.win_t[0] - 1;
      const int jjt = jj0 * jj1 * jj2;
      if constexpr (kHuRows) {
        const int l_tot = a.win_l[0] * a.win_l[1] * a.win_l[2];
        const int b = row / l_tot, lf = row - b * l_tot;
        const int l2 = lf % a.win_l[2], l01 = lf / a.win_l[2], l1 = l01 % a.win_l[1], l0 = l01 / a.win_l[1];
        trow = b < a.tB ? b * a.tR * jjt + ((l0 + a.win_t[0] - 1) * jj1 + l1 + a.win_t[1] - 1) * jj2 + l2 + a.win_t[2] - 1 : -1;
      } else {
        const int r = row / a.tT, tf = row - r * a.tT;
        const int t2 = tf % a.win_t[2], t01 = tf / a.win_t[2], t1 = t01 % a.win_t[1], t0 = t01 / a.win_t[1];
        trow = r < a.tR ? r * jjt + ((a.win_t[0] - 1 - t0) * jj1 + a.win_t[1] - 1 - t1) * jj2 + a.win_t[2] - 1 - t2 : -1;
      }
    } else if constexpr (kHuRows) {   // row = b L + l  ->  (b R) JJ + l + (T - 1)
      const int b = row / tL, l = row - b * tL;
      trow = b < a.tB ? b * a.tR * tJJ + l + a.tT - 1 : -1;
    } else {                   // row = r T + t  ->  r JJ - t + (T - 1)
      const int r = row / a.tT, t = row - r * a.tT;
      trow = r < a.tR ? r * tJJ - t + a.tT - 1 : -1;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // (wave-uniform: a wave's 64 lanes are 64 consecutive rows of ONE k-chunk -- kept in scalar registers, so the
      // per-k-tile position arithmetic below costs no vector instructions)
      const int kc = kt0 * 8 + KPP * p + __builtin_amdgcn_readfirstlane(tid / TROWS);     // 8 chunks per k-tile
      if constexpr (ND) kq[p] = kc, nd_soff[p] = a.koff[kc];     // the k-chunk itself: index into koff
      else if constexpr (kHuRows) kq[p] = kc / tT8, kr[p] = kc - kq[p] * tT8;
      else kq[p] = (kc * 8) / tL, kr[p] = kc * 8 - kq[p] * tL;
    }
  }
  auto toep_index = [&](int p) __attribute__((always_inline)) -> int {   // table chunk of (row, k-chunk p); 0 = the all-zero chunk
    // scalar part first (k position), then one vector add / select for the lane's row
    int soff;
    bool live;
    if constexpr (ND) soff = nd_soff[p], live = soff != INT_MIN;
    else if constexpr (kHuRows) soff = 1 + kq[p] * tJJ - 8 * kr[p], live = kq[p] < a.tR;
    else soff = 1 + kq[p] * a.tR * tJJ + kr[p], live = kq[p] < a.tB;
    return (live && trow >= 0) ? trow + soff : 0;
  };
  auto toep_advance = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if constexpr (ND) {
        kq[p] += 8;
        nd_soff[p] = a.koff[kq[p]];     // for the NEXT k-tile: the scalar load has a whole k-tile to land (koff carries 8 spare ints)
      } else if constexpr (kHuRows) {
        kr[p] += 8;
        while (kr[p] >= tT8) kr[p] -= tT8, ++kq[p];
      } else {
        kr[p] += 64;
        while (kr[p] >= tL) kr[p] -= tL, ++kq[p];
      }
    }
  };
  // ---- window staging (WS): per-thread constants.  A region holds WIN_REG slots, slot u <- table entry ws_tsl[region] + u +
  // (the k-tile's scalar offset); waves 0-2 carry the 192 slots of a region (one 1-KiB piece each).
  //   rows (b,l), k = (r,t)  [B_HU / A_HU; one b per tile: L % 128 == 0]:  entry(row m, chunk q) = trow(m) + soff(q), soff falling
  //     by 8 per chunk -> slot m + 56 - 8 q.  Region 1 = the chunks behind an r boundary inside the k-tile (T >= 64: one at most).
  //   rows (r,t), k = (b,l)  [B_HUT; one b per k-tile: L % 64 == 0]:  entries fall with t, rise by 8 per chunk -> slot
  //     (n - 1 - m') + 8 q inside the row segment (n rows of one r).  Region 1 = the rows of the tile's second r (T >= 128: two at most).
  int ws_tsl[2] = {0, 0};
  int ws_q0 = 0, ws_r0 = 0;        // k position of chunk 0 of the next k-tile to issue (the meaning of kq / kr above)
  int ws_qs0 = 8, ws_qs1 = 8;      // rows (b,l): first chunk behind the r boundary of the k-tile in staging buffer 0 / 1 (8: none)
  bool ws_two = false;             // rows (r,t): the tile's rows span two r
  int ws_nent = 1;
  constexpr int WI = TOP == 0 ? MI : NI;
  int ws_base[WI] = {};            // byte offset of this lane's fragment rows inside the window block (lane part of the address)
  if constexpr (WS) {
    static_assert(TROWS == 128, "window staging: 128 implicit rows per tile");
    ws_nent = 1 + a.tB * a.tR * tJJ;
    const int row0 = (TOP == 0 ? bm : bn) * TROWS;
    const int u = tid;             // (tid < 192 issue)
    if constexpr (kHuRows) {
      const int b = row0 / tL, l0 = row0 - b * tL;
      ws_tsl[0] = ws_tsl[1] = b * a.tR * tJJ + l0 + a.tT - 1 + (u - 56);
      const int kc = kt0 * 8;
      ws_q0 = kc / tT8, ws_r0 = kc - ws_q0 * tT8;
#pragma unroll
      for (int i = 0; i < WI; ++i) ws_base[i] = ((TOP == 0 ? wm * MI : wn * NI) * 32 + i * 32 + j) * 16 + 128 - 128 * hl;
    } else {
      const int r0 = row0 / a.tT, t0 = row0 - r0 * a.tT;
      const int n0 = min(128, a.tT - t0), n1 = 128 - n0;
      ws_two = n1 > 0;
      ws_tsl[0] = r0 * tJJ - t0 - n0 + a.tT + u;
      ws_tsl[1] = (r0 + 1) * tJJ - n1 + a.tT + u;
      const int k0 = kt0 * 64;
      ws_q0 = k0 / tL, ws_r0 = k0 - ws_q0 * tL;
#pragma unroll
      for (int i = 0; i < WI; ++i) {
        const int row = (TOP == 0 ? wm * MI : wn * NI) * 32 + i * 32 + j;
        const bool seg = row >= n0;
        const int rl = seg ? row - n0 : row, ng = seg ? n1 : n0;
        ws_base[i] = ((ng - 1 - rl) + (seg ? C::WIN_REG : 0) + 8 * hl) * 16;
      }
    }
  }
  // LDS offset of plane pl of operand op inside a stage
  auto tile_off = [](int op, int pl) { return op == 0 ? pl * C::A_BYTES : C::NPL * C::A_BYTES + pl * C::B_BYTES; };

  // LDS-DMA by inline asm in the scalar-base form (global_load_lds_dwordx4 v_off, s[base:base+1]): the k-tile advance of
  // an explicit operand is a scalar add on its base, the pass offsets are four precomputed registers, the implicit
  // operand's offset is (row + scalar k position) * 16 -- a handful of vector instructions per k-tile instead of a
  // 64-bit address per pass.  hipcc does not count these loads: the waits before the barriers are explicit.
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned voff_exp[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) voff_exp[p] = (unsigned)(thr_off + (size_t)p * (THREADS / 8) * ldk);
  // ---- window operand (kOpsAWin): tile row m = (b, j) reads plane row (b, j + t); the row map is per thread and pass,
  // the tap offset of the k-tile (t, ck) is scalar and advances with carries (k-tiles come strictly in order)
  unsigned voff_win[4] = {0, 0, 0, 0};
  int wk_ck = 0, wk_t1 = 0, wk_t2 = 0, wk_off = 0;
  if constexpr (OPS == kOpsAWin) {
    const int lh_tot = a.win_lh[0] * a.win_lh[1] * a.win_lh[2], l_tot = a.win_l[0] * a.win_l[1] * a.win_l[2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = min(bm * C::BM + row_t + p * (THREADS / 8), a.win_rows - 1);
      const int b = m / lh_tot, jf = m - b * lh_tot;
      const int j2 = jf % a.win_lh[2], j01 = jf / a.win_lh[2], j1 = j01 % a.win_lh[1], j0 = j01 / a.win_lh[1];
      voff_win[p] = (unsigned)(b * l_tot + (j0 * a.win_l[1] + j1) * a.win_l[2] + j2) * a.win_pitch + sslot * 16;
    }
    const int tf = kt0 / a.win_ck;
    wk_ck = kt0 - tf * a.win_ck;
    wk_t2 = tf % a.win_t[2];
    const int t01 = tf / a.win_t[2];
    wk_t1 = t01 % a.win_t[1];
    wk_off = ((t01 / a.win_t[1]) * a.win_l[1] + wk_t1) * a.win_l[2] + wk_t2 * a.win_tstep;
  }
  auto win_advance = [&]() __attribute__((always_inline)) {
    if (++wk_ck == a.win_ck) {
      wk_ck = 0, wk_off += a.win_tstep;
      if (++wk_t2 == a.win_t[2]) {
        wk_t2 = 0, wk_off += a.win_l[2] - a.win_t[2] * a.win_tstep;
        if (++wk_t1 == a.win_t[1]) wk_t1 = 0, wk_off += (a.win_l[1] - a.win_t[1]) * a.win_l[2];
      }
    }
  };
  auto dma1k = [&](const char* sbase, unsigned voff, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory", "m0");
  };
  auto stage_issue = [&](int kt, int buf) __attribute__((always_inline)) {
    // window staging: scalar table offsets of this k-tile's one or two regions
    int ws_soff[2] = {0, 0};
    bool ws_need1 = false;
    if constexpr (WS) {
      if constexpr (kHuRows) {
        const int qs = tT8 - ws_r0;                       // chunks left in the current r
        ws_need1 = qs < 8;
        ws_soff[0] = 1 + ws_q0 * tJJ - 8 * ws_r0;
        ws_soff[1] = 1 + (ws_q0 + 1) * tJJ + 8 * qs;      // chunk q >= qs: entry = trow + 1 + (r + 1) JJ - 8 (q - qs)
        if (buf) ws_qs1 = ws_need1 ? qs : 8;
        else ws_qs0 = ws_need1 ? qs : 8;
      } else {
        ws_soff[0] = ws_soff[1] = 1 + ws_q0 * a.tR * tJJ + ws_r0;
        ws_need1 = ws_two;
      }
    }
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
      for (int pl = 0; pl < C::NPL; ++pl) {
        const bool implicit = kToep && op == TOP;
        if constexpr (WS) {
          if (implicit) {
            // (entries outside the table can only sit in slots no fragment reads -- the unused head / tail of a region --
            // but the address must stay inside the allocation: clamp)
            if (wave < 3) {
              const unsigned dst = lds_base + buf * C::STAGE + tile_off(op, pl) + wave * 1024;
              dma1k(tab[pl], (unsigned)min(max(ws_tsl[0] + ws_soff[0], 0), ws_nent - 1) * 16u, dst);
              if (ws_need1) dma1k(tab[pl], (unsigned)min(max(ws_tsl[1] + ws_soff[1], 0), ws_nent - 1) * 16u, dst + C::WIN_REG * 16);
            }
            continue;
          }
        }
        const char* s0 = (OPS == kOpsAWin && op == 0)
                             ? src[op * C::NPL + pl] + (size_t)wk_off * a.win_pitch + wk_ck * (C::BK * 2)
                             : src[op * C::NPL + pl] + (size_t)(kt0 + kt) * (C::BK * 2);     // scalar
        const int passes = op == 0 ? C::PA : C::PB;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (p < passes) {
            const unsigned dst = lds_base + buf * C::STAGE + tile_off(op, pl) + p * (THREADS * 16) + wave * 1024;
            if (implicit) dma1k(tab[pl], (unsigned)toep_index(p) * 16u, dst);
            else dma1k(s0, (OPS == kOpsAWin && op == 0) ? voff_win[p] : voff_exp[p], dst);
          }
        }
      }
    if constexpr (RAGK) {
      // rows rag_c0 .. rag_c0 + 15 of the explicit operand: 128 chunks = one more pass of waves 0 and 1 (same source
      // swizzle as pass 0 of the tiles: row = tid >> 3)
      if (rag_on && wave < 2) {
#pragma unroll
        for (int pl = 0; pl < C::NPL; ++pl)
          dma1k(rag_src[pl] + (size_t)(kt0 + kt) * (C::BK * 2), voff_exp[0],
                lds_base + 2 * C::STAGE + (buf * C::NPL + pl) * C::RAG_TILE + wave * 1024);
      }
    }
    if constexpr (WS) {
      if constexpr (kHuRows) {
        ws_r0 += 8;
        if (ws_r0 >= tT8) ws_r0 -= tT8, ++ws_q0;
      } else {
        ws_r0 += 64;
        if (ws_r0 >= tL) ws_r0 -= tL, ++ws_q0;
      }
    } else if constexpr (kToep) {
      toep_advance();
    }
    if constexpr (OPS == kOpsAWin) win_advance();
  };
  (void)TP; (void)KPP;
  f32x4 racc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (RAGK && BETA != kEuc) racc = f32x4{kEps, kEps, kEps, kEps};

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = ((EPI == kEpiRatio || EPI == kEpiLoss) && BETA != kEuc) ? kEps : 0.f;

  const int a_rowoff = (wm * MI * 32 + j) * 128;  // + mi * 4096
  const int b_rowoff = (wn * NI * 32 + j) * 128;
  const int swz = ((j >> 1) & 7) << 4;

  if (ktiles > 0) stage_issue(0, 0);   // (an empty contraction part -- split factor not dividing the k-tiles -- stores zeros)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // one k-tile; the staging buffer index is a compile-time constant (the loop below is unrolled by two), so every LDS
  // address of the fragment reads and of the DMA destinations is a register base plus an immediate
  auto k_body = [&](int kt, auto bufc, auto strc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    constexpr bool STR = decltype(strc)::value;   // WS, rows (b,l): this k-tile crosses an r boundary (second window region)
    const int qs_cur = buf ? ws_qs1 : ws_qs0;
#ifdef NMFMU_GEMM_ABL_NODMA   // timing-only ablation (wrong results): no LDS-DMA in the loop, every k-tile multiplies stale LDS
    (void)0;
#else
    if (kt + 1 < ktiles) stage_issue(kt + 1, buf ^ 1);
#endif
    const char* sb = smem + buf * C::STAGE;
    // operand fragments are fetched one 16-wide k-step ahead of the MFMAs that consume them (pinned below)
    u32x4 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];
#ifdef NMFMU_GEMM_ABL_NOFRAG   // timing-only ablation (wrong results): no fragment reads, the MFMAs take whatever the registers hold
    auto ldf = [&](const char*) __attribute__((always_inline)) { u32x4 r; asm volatile("" : "=v"(r)); return r; };
#else
    auto ldf = [&](const char* p_) __attribute__((always_inline)) { return ld16(p_); };
#endif
    auto load_frags = [&](int ks, int fb) __attribute__((always_inline)) {
      const int so = ((2 * ks + hl) << 4) ^ swz;          // row-major tile: 128-byte rows, XOR-swizzled 16-byte slots
      // window staging: lane base + an immediate per k-step (two chunks = 256 bytes of slots), + the second region for the
      // chunks behind the r boundary of a straddling k-tile
      int wo = 0;
      if constexpr (WS) {
        wo = kHuRows ? (3 - ks) * 256 : ks * 256;
        if constexpr (STR) wo += (2 * ks + hl >= qs_cur) ? C::WIN_REG * 16 : 0;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        int ao;
        if constexpr (WS && TOP == 0) ao = ws_base[i] + wo;
        else ao = (OPS == kOpsAHu) ? (2 * ks + hl) * (C::BM * 16) + (wm * MI * 32 + i * 32 + j) * 16 : a_rowoff + i * 4096 + so;
        ah[fb][i] = ldf(sb + ao);
        if constexpr (X3) al[fb][i] = ldf(sb + C::A_BYTES + ao);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int bo;
        if constexpr (WS && TOP == 1) bo = ws_base[i] + wo;
        else bo = (OPS == kOpsBHu || OPS == kOpsBHuT) ? (2 * ks + hl) * (C::BN * 16) + (wn * NI * 32 + i * 32 + j) * 16
                                                      : b_rowoff + i * 4096 + so;
        bh[fb][i] = ldf(sb + C::NPL * C::A_BYTES + bo);
        if constexpr (X3) bl[fb][i] = ldf(sb + C::NPL * C::A_BYTES + C::B_BYTES + bo);
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int fb = ks & 1;
      if (ks + 1 < 4) load_frags(ks + 1, fb ^ 1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if constexpr (X3) {
            acc[mi][ni] = mfma_bf16(al[fb][mi], bh[fb][ni], acc[mi][ni]);
            acc[mi][ni] = mfma_bf16(ah[fb][mi], bl[fb][ni], acc[mi][ni]);
          }
#ifdef NMFMU_GEMM_ABL_NOMFMA  // timing-only ablation: fragment reads kept alive, no matrix instructions
          asm volatile("" ::"v"(ah[fb][mi]), "v"(bh[fb][ni]));
#else
          acc[mi][ni] = mfma_op<OPT>(ah[fb][mi], bh[fb][ni], acc[mi][ni]);
#endif
        }
    }
    {
      // pin the order: the first k-step's fragment reads, then per MFMA (group) its share of the next k-step's reads
      constexpr int RD = (MI + NI) * C::NPL, NM = MI * NI, MF = X3 ? 3 : 1;
      __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
      static_for<4>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        static_for<NM>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
          constexpr int nrd = (RD * (q + 1)) / NM - (RD * q) / NM;
          if constexpr (ks + 1 < 4 && nrd > 0) __builtin_amdgcn_sched_group_barrier(0x100, nrd, 0);
        });
      });
    }
    if constexpr (RAGK) {
      if (rag_on && wave == 0) {
        // 16 x 16 x 32 fragments: lane = (row r16, k-chunk g4 of the 32-wide step); the ragged tile is row-major with the
        // tiles' XOR swizzle, the implicit operand's tile is chunk-major (no swizzle)
        const int r16 = lane & 15, g4 = lane >> 4;
        const char* rt = smem + 2 * C::STAGE + buf * C::NPL * C::RAG_TILE;
        const char* it = sb + (TOP == 0 ? 0 : C::NPL * C::A_BYTES);
        constexpr int IT_PLANE = TOP == 0 ? C::A_BYTES : C::B_BYTES;
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
          const int q = 4 * ks2 + g4;
          const int eo = r16 * 128 + ((q ^ ((r16 >> 1) & 7)) << 4);
          int io = (q * TROWS + rag_sub0 + r16) * 16;
          if constexpr (WS) io = (rag_sub0 + r16 + 56 - 8 * q) * 16 + ((STR && q >= qs_cur) ? C::WIN_REG * 16 : 0);
          const u32x4 eh = ld16(rt + eo), ih = ld16(it + io);
          // D[m][n] = sum_k A[m][k] B[n][k]: B_HU -> the ragged rows are rows of A; A_HU -> rows of B
          if constexpr (X3) {
            const u32x4 el = ld16(rt + C::RAG_TILE + eo), il = ld16(it + IT_PLANE + io);
            racc = OPS == kOpsBHu ? mfma16_bf16(el, ih, racc) : mfma16_bf16(il, eh, racc);
            racc = OPS == kOpsBHu ? mfma16_bf16(eh, il, racc) : mfma16_bf16(ih, el, racc);
          }
          racc = OPS == kOpsBHu ? mfma16_op<OPT>(eh, ih, racc) : mfma16_op<OPT>(ih, eh, racc);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage has landed (this wave's share of it)
#ifndef NMFMU_GEMM_ABL_NOBAR   // (timing-only ablation: no barrier in the loop)
    __syncthreads();
#endif
  };
  auto k_tile = [&](int kt, auto bufc) __attribute__((always_inline)) {
    if constexpr (WS && kHuRows) {
      if (__builtin_amdgcn_readfirstlane(decltype(bufc)::value ? ws_qs1 : ws_qs0) < 8) {
        k_body(kt, bufc, std::true_type{});
        return;
      }
    }
    k_body(kt, bufc, std::false_type{});
  };
  {
    int kt = 0;
    for (; kt + 2 <= ktiles; kt += 2) {
      k_tile(kt, std::integral_constant<int, 0>{});
      k_tile(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < ktiles) k_tile(kt, std::integral_constant<int, 0>{});
  }

  // ---------------- epilogue: accumulator e of lane (j, hl): row (e&3) + 8*(e>>2) + 4*hl, column j
  if constexpr (EPI == kEpiFold) {
    // rows m = r T + t, columns n = b L + l; neg[b][r][j] = sum_t Y[(r,t)][(b, j+t)] runs along the diagonals
    // n - m = j + (b L - r T).  With T, L >= 128 a 128 x 128 quadrant of the tile holds at most two r and two b: four
    // (r, b) segments, 255 diagonals each.  A quadrant goes through LDS (the staging buffers, every wave is past the last
    // barrier; rows padded to 129 floats: the lanes of a wave walk their diagonals from different rows, and with a
    // 128-float pitch the short diagonals of one wave would all sit in one bank), thread dd sums diagonal
    // dd = nl - ml + 127 top to bottom, split by segment.  THREADS / 256 quadrants are in flight at a time.  Entries whose
    // j falls outside [0, Lh) land on diagonals the gather never reads; padding rows / columns are exact zeros.
    constexpr int QM = C::BM / 128, QN = C::BN / 128, QP = THREADS / 256, QBUF = 128 * kFoldLd;
    static_assert((QM * QN) % QP == 0, "whole quadrant groups");
    float* tl_all = reinterpret_cast<float*>(smem);
    const int L = a.tLh + a.tT - 1;
    const int tiles_n = a.n_pad / 128;
    for (int g = 0; g < QM * QN; g += QP) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int r0 = wm * MI * 32 + mi * 32, c0 = wn * NI * 32 + ni * 32;   // block origin inside the tile
          const int q = (r0 >> 7) * QN + (c0 >> 7);
          if (q >= g && q < g + QP) {
            float* tl = tl_all + (q - g) * QBUF;
#pragma unroll
            for (int e = 0; e < 16; ++e)
              tl[((r0 & 127) + (e & 3) + 8 * (e >> 2) + 4 * hl) * kFoldLd + (c0 & 127) + j] = acc[mi][ni][e];
          }
        }
      __syncthreads();
      {
        const int q = g + (tid >> 8), qm = q / QN, qn = q - qm * QN;
        const int tm = bm * QM + qm, tn = bn * QN + qn;              // 128 x 128 tile coordinates in Y
        const int rb = (tm * 128 / a.tT + 1) * a.tT - tm * 128;      // first quadrant row of the second r (>= 128: none)
        const int nb = (tn * 128 / L + 1) * L - tn * 128;            // first quadrant column of the second b
        const int dd = tid & 255;
        if (dd < 255) {
          // every (second r?, second b?) segment of a diagonal is one contiguous run of rows: the r switch is at row rb,
          // the b switch at row nb - (dd - 127); straight sums over up to four runs, four independent partial sums each
          const int lo = max(0, 127 - dd), hi1 = min(127, 254 - dd) + 1;
          const int rsw = min(max(rb, lo), hi1), bsw = min(max(nb - dd + 127, lo), hi1);
          const float* pd = tl_all + (tid >> 8) * QBUF + dd - 127;
          auto run = [&](int r0, int r1) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            int ml = r0;
            for (; ml + 16 <= r1; ml += 16) {   // sixteen LDS reads in flight
              float v[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) v[u] = pd[(ml + u) * (kFoldLd + 1)];
#pragma unroll
              for (int u = 0; u < 16; u += 4) p0 += v[u], p1 += v[u + 1], p2 += v[u + 2], p3 += v[u + 3];
            }
            for (; ml + 4 <= r1; ml += 4) {
              p0 += pd[ml * (kFoldLd + 1)], p1 += pd[(ml + 1) * (kFoldLd + 1)], p2 += pd[(ml + 2) * (kFoldLd + 1)],
                  p3 += pd[(ml + 3) * (kFoldLd + 1)];
            }
            for (; ml < r1; ++ml) p0 += pd[ml * (kFoldLd + 1)];
            return (p0 + p1) + (p2 + p3);
          };
          const float s0 = run(lo, min(rsw, bsw));        // first r, first b
          const float s1 = run(max(lo, bsw), rsw);        // first r, second b
          const float s2 = run(rsw, max(rsw, bsw));       // second r, first b   (rows [rsw, bsw))
          const float s3 = run(max(rsw, bsw), hi1);       // second r, second b
          float* po = a.out + (size_t)zz * ((size_t)(a.m_pad / 128) * tiles_n * 1024) + ((size_t)(tm * tiles_n + tn) * 4) * 256 + dd;
          po[0] = s0, po[256] = s1, po[512] = s2, po[768] = s3;
        }
      }
      if (g + QP < QM * QN) __syncthreads();
    }
    return;
  }
  float lacc = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = bn * C::BN + wn * NI * 32 + ni * 32 + j;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm * C::BM + wm * MI * 32 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
        const size_t idx = (size_t)m * a.ldn + n;
        const float s = acc[mi][ni][e];
        if constexpr (EPI == kEpiF32) {
          a.out[(size_t)zz * a.m_pad * a.ldn + idx] = s;
        } else if constexpr (EPI == kEpiLoss) {
          const float x = a.x[idx];
          lacc += (m < a.m_valid && n < a.n_valid) ? loss_elem<BETA>(s, x, a.beta) : 0.f;
        } else {
          const float x = a.x[idx];
          float gn, gp;
          mu_elem<BETA>(s, x, a.beta, gn, gp);
          const uint32_t nh = pack_op<OPT>(gn, 0.f);
          a.gn_hi[idx] = (uint16_t)nh;
          if constexpr (X3) a.gn_lo[idx] = (uint16_t)pack_bf16(gn - bf16_lo(nh), 0.f);
          if constexpr (BETA != kKL) {
            const uint32_t ph = pack_op<OPT>(gp, 0.f);
            a.gp_hi[idx] = (uint16_t)ph;
            if constexpr (X3) a.gp_lo[idx] = (uint16_t)pack_bf16(gp - bf16_lo(ph), 0.f);
          }
        }
      }
    }
  if constexpr (RAGK) {
    if (rag_on && wave == 0) {
      // C/D of the 16 x 16 block: column lane & 15, rows 4 (lane >> 4) + i
      const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * g4 + i;
        const int c = a.rag_c0 + (OPS == kOpsBHu ? row : r16);
        const size_t idx = OPS == kOpsBHu ? (size_t)c * a.ldn + bn * C::BN + rag_sub0 + r16
                                          : (size_t)(bm * C::BM + rag_sub0 + row) * a.ldn + c;
        if (c < a.rag_C) {
          const float x = a.x[idx];
          float gn, gp;
          mu_elem<BETA>(racc[i], x, a.beta, gn, gp);
          const uint32_t nh = pack_op<OPT>(gn, 0.f);
          a.gn_hi[idx] = (uint16_t)nh;
          if constexpr (X3) a.gn_lo[idx] = (uint16_t)pack_bf16(gn - bf16_lo(nh), 0.f);
          if constexpr (BETA != kKL) {
            const uint32_t ph = pack_op<OPT>(gp, 0.f);
            a.gp_hi[idx] = (uint16_t)ph;
            if constexpr (X3) a.gp_lo[idx] = (uint16_t)pack_bf16(gp - bf16_lo(ph), 0.f);
          }
        }
      }
    }
  }
  if constexpr (EPI == kEpiLoss) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lacc += __shfl_xor(lacc, o, 64);
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) red[wave] = lacc;
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < THREADS / 64; ++w) tot += red[w];
      a.out[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
  }
}

template <bool X3, int EPI, int BETA, int OPS = kOpsPlanes, class SH = GemmSmall, int OPT = kOpBf16, bool ND = false, bool WS = false>
int launch_gemm_one(const GemmArgs& a, hipStream_t s) {
  using C = GemmCfg<X3, SH, WS, (OPS == kOpsAHu ? 0 : 1)>;
  constexpr int kFoldBytes = (SH::THREADS / 256) * 128 * kFoldLd * 4;
  constexpr bool kRag = EPI == kEpiRatio && (OPS == kOpsBHu || OPS == kOpsAHu) && SH::THREADS == 256;
  constexpr int kLds = (EPI == kEpiFold && C::LDS_BYTES < kFoldBytes) ? kFoldBytes : C::LDS_BYTES + (kRag ? C::RAG_BYTES : 0);
  static_assert(kLds <= 160 * 1024, "LDS budget");
  if (a.m_pad % C::BM || a.n_pad % C::BN) return -3;
  auto kern = nt_gemm_kernel<X3, EPI, BETA, OPS, SH, OPT, ND, WS>;
  static bool done[64] = {};   // per device (nmfmu_fused.h: attr_flag)
  bool* flag = attr_flag(done);
  if (!*flag) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kLds);
    if (e != hipSuccess) return (int)e;
    *flag = true;
  }
  if (a.k_split > 1 && EPI != kEpiF32 && !(EPI == kEpiFold && a.tail_rows > 0)) return -3;
  if (a.rag_C > a.rag_c0) {   // eight workgroups share out a tile's 128 frames: the other dimension needs >= 8 tiles
    if (!kRag || a.rag_C - a.rag_c0 > 16 || (OPS == kOpsBHu ? a.m_pad : a.n_pad) < 8 * 128) return -3;
  }
  const int grid_y = a.m_pad / C::BM + (EPI == kEpiFold ? a.tail_rows * (a.k_split - 1) : 0);
  hipLaunchKernelGGL(kern, dim3(a.n_pad / C::BN, grid_y, EPI == kEpiFold ? 1 : a.k_split), dim3(SH::THREADS), kLds, s, a);
  return (int)hipGetLastError();
}

// f16 != 0: fp16 operand planes / window tables and fp16 ratio planes (single plane; the beta == 1 NMFD path)
int launch_gemm(int x3, int epi, int beta_kind, int ops, int f16, const GemmArgs& a, hipStream_t s);
// the same combinations with the implicit operand staged as a window of table entries (WS; nmfmu_nmfd_ws.hip); the caller
// (nmfmu_gemm) has checked gemm_window_stageable()
int launch_gemm_ws(int x3, int epi, int beta_kind, int ops, int f16, const GemmArgs& a, hipStream_t s);
// Shapes the window staging serves: one shift axis, 128 x 128 tiles, the implicit operand's rows and k extent exactly
// the logical ones (no padding rows, no dead k-chunks inside any tile), and line lengths that keep a tile inside one
// batch entry: rows (b,l): L % 128 == 0, T >= 64, k_len == R T;  rows (r,t): rows == R T (a multiple of 128), T >= 128,
// L % 64 == 0, k_len == B L.
inline bool gemm_window_stageable(int ops, const GemmArgs& a) {
  if (a.koff || (ops != kOpsBHu && ops != kOpsBHuT && ops != kOpsAHu)) return false;
  const int64_t L = (int64_t)a.tLh + a.tT - 1, bl = (int64_t)a.tB * L, rt = (int64_t)a.tR * a.tT;
  const int rows = ops == kOpsAHu ? a.m_pad : a.n_pad;
  if (ops == kOpsBHuT) return rows == rt && rt % 128 == 0 && a.tT >= 128 && L % 64 == 0 && a.k_len == bl;
  return rows == bl && L % 128 == 0 && a.tT >= 64 && a.k_len == rt && rt % 64 == 0;
}

}  // namespace nmfmu
