// Ping-pong form of the NMFD reconstruction GEMM (round 4):  D[m][n] = sum_k A[m][k] B[n][k] + the ratio epilogue of
// nmf.py:64-66, one operand explicit (W planes), the other the implicit Toeplitz operand of H (nmfmu_gemm.h).
//
// Why: the 128 x 128 / four-wave loop of nmfmu_gemm.h is bound by what goes through the CU's LDS port and by the issue
// time of its LDS-DMA pieces: a 64 x 64 wave tile reads four fragments per four MFMAs and issues eight
// 1-KiB DMA pieces per 16 MFMAs; two workgroups per CU hide each other's issue time, at 49 % matrix-pipe occupancy.  A
// 256 x 128 tile with 128 x 64 wave tiles moves a quarter fewer bytes per MFMA -- but with ONE wave per SIMD its DMA
// issue serialises with its MFMAs (measured: 71 us vs 53).  Here the workgroup is EIGHT waves, two per SIMD, and the two
// halves split the CONTRACTION: waves 0-3 multiply the even k-tiles, waves 4-7 the odd ones, both into their own
// 256 x 128 accumulator set (128 x 64 per wave).  In time slot t the half (t & 1) runs nothing but MFMAs and fragment
// reads on k-tile t while the other half issues the LDS-DMA of k-tile t + 2 into the ring slot k-tile t - 1 has just
// vacated; one barrier per slot.  After the loop the halves exchange half of their accumulators through LDS (the ring
// is free by then), add, and each runs the elementwise epilogue on 64 of its 128 rows.
//
//   slot            0        1        2        3
//   waves 0-3    MFMA(0)  DMA(3)   MFMA(2)  DMA(5)
//   waves 4-7    DMA(2)   MFMA(1)  DMA(4)   MFMA(3)        (k-tiles 0 and 1 are issued in the prologue)
//
// One 256 x 128 tile per CU at configs[3] (1024 main channels x 8192 frames = 256 tiles).  fp16 or bf16 operands, one
// plane, beta == 1, ratio epilogue, ragged channels as two 16 x 16 x 32 blocks per workgroup (nmfmu_gemm.h: RAGK).
#pragma once
#include "nmfmu_gemm.h"

namespace nmfmu {

struct GemmPP {
  static constexpr int BM = 256, BN = 128, BK = 64, THREADS = 512, ITH = 256;   // ITH: threads that issue one stage
  static constexpr int MI = 4, NI = 2;                                          // 128 x 64 per wave
  static constexpr int A_TILE = BM * BK * 2, B_TILE = BN * BK * 2, STAGE = A_TILE + B_TILE, NSTG = 3;
  static constexpr int RAG_TILE = 16 * BK * 2;
  static constexpr int RING = NSTG * STAGE, LOOP_BYTES = RING + NSTG * RAG_TILE;
  static constexpr int XCH_ACC = 8 * 64 * 64 * 4, XCH_BYTES = XCH_ACC + 8 * 4 * 64 * 4;   // accumulator / ragged exchange
  static constexpr int LDS_BYTES = LOOP_BYTES > XCH_BYTES ? LOOP_BYTES : XCH_BYTES;
  static constexpr int PA = BM * 8 / ITH, PB = BN * 8 / ITH;                     // DMA passes per operand tile: 8, 4
};

template <int OPS, int OPT>
__global__ void __launch_bounds__(512, 1) nt_gemm_pp_kernel(const GemmArgs a) {
  using C = GemmPP;
  static_assert(OPS == kOpsBHu || OPS == kOpsAHu, "one explicit operand, one window table");
  if constexpr (OPT == kOpF16) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  constexpr int MI = C::MI, NI = C::NI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, w4 = wave & 3, tidh = tid & (C::ITH - 1);
  const int j = lane & 31, hl = lane >> 5;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int bm = blockIdx.y, bn = blockIdx.x;
  const int ktiles = a.k_len / C::BK;
  const size_t ldk = (size_t)a.k_pad * 2;

  // ---- ragged channels: the tile's frames (128 columns resp. 256 rows) are shared out 32 apiece over the workgroups of
  // the tile column (resp. row); waves (w4 < 2) of the multiplying half carry one 16 x 16 block each
  constexpr int TOP = OPS == kOpsAHu ? 0 : 1;                       // which operand is the implicit one
  constexpr int TROWS = TOP == 0 ? C::BM : C::BN;
  const int sub = OPS == kOpsBHu ? bm : bn;
  const bool rag_on = a.rag_C > a.rag_c0 && sub < TROWS / 32;
  const int rag_sub0 = 32 * sub + 16 * (w4 & 1);
  const char* exp_all = reinterpret_cast<const char*>(OPS == kOpsBHu ? a.a_hi : a.b_hi);
  const char* rag_src = exp_all + (size_t)a.rag_c0 * ldk;

  // ---- explicit operand: chunk c = p * ITH + tidh of a tile: row = c >> 3, LDS slot = c & 7, source slot = slot ^ ((row >> 1) & 7)
  constexpr int EROWS = TOP == 0 ? C::BN : C::BM, EP = TOP == 0 ? C::PB : C::PA;
  const char* exp_base = exp_all + (size_t)((OPS == kOpsBHu ? bm : bn) * EROWS) * ldk;
  const int row_t = tidh >> 3;
  const int sslot = (tidh & 7) ^ ((row_t >> 1) & 7);
  unsigned voff_exp[EP];
#pragma unroll
  for (int p = 0; p < EP; ++p) voff_exp[p] = (unsigned)((size_t)(row_t + p * (C::ITH / 8)) * ldk + sslot * 16);

  // ---- implicit operand (window table of H, rows (b, l), k = (r, t)): LDS tile chunk-major [8 k-chunks][TROWS] x 16 B,
  // chunk c = p * ITH + tidh -> k-chunk c / TROWS, row c % TROWS.  This half issues every other k-tile: its k positions
  // start at its first tile (half 1: k-tile 0, half 0: k-tile 1) and advance by two k-tiles per stage.
  constexpr int TP = TOP == 0 ? C::PA : C::PB;
  constexpr int KPP = C::ITH / TROWS;                               // k-chunks per DMA pass: 1 (A implicit) or 2
  static_assert(KPP * TP == 8, "whole k-chunks per pass");
  const char* tab = reinterpret_cast<const char*>(OPS == kOpsAHu ? a.a_hi : a.b_hi);
  const int tL = a.tLh + a.tT - 1, tJJ = a.tLh + 2 * a.tT - 2, tT8 = a.tT / 8;
  int trow;
  {
    const int row = (TOP == 0 ? bm : bn) * TROWS + (tidh % TROWS);
    const int b = row / tL, l = row - b * tL;
    trow = b < a.tB ? b * a.tR * tJJ + l + a.tT - 1 : -1;
  }
  int kq[TP], kr[TP];
#pragma unroll
  for (int p = 0; p < TP; ++p) {
    const int kc = (half == 1 ? 0 : 8) + KPP * p + __builtin_amdgcn_readfirstlane(tidh / TROWS);   // 8 chunks per k-tile
    kq[p] = kc / tT8, kr[p] = kc - kq[p] * tT8;
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto dma1k = [&](const char* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory", "m0");
  };
  // issue k-tile kt (of this half's parity) into ring slot `st`
  auto stage_issue = [&](int kt, int st) {
    const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(st * C::STAGE) + (unsigned)w4 * 1024u);
    const char* s0 = exp_base + (size_t)kt * (C::BK * 2);
#pragma unroll
    for (int p = 0; p < EP; ++p) dma1k(s0, voff_exp[p], dst0 + (TOP == 0 ? C::A_TILE : 0) + p * (C::ITH * 16));
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const int soff = 1 + kq[p] * tJJ - 8 * kr[p];
      const int idx = (kq[p] < a.tR && trow >= 0) ? trow + soff : 0;      // 0 = the all-zero chunk
      dma1k(tab, (unsigned)idx * 16u, dst0 + (TOP == 0 ? 0 : C::A_TILE) + p * (C::ITH * 16));
    }
    if (rag_on && w4 < 2)
      dma1k(rag_src + (size_t)kt * (C::BK * 2), voff_exp[0],
            __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(C::RING + st * C::RAG_TILE) + (unsigned)w4 * 1024u));
#pragma unroll
    for (int p = 0; p < TP; ++p) {          // two k-tiles ahead: k = r T + 8 tc advances by 16 chunks
      kr[p] += 16;
      while (kr[p] >= tT8) kr[p] -= tT8, ++kq[p];
    }
  };

  // accumulators: the "+ eps" of nmf.py:65 is seeded once -- in half 0
  const float seed = half == 0 ? kEps : 0.f;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = seed;
  f32x4 racc = {seed, seed, seed, seed};

  const int swz = ((j >> 1) & 7) << 4;
  // per-lane fragment bases inside a stage (row-major explicit tile: 128-byte rows, XOR-swizzled slots; chunk-major implicit tile)
  const int a_lane = TOP == 0 ? (wm * 128 + j) * 16 : (wm * 128 + j) * 128;
  const int b_lane = C::A_TILE + (TOP == 1 ? (wn * 64 + j) * 16 : (wn * 64 + j) * 128);
  auto compute = [&](int st) {
    const char* sb = smem + st * C::STAGE;
    u32x4 af[2][MI], bf[2][NI];
    auto load_frags = [&](int ks, int fb) {
      const int so = ((2 * ks + hl) << 4) ^ swz;
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[fb][i] = ld16(sb + a_lane + (TOP == 0 ? (2 * ks + hl) * (C::BM * 16) + i * 512 : i * 4096 + so));
#pragma unroll
      for (int i = 0; i < NI; ++i)
        bf[fb][i] = ld16(sb + b_lane + (TOP == 1 ? (2 * ks + hl) * (C::BN * 16) + i * 512 : i * 4096 + so));
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int fb = ks & 1;
      if (ks + 1 < 4) load_frags(ks + 1, fb ^ 1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma_op<OPT>(af[fb][mi], bf[fb][ni], acc[mi][ni]);
    }
    {   // pin the order: the first k-step's reads, then per MFMA its share of the next k-step's reads
      constexpr int RD = MI + NI, NM = MI * NI;
      __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
      static_for<4>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        static_for<NM>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          constexpr int nrd = (RD * (q + 1)) / NM - (RD * q) / NM;
          if constexpr (ks + 1 < 4 && nrd > 0) __builtin_amdgcn_sched_group_barrier(0x100, nrd, 0);
        });
      });
    }
    if (rag_on && w4 < 2) {
      const int r16 = lane & 15, g4 = lane >> 4;
      const char* rt = smem + C::RING + st * C::RAG_TILE;
      const char* it = sb + (TOP == 0 ? 0 : C::A_TILE);
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const int q = 4 * ks2 + g4;
        const u32x4 eh = ld16(rt + r16 * 128 + ((q ^ ((r16 >> 1) & 7)) << 4));
        const u32x4 ih = ld16(it + (q * TROWS + rag_sub0 + r16) * 16);
        racc = OPS == kOpsBHu ? mfma16_op<OPT>(eh, ih, racc) : mfma16_op<OPT>(ih, eh, racc);
      }
    }
  };

  // ---- prologue: k-tile 0 (issued by half 1) and k-tile 1 (half 0) land before slot 0
  if (half == 1) {
    if (ktiles > 0) stage_issue(0, 0);
  } else if (ktiles > 1) {
    stage_issue(1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int st = 0;
  for (int slot = 0; slot < ktiles; ++slot) {
    if ((slot & 1) == half) {
      compute(st);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this half's pieces of k-tile slot + 1 (issued a slot ago) have landed
    } else if (slot + 2 < ktiles) {
      stage_issue(slot + 2, st == 0 ? 2 : st - 1);        // ring slot of k-tile slot - 1: free since the last barrier
    }
    __syncthreads();
    st = st == 2 ? 0 : st + 1;
  }

  // ---- merge the two accumulator sets: wave w and wave w ^ 4 own the same 128 x 64 sub-tile; half h keeps the 32-row
  // blocks mi = 2 h, 2 h + 1 and hands the other two to its partner (lane-contiguous floats: conflict free)
  float* xch = reinterpret_cast<float*>(smem);
  {
    float* mine = xch + (size_t)wave * (64 * 64);
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int mi = 2 * (1 - half) + m2;      // what the PARTNER keeps
          // (mi depends on the wave-uniform `half`: both cases are compiled as register selects below)
          const float v = half == 0 ? acc[2 + m2][ni][e] : acc[m2][ni][e];
          (void)mi;
          mine[((m2 * NI + ni) * 16 + e) * 64 + lane] = v;
        }
    if (w4 < 2) {
      float* rm = xch + C::XCH_ACC / 4 + (size_t)wave * (4 * 64);
#pragma unroll
      for (int i = 0; i < 4; ++i) rm[i * 64 + lane] = racc[i];
    }
  }
  __syncthreads();
  f32x16 fin[2][NI];
  {
    const float* theirs = xch + (size_t)(wave ^ 4) * (64 * 64);
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          fin[m2][ni][e] = (half == 0 ? acc[m2][ni][e] : acc[2 + m2][ni][e]) + theirs[((m2 * NI + ni) * 16 + e) * 64 + lane];
  }
  // ---- epilogue (nmf.py:64-66): Gn = X / (S + eps) as 16-bit plane, rows 64 half .. 64 half + 63 of the wave's sub-tile
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = bn * C::BN + wn * 64 + ni * 32 + j;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm * C::BM + wm * 128 + (2 * half + m2) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
        const size_t idx = (size_t)m * a.ldn + n;
        float gn, gp;
        mu_elem<kKL>(fin[m2][ni][e], a.x[idx], a.beta, gn, gp);
        a.gn_hi[idx] = (uint16_t)pack_op<OPT>(gn, 0.f);
      }
    }
  if (rag_on && half == 0 && w4 < 2) {
    const float* rp = xch + C::XCH_ACC / 4 + (size_t)(wave ^ 4) * (4 * 64);
    const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g4 + i;
      const int c = a.rag_c0 + (OPS == kOpsBHu ? row : r16);
      const size_t idx = OPS == kOpsBHu ? (size_t)c * a.ldn + bn * C::BN + rag_sub0 + r16
                                        : (size_t)(bm * C::BM + rag_sub0 + row) * a.ldn + c;
      if (c < a.rag_C) {
        float gn, gp;
        mu_elem<kKL>(racc[i] + rp[i * 64 + lane], a.x[idx], a.beta, gn, gp);
        a.gn_hi[idx] = (uint16_t)pack_op<OPT>(gn, 0.f);
      }
    }
  }
}

template <int OPS, int OPT>
int launch_gemm_pp_one(const GemmArgs& a, hipStream_t s) {
  using C = GemmPP;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
  if (a.m_pad % C::BM || a.n_pad % C::BN || a.k_len % C::BK || a.k_split > 1) return -3;
  if (a.rag_C > a.rag_c0 && (a.rag_C - a.rag_c0 > 16 || (OPS == kOpsBHu ? a.m_pad : a.n_pad) < 8 * 128)) return -3;
  auto kern = nt_gemm_pp_kernel<OPS, OPT>;
  static bool done[64] = {};
  bool* flag = attr_flag(done);
  if (!*flag) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       C::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    *flag = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.n_pad / C::BN, a.m_pad / C::BM), dim3(C::THREADS), C::LDS_BYTES, s, a);
  return (int)hipGetLastError();
}

// ops: kOpsBHu | kOpsAHu; f16: fp16 (else bf16) operands.  beta == 1, ratio epilogue, one operand plane.
int launch_gemm_pp(int ops, int f16, const GemmArgs& a, hipStream_t s);

}  // namespace nmfmu
