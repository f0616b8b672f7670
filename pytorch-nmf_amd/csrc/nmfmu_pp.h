// Ping-pong form of the fused beta = 1 MU half-step for gfx950 (MI355X, CDNA4).
//
// Same mathematics, data layouts and epilogues as nmfmu::fused_kernel (nmfmu_fused.h; reference seam nmf.py:376-378 /
// 389-391 + nmf.py:61-74, 122-131), different execution structure.  The four-wave kernel runs every wave through
// GEMM1 -> elementwise -> GEMM2 in the same phase, so matrix-pipe time and VALU / memory-issue time ADD.  Here a
// workgroup is EIGHT waves -- two per SIMD (waves i and i+4 share SIMD i) -- and the two halves run the same
// instruction stream ONE SEGMENT APART:
//
//   segment   s = 2t          2t+1          2t+2          2t+3
//   waves 0-3 M(t)           E(t)          M(t+1)        E(t+1)
//   waves 4-7 E(t-1)         M(t)          E(t)          M(t+1)
//
//   M(t) "matrix segment": G1(t) = S^T tiles of k-tile t (16 MFMA at rank pad 128), then G2(t-1) = numerator update
//        with the ratios of the previous tile (16 MFMA); fillers are only the LDS operand reads (1 per MFMA).
//   E(t) "elementwise segment": ratios Gn = X / (S + eps) -> packed 16-bit operands (VALU), the LDS-DMA of the panel
//        tiles two k-tiles ahead and the X loads two k-tiles ahead (VMEM issue), the first operand reads of the
//        next M segment.
//
// One raw s_barrier separates the segments, so a SIMD's matrix pipe always has exactly one wave feeding it while the
// partner wave does everything that is slow to issue (MI355X_MICROARCH.md, "Two waves per SIMD").  Nothing is
// exchanged between the waves: each wave owns 32 owner rows, its S tile, ratios and numerator accumulators stay in
// its registers exactly as in the four-wave kernel.
//
// Memory pipeline (everything is issued from inline asm, so hipcc neither counts nor drains it):
//   panel images P1 / P2 : three-slot LDS rings filled by LDS-DMA; waves 0-3 issue P1(t+2) and P2(t+1) in their E(t)
//                          and wait for them with a COUNTED vmcnt at the end of their next M segment -- one barrier
//                          before anybody (the operand prefetch of E(t+1)) reads them.  Waves 4-7 issue no panel
//                          traffic: their segments are one barrier later, which would be one barrier too late.
//   X                    : global_load_dwordx4 ... nt straight into the lane that needs it (fragment order of
//                          nmfmu_layout.h: a wave's share of a tile is one contiguous 4 KiB piece), two register
//                          buffers, X(t+2) issued at the end of E(t) once the wave has consumed X(t).
//   the only vmcnt in the loop is `vmcnt(4)` at the end of every M segment: it leaves exactly the wave's four youngest
//   loads (its X piece two tiles ahead) in flight.
//
// Operand types: bf16 (as nmfmu_fused.h) or fp16 -- same MFMA rate, 11 instead of 8 significant bits, which is what
// brings the factors within 1e-4 of the fp32 reference at the BASELINE shapes.  In fp16 mode X
// is stored fp16 and the ratio is ONE v_fma_mix_f32 per element (fp16 source half selected by op_sel); MODE.FP16_OVFL
// is set so that an overflowing ratio saturates at 65504 instead of becoming inf.
//
// What was tried on this loop and measured not faster (profiles/r02_clock.md; the code is in the git history of
// round 2): X through an LDS ring, an 8-deep operand ring, s_setprio for the matrix segments or for the younger
// half, accumulators in AGPRs, ONE panel image with ds_read_b64_tr_b16 gathers for G2 (half the panel stream, +7 %
// clock, +15 % cycles), column sums handed over as partials instead of finalize launches; round 3 (profiles/r03_clock.md):
// touching the fused apply's fp32 master rows during the last tiles (LDS-DMA prefetch into L2) and issuing all of a
// lane's master loads before the numerators are staged -- the epilogue got longer, not shorter (29 -> 35 us): it is
// bound by instruction issue at the loop's low clock, which the branch-free "plain" form below addresses (29 -> 21 us).
// Two timing-only ablations (NMFMU_PP_ABLATE_*, diagnostic builds) show what the loop is limited by: with half the
// operand reads skipped OR performed into a sink register -- the same MFMAs seeing stale operands either way -- the
// cycles per tile stay at 2 430 and the core clock rises from 1.25 to 1.75 GHz: the energy goes into the matrix pipe's
// operand toggling, not into the LDS reads, so a tiling that halves the reads per MFMA (64 rows per wave) buys nothing.
#pragma once
#include "nmfmu_fused.h"

#ifndef NMFMU_PP_EPI_PLAIN
#define NMFMU_PP_EPI_PLAIN 1  // fused apply: branch-free form for the unregularised full-tile case
#endif
// Joule budget by result-preserving DUPLICATION (round 6; tools/gpu_r6a.sh, profiles/r06_joule_budget.md):
// -DNMFMU_PP_DUP=<bits> executes one component of the tile loop TWICE with identical results (every earlier energy
// experiment was a deletion, which corrupts the factors -- and the clock follows the data).  1: every panel ds_read_b128;
// 2: every v_rcp_f32 / v_fma_mix_f32 of the ratio stage (second copy into a dead register); 4: every X global_load (second
// copy of the SAME address into dead registers); 8: every LDS-DMA panel piece (same source, same LDS bytes); 16: every
// barrier; 32: X duplicate from the MIRRORED row block instead (a second, real HBM stream of the same size).
#ifndef NMFMU_PP_DUP
#define NMFMU_PP_DUP 0
#endif
// Where a wave issues its X loads (round 6): 0 = all of them at the end of its E segment, behind the ratios (rounds 2-5);
// 1 = one at a time in the gaps of its next M segment.  At the end of E the four E-ing waves of a CU issue their loads in the
// same few hundred cycles and the texture-address path takes 64 B per clock: ~70 cycles per 1-KiB load on the segment that
// is the pole of the ping-pong pair (profiles/r06_joule_budget.md); spread over the M segment the same loads wait for nobody.
// -1 = the shipped choice: in M for the 3-byte target (six loads per tile), at the end of E otherwise.
#ifndef NMFMU_PP_XM
#define NMFMU_PP_XM -1
#endif

namespace nmfmu {

template <int R_PAD, int OPT, int MODE, bool XR_ = false, bool LACC_ = false>
struct PPCfg {
  static constexpr int BM = 256, WAVES = 8, THREADS = 512;
  static constexpr int KS = R_PAD / 16;      // k-steps of G1 (contraction over rank)
  static constexpr int RT = R_PAD / 32;      // 32-wide rank tiles of G2's output
  static constexpr int ROWB = 2 * R_PAD;     // bytes per P1 row
  static constexpr int IMG = kBK * ROWB;     // bytes of one image tile
  static constexpr bool LOSS = MODE == kModeLoss;
  static constexpr int LEAD = 2;             // P1 runs LEAD tiles ahead, P2 LEAD - 1
  static constexpr int NSLOT = 3;            // ring depth of P1 and of P2
  static constexpr bool XR = XR_;            // 3-byte target (NMFMU_PREC_F16R): the f16 layout's words + one byte per element
  static constexpr int NX = XR ? 6 : 4;      // 16-byte X chunks per lane per tile
  static constexpr int XTILE = BM * kBK * (XR ? 3 : 2);   // one X tile: 256 rows x 64 columns x 2 (3) bytes = 32 (48) KiB
  static_assert(!XR || OPT == kOpF16, "the 3-byte target comes with fp16 operands");
  // "riding loss" (round 6, fit()'s periodic loss without its own pass over V): the MU half-step also accumulates
  // sum x log2(s) over its elements -- s = owner panel^T + eps is the reconstruction of the factors BEFORE this update, i.e. the
  // KL term of the iteration just finished that needs V (metrics.py:22: target @ log(input + eps)) -- and sum s (input.sum() +
  // eps per element, padding included: the host subtracts the count); 8 pairs of partials per workgroup into a.loss_part.
  // Three VALU per element (v_log_f32, fused multiply-add, add) on the half-steps that carry it.
  static constexpr bool LACC = LACC_;
  static_assert(!LACC || (OPT == kOpF16 && MODE == kModeMU), "riding loss: fp16 operands, MU half-step");
  static constexpr bool XM = (NMFMU_PP_DUP & (4 | 32)) ? false : (NMFMU_PP_XM < 0 ? XR : NMFMU_PP_XM != 0);   // X loads in the M segment
  static constexpr int P1_BASE = 0, P2_BASE = NSLOT * IMG;
  static constexpr int LDS_MAIN = 2 * NSLOT * IMG;
  static constexpr int LDS_EPI = LOSS ? 64 : WAVES * 32 * R_PAD * 4;   // fused-apply staging tile per wave
  static constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static constexpr int NPIECE = IMG / 1024;                 // 1-KiB DMA pieces per image tile
  static constexpr int ND = (NPIECE + 3) / 4;               // pieces per issuing wave (waves 0-3) and image
  static constexpr int NSTEP1 = 2 * KS, NSTEP2 = LOSS ? 0 : 4 * RT;
  static constexpr int PF = 4;                               // operand prefetch ring depth
  static constexpr bool SCALED = OPT == kOpBf16;             // S' = 2^23 (S + eps), seeded with the inline constant 1.0
  static_assert(NSTEP1 >= PF, "ring deeper than G1");
};

template <int R_PAD, int OPT, int MODE, bool XR = false, bool LACC = false>
__global__ void __launch_bounds__(512, 2) pp_kernel(const FusedArgs a) {
  using C = PPCfg<R_PAD, OPT, MODE, XR, LACC>;
  constexpr int NX = C::NX;
  constexpr int KS = C::KS, RT = C::RT, ROWB = C::ROWB, IMG = C::IMG, PF = C::PF;
  constexpr int NSTEP1 = C::NSTEP1, NSTEP2 = C::NSTEP2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;
  const int j = lane & 31;   // MFMA column = owner row within the wave's 32
  const int hl = lane >> 5;  // lane half
  const int mb = blockIdx.x / a.nsplit;
  const int ks = blockIdx.x - mb * a.nsplit;
  const int t0 = ks * a.tiles_per_split;
  const int t1 = min(t0 + a.tiles_per_split, a.ktiles);
  const int nt = t1 - t0;
  const int m0 = mb * C::BM + wave * 32 + j;

  if constexpr (OPT == kOpF16) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");  // FP16_OVFL: saturate
  // Clock stamps (nmfmu_step.stamps; round 6: compiled into the product build -- measured free, nothing sits in the loop;
  // tools/pp_timeline.py, bench.py roofline.in_kernel): with a.debug set, wave 0 / wave 4 of
  // workgroup 0 record the shader clock and the constant 100 MHz clock at kernel entry (slot 2), at the start (0) and
  // the end (1) of the tile loop and at kernel exit (3) -- cycles per tile, the core frequency, what prologue and
  // epilogue cost, unperturbed (nothing inside the loop); every workgroup records the 100 MHz clock at the four points
  // ([64 + 5 wg + slot]) and where it ran ([.. + 4]: XCC_ID << 32 | HW_ID).
  auto stamp = [&](int slot) {
    if constexpr (MODE == kModeMU) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.debug);
      if (!dbg) return;
      if (blockIdx.x == 0 && (wave & 3) == 0) {
        const unsigned long long c = __builtin_amdgcn_s_memtime();
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
          dbg[(half * 4 + slot) * 4 + 0] = c;
          dbg[(half * 4 + slot) * 4 + 1] = r;
          dbg[(half * 4 + slot) * 4 + 2] = (unsigned long long)nt;
        }
      }
      if (wave == 0) {
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
          dbg[64 + 5 * blockIdx.x + slot] = r;
          if (slot == 2)
            dbg[64 + 5 * blockIdx.x + 4] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |
                                           (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));
        }
      }
    }
  };
  stamp(2);

  // ---- owner fragments (B operand of G1): row m0, rank slice 16*kk + 8*hl .. +7
  // bf16: the fragments are scaled by 2^23 = 1 / eps (exact), so that the "+ eps" of nmf.py:65 becomes "+ 1.0" -- an
  // MFMA inline constant -- on S' = 2^23 (S + eps); the ratios and the numerators then carry the factor 2^-23, which
  // the epilogue removes (exact again).  This frees the sixteen registers a broadcast eps tile would occupy.
  // fp16 has no exponent range for that: it seeds the accumulators from a register tile of eps.
  constexpr bool SCALED = C::SCALED;
  // (loaded in the prologue below, after the first panel tiles' DMA and the first X loads have been issued: one
  // cold-miss latency for all three instead of two in a row)
  u32x4 q[KS];
  auto load_owner = [&]() {
    const int sw = P1Swz<R_PAD>::of(m0) << 4;
    const char* row = reinterpret_cast<const char*>(a.a1_hi) + (size_t)m0 * ROWB;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) q[kk] = ld16(row + ((kk * 32 + hl * 16) ^ sw));
  };
  auto scale_owner = [&]() {
    if constexpr (SCALED) {
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) q[kk][i] = pack_bf16(bf16_lo(q[kk][i]) * 8388608.f, bf16_hi(q[kk][i]) * 8388608.f);
    }
  };
  // ---- per-lane LDS offsets (same maps as nmfmu_fused.h: row permutation pi for G1, swizzled 16-byte slots).
  // G1 operand of step (tt, kk): a_base[tt] ^ (kk * 32)  -- the k-step only flips bits 5..7 of the slot offset, which
  // neither the row part (a multiple of ROWB) nor the ring-slot offset (a multiple of IMG) touches, so ONE register per
  // S^T tile plus an inline-constant XOR replaces sixteen precomputed addresses.
  // G2 operand of step (rt, tt, m2): b_base[tt][m2] + rt * 4096 (immediate).
  int a_base[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int row = 32 * ((j >> 2) & 1) + 16 * tt + (j & 3) + 4 * (j >> 3);
    const int sw = P1Swz<R_PAD>::of(row) << 4;
    a_base[tt] = row * ROWB + ((hl * 16) ^ sw);
  }
  int b_base[2][2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) b_base[tt][m2] = j * 128 + (((4 * hl + 2 * tt + m2) << 4) ^ (((j >> 1) & 7) << 4));
  static_assert((KS - 1) * 32 < ROWB, "k-step bits stay inside one P1 row");

  f32x16 acc[C::LOSS ? 1 : RT];
#pragma unroll
  for (int rt = 0; rt < (C::LOSS ? 1 : RT); ++rt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
  // accumulator seed of the fp16 kernels: a register tile of eps.  It is laundered through an empty asm so that hipcc
  // keeps it resident instead of re-materialising it with v_mov right in front of the (asm, hence unpadded) MFMA that
  // reads it as its C operand -- a VALU write -> XDL SrcC read hazard that made S' = S_new + stale S_old.
  f32x16 epsv;
#pragma unroll
  for (int e = 0; e < 16; ++e) epsv[e] = SCALED ? 1.0f : kEps;
  if constexpr (!SCALED) asm volatile("" : "+v"(epsv));
  float lacc = 0.f;
  float la[4] = {0.f, 0.f, 0.f, 0.f};   // riding loss: four independent chains of sum x log2(s) ...
  float ls[4] = {0.f, 0.f, 0.f, 0.f};   // ... and of sum s (the SAME s: consistent with the loss pass to the last bit of the images)

  if (nt > 0) {
    // ---- address generators.  All bases are wave-uniform (SGPR); the per-lane part is a 32-bit offset.
    const unsigned lds_base =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const char* xsrc = reinterpret_cast<const char*>(a.xp) + ((size_t)mb * a.ktiles + t0) * (size_t)C::XTILE +
                       (size_t)wave * (size_t)(NX * 1024);
    const char* p1src = reinterpret_cast<const char*>(a.p1_hi) + (size_t)t0 * IMG;
    const char* p2src = reinterpret_cast<const char*>(a.p2_hi) + (size_t)t0 * IMG;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned pvoff = (unsigned)(wave & 3) * 1024u + lane16;   // panel DMA: piece = (wave & 3) + 4 i
    auto clampt = [&](int t) { return t < nt ? t : nt - 1; };   // tail prefetches re-read the last tile (never used)
    auto dma1k = [&](const char* src, unsigned voff, unsigned lds_addr) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1"
#if NMFMU_PP_DUP & 8
                   "\n\tglobal_load_lds_dwordx4 %0, %1"
#endif
                   :
                   : "v"(voff), "s"(src), "s"(lds_addr)
                   : "memory", "m0");
    };
    // one image tile = NPIECE 1-KiB pieces; issuing wave w (0..3) moves pieces (w + 4 i) mod NPIECE (when NPIECE < 4
    // the duplicates rewrite the same bytes; they keep every issuing wave's vmcnt arithmetic identical)
    auto dma_img = [&](const char* src_tile, unsigned lds_off) {
#pragma unroll
      for (int i = 0; i < C::ND; ++i) {
        const unsigned pc = (4u * i) % (unsigned)C::NPIECE;       // compile-time part of the piece index
        if constexpr (C::NPIECE >= 4)
          dma1k(src_tile + pc * 1024u, pvoff, lds_base + lds_off + pc * 1024u + (unsigned)(wave & 3) * 1024u);
        else
          dma1k(src_tile, ((unsigned)(wave & 3) % (unsigned)C::NPIECE) * 1024u + lane16,
                lds_base + lds_off + ((unsigned)(wave & 3) % (unsigned)C::NPIECE) * 1024u);
      }
    };
    // ring slots advance by one per tile: offsets are carried incrementally (no division in the loop)
    unsigned p1_issue_off = (unsigned)(C::LEAD % C::NSLOT) * IMG;         // slot of P1(t + LEAD) at t = 0
    unsigned p2_issue_off = (unsigned)((C::LEAD - 1) % C::NSLOT) * IMG;   // slot of P2(t + LEAD - 1) at t = 0
    auto next_off = [&](unsigned off) { return off == (unsigned)(C::NSLOT - 1) * IMG ? 0u : off + IMG; };
    auto issue_panel = [&](int t) {   // called in E(t) by waves 0-3: P1(t + LEAD), P2(t + LEAD - 1)
      dma_img(p1src + (size_t)clampt(t + C::LEAD) * IMG, C::P1_BASE + p1_issue_off);
      if constexpr (!C::LOSS) dma_img(p2src + (size_t)clampt(t + C::LEAD - 1) * IMG, C::P2_BASE + p2_issue_off);
    };
    // this wave's four 1-KiB pieces of X(t), one 16-byte chunk per lane each, loaded by asm so that hipcc neither
    // counts nor waits for them; wait_x() is the counted wait that makes a buffer readable.  Non-temporal: X is read once
    // per launch (-DNMFMU_PP_X_DEFAULT_POLICY builds the default cache policy for the Infinity-Cache experiment of round 5,
    // tools/mall_probe.py / profiles/r05_mall.md: no difference)
#ifdef NMFMU_PP_X_DEFAULT_POLICY
#define NMFMU_PP_XPOL ""
#else
#define NMFMU_PP_XPOL " nt"
#endif
    u32x4 xA[NX], xB[NX];   // X(even tiles) / X(odd tiles)
#if NMFMU_PP_DUP & (4 | 32)
    // dead landing registers of the duplicated X loads, one set per X buffer: a duplicate stays in flight exactly as long
    // as its original, so it is tied at the SAME counted wait (one set tied a tile early let hipcc reuse registers that
    // loads were still landing in: memory access faults)
    u32x4 xdupA[4], xdupB[4];
#if NMFMU_PP_DUP & 32
    const long long xmirror = (long long)((int)(gridDim.x / a.nsplit) - 1 - 2 * mb) * (long long)a.ktiles * (long long)C::XTILE;
#else
    const long long xmirror = 0;
#endif
#endif
#if NMFMU_PP_DUP & (4 | 32)
#define NMFMU_PP_XDUP(x) ((&(x)[0] == &xA[0]) ? xdupA : xdupB)
#endif
    auto load_x = [&](int t, u32x4(&x)[NX]) {
      const char* src = xsrc + (size_t)clampt(t) * (size_t)C::XTILE;
#if NMFMU_PP_DUP & (4 | 32)
      u32x4(&xdup)[4] = NMFMU_PP_XDUP(x);
      asm volatile(
          "s_nop 4\n\t"
          "global_load_dwordx4 %0, %4, %5" NMFMU_PP_XPOL "\n\t"
          "global_load_dwordx4 %1, %4, %5 offset:1024" NMFMU_PP_XPOL "\n\t"
          "global_load_dwordx4 %2, %4, %5 offset:2048" NMFMU_PP_XPOL "\n\t"
          "global_load_dwordx4 %3, %4, %5 offset:3072" NMFMU_PP_XPOL
          : "=&v"(xdup[0]), "=&v"(xdup[1]), "=&v"(xdup[2]), "=&v"(xdup[3])
          : "v"(lane16), "s"(src + xmirror)
          : "memory");
#endif
      if constexpr (C::XR) {
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %6, %7" NMFMU_PP_XPOL "\n\t"
            "global_load_dwordx4 %1, %6, %7 offset:1024" NMFMU_PP_XPOL "\n\t"
            "global_load_dwordx4 %2, %6, %7 offset:2048" NMFMU_PP_XPOL "\n\t"
            "global_load_dwordx4 %3, %6, %7 offset:3072" NMFMU_PP_XPOL "\n\t"
            "global_load_dwordx4 %4, %6, %8" NMFMU_PP_XPOL "\n\t"               // the third bytes of the lane's 32 elements
            "global_load_dwordx4 %5, %6, %8 offset:1024" NMFMU_PP_XPOL              // (13-bit signed offsets end at 4095)
            : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[NX - 2]), "=&v"(x[NX - 1])
            : "v"(lane16), "s"(src), "s"(src + 4096)
            : "memory");
      } else {
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %5" NMFMU_PP_XPOL "\n\t"
            "global_load_dwordx4 %1, %4, %5 offset:1024" NMFMU_PP_XPOL "\n\t"
            "global_load_dwordx4 %2, %4, %5 offset:2048" NMFMU_PP_XPOL "\n\t"
            "global_load_dwordx4 %3, %4, %5 offset:3072" NMFMU_PP_XPOL
            : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
            : "v"(lane16), "s"(src)
            : "memory");
      }
    };
    auto tie_x = [&](u32x4(&x)[NX]) {   // "these registers are written by loads in flight / have landed": hipcc may not move them
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[NX - 2]), "+v"(x[NX - 1]));
    };
    // one 1-KiB piece of a tile's X (piece i of NX) -- the form the M segment issues between its MFMAs.  `src` / `src4` (the
    // tile's base and base + 4 KiB: 13-bit signed instruction offsets end at 4095) are formed and laundered by the caller
    // BEFORE the segment, far from their first VMEM use
    auto load_x1 = [&](auto ic, u32x4(&x)[NX], const char* src, const char* src4) {
      constexpr int i = decltype(ic)::value;
      const unsigned l16 = lane16;
      if constexpr (i < 4)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" NMFMU_PP_XPOL : "=&v"(x[i]) : "v"(l16), "s"(src), "n"(i * 1024) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" NMFMU_PP_XPOL : "=&v"(x[i]) : "v"(l16), "s"(src4), "n"((i - 4) * 1024) : "memory");
    };
    auto wait_x = [&](u32x4(&x)[NX]) {
#if NMFMU_PP_DUP & (4 | 32)
      u32x4(&xdup)[4] = NMFMU_PP_XDUP(x);
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(xdup[0]), "+v"(xdup[1]), "+v"(xdup[2]), "+v"(xdup[3])::"memory");
#else
      if constexpr (C::XR) asm volatile("s_waitcnt vmcnt(6)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[NX - 2]), "+v"(x[NX - 1])::"memory");
      else asm volatile("s_waitcnt vmcnt(4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])::"memory");
#endif
    };
    auto barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
#if NMFMU_PP_DUP & 16
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
    };

    uint32_t gn[2][8];
    f32x16 S[2];
    u32x4 ring[PF];
#ifdef NMFMU_PP_ABLATE_STALE_OPERANDS
    u32x4 abl_sink = {0u, 0u, 0u, 0u};
#endif

    // operand stream of one M segment: entries 0 .. NSTEP1-1 are G1's panel rows (P1 slot of tile t), entries
    // NSTEP1 .. NSTEP1+NSTEP2-1 G2's transposed panel slices (P2 slot of tile t-1).  `sa` / `sb` are the per-lane bases
    // with the ring-slot offset already added (loop variant, so nothing here is hoisted out of the tile loop).
    // They start at the slot of tile 0 (P1) / tile -1 (P2) and are advanced in place by +IMG or -(NSLOT-1)*IMG once per
    // tile (advance_slots), so the lane-only parts a_base / b_base are dead after this point.
    constexpr int BACK = (C::NSLOT - 1) * IMG;   // slot of "tile -1" at the start
    int sa[2] = {a_base[0] + C::P1_BASE, a_base[1] + C::P1_BASE};
    int sb[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) sb[tt][m2] = b_base[tt][m2] + C::P2_BASE + BACK;
    int rd1 = 0, rd2 = BACK;   // current slot offsets of sa / sb (uniform)
    auto advance_slots = [&]() {   // sa -> P1 slot of the next tile, sb -> P2 slot of the tile before it
      const int d1 = rd1 == BACK ? -BACK : IMG;
      const int d2 = rd2 == BACK ? -BACK : IMG;
      rd1 += d1, rd2 += d2;
      sa[0] += d1, sa[1] += d1;
      if constexpr (!C::LOSS) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2) sb[tt][m2] += d2;
      }
    };
    // The M segment is written instruction by instruction (asm volatile keeps the order): hipcc's scheduler re-orders a
    // builtin MFMA / ds_read stream and degrades the counted LDS waits to lgkmcnt(0).  Entry e of the operand stream
    // lives in ring[e % PF]; LDS returns in order, so "entry e has landed" = at most min(PF-1, NS-1-e) younger reads
    // outstanding.
    auto rd = [&](u32x4& dst, int addr, auto offc) {
#if NMFMU_PP_DUP & 1
      asm volatile("ds_read_b128 %0, %1 offset:%2\n\tds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(offc)::value));
#else
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(offc)::value));
#endif
    };
    auto opnd = [&](u32x4& dst, auto ec, auto g1c) {   // issue the LDS read of stream entry e
      constexpr int e0 = decltype(ec)::value;
      constexpr bool g1 = decltype(g1c)::value;
#ifdef NMFMU_PP_ABLATE_HALF_READS
      // timing-only ablation (wrong results): every other operand read is skipped, the MFMA re-uses a stale ring entry --
      // what a tiling with twice the rows per wave (one read feeding two MFMAs) could gain at best (profiles/r03_clock.md)
      if constexpr ((e0 & 1) != 0) return;
#endif
#ifdef NMFMU_PP_ABLATE_STALE_OPERANDS
      // control for the ablation above: every read is performed, but every other one lands in a sink register, so the
      // same MFMAs see the same stale operands -- separates the cost of the LDS reads from that of operand toggling
      if constexpr ((e0 & 1) != 0) {
        if constexpr (g1 && e0 < NSTEP1) {
          asm volatile("ds_read_b128 %0, %1" : "+v"(abl_sink) : "v"(sa[e0 & 1] ^ ((e0 >> 1) * 32)));
        } else {
          constexpr int e = e0 - (g1 ? NSTEP1 : 0);
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(abl_sink) : "v"(sb[(e / RT) >> 1][(e / RT) & 1]), "n"((e % RT) * 4096));
        }
        return;
      }
#endif
#ifdef NMFMU_PP_ABLATE_GRAY
      // timing-only ablation (wrong results), VERDICT r3 item 1a / profiles/r04_clock.md: the operand-port pattern of a
      // wave that owns 64 rows and walks (owner half, tt) resp. (owner half, rt) in Gray-code order -- exactly ONE
      // matrix-pipe input changes per MFMA (shipped loop: 1.5 in G1, 1.25 in G2).  The instruction stream is untouched:
      // every odd stream entry reads the LDS address of its even partner (identical data on the port), its MFMA takes
      // the other input from a neighbouring register (q[kk+1] in G1, the other ratio quad in G2).  =1: G1 only (the
      // experiment the verdict names; no clock change), =2: G1 and G2 (its G2 part corrupts the numerators; the factors
      // degenerate within tens of iterations and the clock follows the DATA -- not a valid bound, see r04_clock.md).
      if constexpr (g1 && e0 < NSTEP1) {
        rd(dst, sa[0] ^ ((e0 >> 1) * 32), std::integral_constant<int, 0>{});
      } else {
        constexpr int ee = e0 - (g1 ? NSTEP1 : 0);
        constexpr int e = NMFMU_PP_ABLATE_GRAY >= 2 ? (ee & ~1) : ee;
        constexpr int rt = e % RT, c = e / RT;
        rd(dst, sb[c >> 1][c & 1], std::integral_constant<int, rt * 4096>{});
      }
#else
      if constexpr (g1 && e0 < NSTEP1) {
        rd(dst, sa[e0 & 1] ^ ((e0 >> 1) * 32), std::integral_constant<int, 0>{});
      } else {
        constexpr int e = e0 - (g1 ? NSTEP1 : 0);
        constexpr int rt = e % RT, c = e / RT;
        rd(dst, sb[c >> 1][c & 1], std::integral_constant<int, rt * 4096>{});
      }
#endif
    };
    auto prefetch = [&](auto g1c) {   // first PF operands of the next M segment; issued in the preceding E segment
      static_for<PF>([&](auto pc) { opnd(ring[decltype(pc)::value], pc, g1c); });
    };
    auto mma = [&](f32x16& d, const u32x4& x, const u32x4& y) {
      if constexpr (OPT == kOpF16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y));
    };
    // M(t): G1(t) if g1, then G2(t-1) if g2
    auto no_fill = [](auto) {};
    auto matrix_segment = [&](auto g1c, auto g2c, auto fillc, auto&& fill) {   // fill(i): X piece i of the tile two ahead (C::XM)
      constexpr bool g1 = decltype(g1c)::value, g2 = decltype(g2c)::value && !C::LOSS;
      constexpr int N1 = g1 ? NSTEP1 : 0, N2 = g2 ? NSTEP2 : 0, NS = N1 + N2;
      static_for<NS>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        constexpr int younger = (NS - 1 - e) < (PF - 1) ? (NS - 1 - e) : (PF - 1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((NMFMU_PP_DUP & 1) ? 2 * younger : younger));
        u32x4& op = ring[e % PF];
        if constexpr (e < N1) {
          constexpr int tt = e & 1, kk = e >> 1;
#ifdef NMFMU_PP_ABLATE_GRAY
          constexpr int kq = (kk + tt) % KS;   // B port: q[kk], q[kk+1] | q[kk+1], q[kk+2] | ... (changes on odd entries only)
#else
          constexpr int kq = kk;
#endif
          if constexpr (kk == 0) {   // accumulator seed: the inline constant 1.0 (bf16, scaled) or the eps tile
            if constexpr (SCALED)
              asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 1.0" : "=&v"(S[tt]) : "v"(op), "v"(q[kq]));
            else
              asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(S[tt]) : "v"(op), "v"(q[kq]), "v"(epsv));
          } else {
            mma(S[tt], op, q[kq]);
          }
        } else {
          constexpr int s2 = e - N1;
#if defined(NMFMU_PP_ABLATE_GRAY) && NMFMU_PP_ABLATE_GRAY >= 2
          // Gray order over (ratio quad, rank tile): the B port (operand read, duplicated in pairs above) changes on even
          // entries, the A port (ratio quad) on odd ones (and once more at the half-way point)
          constexpr int rt = s2 % RT, c = (2 * (s2 >> 3)) ^ (((s2 + 1) >> 1) & 1), tt = c >> 1, m2 = c & 1;
#else
          constexpr int rt = s2 % RT, c = s2 / RT, tt = c >> 1, m2 = c & 1;
#endif
          const u32x4 nh = {gn[tt][4 * m2], gn[tt][4 * m2 + 1], gn[tt][4 * m2 + 2], gn[tt][4 * m2 + 3]};
          mma(acc[rt], nh, op);
        }
        if constexpr (e + PF < NS) opnd(ring[e % PF], std::integral_constant<int, e + PF>{}, g1c);
        if constexpr (decltype(fillc)::value) {
          static_for<NX>([&](auto ic) {
            if constexpr ((decltype(ic)::value * NS) / NX == e) fill(ic);
          });
        }
      });
      // asm MFMAs are not padded by hipcc: when no G2 follows G1, the S tiles are read by the VALU right after the
      // barrier -- cover the XDL write -> VALU read distance (18 wait states for a 16-pass MFMA) here
      if constexpr (g1 && N2 == 0) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    };
    // E(t): ratios of tile t from S and X(t); operand prefetch for the next M segment, panel DMA (waves 0-3) and this
    // wave's X piece two tiles ahead.  tail (the last two tiles): nothing is prefetched past the end -- an asm load
    // whose result is never read would land in registers hipcc has already handed to something else
    auto elementwise_segment = [&](int t, auto nextc, u32x4(&x)[NX], auto tailc) {
      constexpr bool next_has_g1 = decltype(nextc)::value;
      constexpr bool tail = decltype(tailc)::value;
      advance_slots();
      if constexpr (next_has_g1) prefetch(std::true_type{});
      else if constexpr (!C::LOSS) prefetch(std::false_type{});
      if constexpr (!tail) {
        if (!half) issue_panel(t);
      } else if constexpr (next_has_g1 && !C::LOSS) {   // tile nt-2: only P2(nt-1) is still needed
        if (!half) dma_img(p2src + (size_t)(t + 1) * IMG, C::P2_BASE + p2_issue_off);
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if constexpr (C::LOSS) {
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            const uint32_t w = x[2 * tt + (d >> 2)][d & 3];
            float x0, x1;
            if constexpr (C::XR) {   // bits 31..16 from the 16-bit word, bits 15..8 from the lane's third bytes (chunks 4, 5)
              const uint32_t uw = x[4 + tt][d >> 1], j0 = 2 * (d & 1);
              x0 = __builtin_bit_cast(float, __builtin_amdgcn_perm(w, uw, 0x0504000cu | (j0 << 8)));
              x1 = __builtin_bit_cast(float, __builtin_amdgcn_perm(w, uw, 0x0706000cu | ((j0 + 1) << 8)));
            } else {
              x0 = unpack_lo<OPT>(w), x1 = unpack_hi<OPT>(w);
            }
            const int k0 = (t0 + t) * kBK + 32 * hl + 16 * tt + 2 * d;
            const bool rowok = m0 < a.M;
            constexpr float un = SCALED ? 1.1920928955078125e-07f : 1.f;
            lacc += (rowok && k0 < a.K) ? loss_elem<kKL>(S[tt][2 * d] * un, x0, 1.f) : 0.f;
            lacc += (rowok && k0 + 1 < a.K) ? loss_elem<kKL>(S[tt][2 * d + 1] * un, x1, 1.f) : 0.f;
          }
        } else if constexpr (C::XR) {
          // 3-byte target, four elements per statement: 4 x v_rcp_f32, 4 x v_perm_b32 (the fp32 of the target from its 16-bit
          // word and its third byte; selector in an SGPR), 4 x v_mul_f32, 2 x v_cvt_pk_f16_f32 -- four VALU slots more than
          // the fp16 target's statement, inside a segment that runs under the partner wave's MFMAs
#pragma unroll
          for (int d = 0; d < 8; d += 2) {
            const uint32_t w0 = x[2 * tt + (d >> 2)][d & 3], w1 = x[2 * tt + (d >> 2)][(d & 3) + 1], uw = x[4 + tt][d >> 1];
            float r0, r1, r2, r3, y0, y1, y2, y3;
            uint32_t g0, g1;
            asm("v_rcp_f32 %2, %10\n\t"
                "v_rcp_f32 %3, %11\n\t"
                "v_rcp_f32 %4, %12\n\t"
                "v_rcp_f32 %5, %13\n\t"
                "v_perm_b32 %6, %14, %16, %17\n\t"
                "v_perm_b32 %7, %14, %16, %18\n\t"
                "v_perm_b32 %8, %15, %16, %19\n\t"
                "v_perm_b32 %9, %15, %16, %20\n\t"
                "v_mul_f32 %2, %6, %2\n\t"
                "v_mul_f32 %3, %7, %3\n\t"
                "v_mul_f32 %4, %8, %4\n\t"
                "v_mul_f32 %5, %9, %5\n\t"
                "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                "v_cvt_pk_f16_f32 %1, %4, %5"
                : "=&v"(g0), "=&v"(g1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
                : "v"(S[tt][2 * d]), "v"(S[tt][2 * d + 1]), "v"(S[tt][2 * d + 2]), "v"(S[tt][2 * d + 3]), "v"(w0), "v"(w1),
                  "v"(uw), "s"(0x0504000cu), "s"(0x0706010cu), "s"(0x0504020cu), "s"(0x0706030cu));
            gn[tt][d] = g0;
            gn[tt][d + 1] = g1;
            if constexpr (C::LACC) {   // y0..y3 = the four targets in fp32 (the v_perm_b32 results above)
              float l0, l1, l2, l3;
              asm("v_log_f32 %4, %12\n\t"
                  "v_log_f32 %5, %13\n\t"
                  "v_log_f32 %6, %14\n\t"
                  "v_log_f32 %7, %15\n\t"
                  "v_fma_f32 %0, %16, %4, %0\n\t"
                  "v_fma_f32 %1, %17, %5, %1\n\t"
                  "v_fma_f32 %2, %18, %6, %2\n\t"
                  "v_fma_f32 %3, %19, %7, %3\n\t"
                  "v_add_f32 %8, %8, %12\n\t"
                  "v_add_f32 %9, %9, %13\n\t"
                  "v_add_f32 %10, %10, %14\n\t"
                  "v_add_f32 %11, %11, %15"
                  : "+v"(la[0]), "+v"(la[1]), "+v"(la[2]), "+v"(la[3]), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "+v"(ls[0]),
                    "+v"(ls[1]), "+v"(ls[2]), "+v"(ls[3])
                  : "v"(S[tt][2 * d]), "v"(S[tt][2 * d + 1]), "v"(S[tt][2 * d + 2]), "v"(S[tt][2 * d + 3]), "v"(y0), "v"(y1),
                    "v"(y2), "v"(y3));
            }
          }
        } else if constexpr (OPT == kOpF16) {
          // four elements per statement: 4 x v_rcp_f32, 4 x v_fma_mix_f32 (fp16 half of the X word x fp32
          // reciprocal), 2 x v_cvt_pk_f16_f32.  Written as one asm block so that every reciprocal is at least one
          // instruction away from its consumer (trans -> VALU forwarding hazard) without hipcc's padding.
#pragma unroll
          for (int d = 0; d < 8; d += 2) {
            const uint32_t w0 = x[2 * tt + (d >> 2)][d & 3], w1 = x[2 * tt + (d >> 2)][(d & 3) + 1];
            float r0, r1, r2, r3;
            uint32_t g0, g1;
#if NMFMU_PP_DUP & 2
            float dd0, dd1;   // dead results of the duplicated reciprocals / mixed multiplies
            asm("v_rcp_f32 %2, %8\n\t"
                "v_rcp_f32 %6, %8\n\t"
                "v_rcp_f32 %3, %9\n\t"
                "v_rcp_f32 %7, %9\n\t"
                "v_rcp_f32 %4, %10\n\t"
                "v_rcp_f32 %6, %10\n\t"
                "v_rcp_f32 %5, %11\n\t"
                "v_rcp_f32 %7, %11\n\t"
                "v_fma_mix_f32 %6, %12, %2, 0 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %2, %12, %2, 0 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %7, %12, %3, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %3, %12, %3, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %6, %13, %4, 0 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %4, %13, %4, 0 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %7, %13, %5, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %5, %13, %5, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                "v_cvt_pk_f16_f32 %1, %4, %5"
                : "=&v"(g0), "=&v"(g1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(dd0), "=&v"(dd1)
                : "v"(S[tt][2 * d]), "v"(S[tt][2 * d + 1]), "v"(S[tt][2 * d + 2]), "v"(S[tt][2 * d + 3]), "v"(w0),
                  "v"(w1));
#else
            asm("v_rcp_f32 %2, %6\n\t"
                "v_rcp_f32 %3, %7\n\t"
                "v_rcp_f32 %4, %8\n\t"
                "v_rcp_f32 %5, %9\n\t"
                "v_fma_mix_f32 %2, %10, %2, 0 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %3, %10, %3, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %4, %11, %4, 0 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %5, %11, %5, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                "v_cvt_pk_f16_f32 %1, %4, %5"
                : "=&v"(g0), "=&v"(g1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                : "v"(S[tt][2 * d]), "v"(S[tt][2 * d + 1]), "v"(S[tt][2 * d + 2]), "v"(S[tt][2 * d + 3]), "v"(w0),
                  "v"(w1));
#endif
            gn[tt][d] = g0;
            gn[tt][d + 1] = g1;
            if constexpr (C::LACC) {   // sum x log2(s): the fp16 half of the stored word times the fp32 logarithm, one mixed FMA
              float l0, l1, l2, l3;
              asm("v_log_f32 %4, %12\n\t"
                  "v_log_f32 %5, %13\n\t"
                  "v_log_f32 %6, %14\n\t"
                  "v_log_f32 %7, %15\n\t"
                  "v_fma_mix_f32 %0, %16, %4, %0 op_sel_hi:[1,0,0]\n\t"
                  "v_fma_mix_f32 %1, %16, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                  "v_fma_mix_f32 %2, %17, %6, %2 op_sel_hi:[1,0,0]\n\t"
                  "v_fma_mix_f32 %3, %17, %7, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                  "v_add_f32 %8, %8, %12\n\t"
                  "v_add_f32 %9, %9, %13\n\t"
                  "v_add_f32 %10, %10, %14\n\t"
                  "v_add_f32 %11, %11, %15"
                  : "+v"(la[0]), "+v"(la[1]), "+v"(la[2]), "+v"(la[3]), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "+v"(ls[0]),
                    "+v"(ls[1]), "+v"(ls[2]), "+v"(ls[3])
                  : "v"(S[tt][2 * d]), "v"(S[tt][2 * d + 1]), "v"(S[tt][2 * d + 2]), "v"(S[tt][2 * d + 3]), "v"(w0), "v"(w1));
            }
          }
        } else {
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            const uint32_t w = x[2 * tt + (d >> 2)][d & 3];
            const float n0 = bf16_lo(w) * __builtin_amdgcn_rcpf(S[tt][2 * d]);
            const float n1 = bf16_hi(w) * __builtin_amdgcn_rcpf(S[tt][2 * d + 1]);
            gn[tt][d] = pack_bf16(n0, n1);
          }
        }
      }
      p1_issue_off = next_off(p1_issue_off);
      p2_issue_off = next_off(p2_issue_off);
      __builtin_amdgcn_sched_barrier(0);   // this wave's reads of X(t) are complete (their values were consumed)
      if constexpr (!tail && !C::XM) load_x(t + 2, x);   // ... before its register buffer is refilled (XM: in the next M segment)
    };

    // ---- prologue: P1(0), P1(1), P2(0), X(0), X(1); everything landed before the first barrier
    if (!half) {
#pragma unroll
      for (int i = 0; i < C::LEAD; ++i) dma_img(p1src + (size_t)clampt(i) * IMG, C::P1_BASE + i * IMG);
      if constexpr (!C::LOSS) {
#pragma unroll
        for (int i = 0; i < C::LEAD - 1; ++i) dma_img(p2src + (size_t)clampt(i) * IMG, C::P2_BASE + i * IMG);
      }
    }
    load_x(0, xA);
    load_x(1, xB);
    load_owner();
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(xA[0]), "+v"(xA[1]), "+v"(xA[2]), "+v"(xA[3]), "+v"(xB[0]), "+v"(xB[1]), "+v"(xB[2]), "+v"(xB[3])::"memory");
    tie_x(xA);
    tie_x(xB);
#if NMFMU_PP_DUP & (4 | 32)
    asm volatile("" : "+v"(xdupA[0]), "+v"(xdupA[1]), "+v"(xdupA[2]), "+v"(xdupA[3]), "+v"(xdupB[0]), "+v"(xdupB[1]), "+v"(xdupB[2]), "+v"(xdupB[3]));
#endif
    scale_owner();
    barrier();
    prefetch(std::true_type{});
    if (half) barrier();                       // waves 4-7 run one segment behind
    matrix_segment(std::true_type{}, std::false_type{}, std::false_type{}, no_fill);
    stamp(0);
    // one tile = barrier, E(t), barrier, M(t+1).  `xc` holds X(t); the wait that ends M(t+1) names the buffer the NEXT
    // elementwise segment reads.  The last tile is peeled (a join of two differently shaped M segments inside the loop
    // would cost a register copy of every accumulator per tile).
    auto tile_full = [&](int t, u32x4(&xc)[NX], u32x4(&xn)[NX], auto tailc) {
      constexpr bool tail = decltype(tailc)::value;
      barrier();
      elementwise_segment(t, std::true_type{}, xc, tailc);
      barrier();
      if constexpr (!tail && C::XM) {
        const char* src = xsrc + (size_t)clampt(t + 2) * (size_t)C::XTILE;
        const char* src4 = src + 4096;
        asm volatile("" : "+s"(src), "+s"(src4));
        matrix_segment(std::true_type{}, std::true_type{}, std::true_type{}, [&](auto ic) { load_x1(ic, xc, src, src4); });
      } else {
        matrix_segment(std::true_type{}, std::true_type{}, std::false_type{}, no_fill);
      }
      if constexpr (tail) {  // nothing younger than X(t+1) except tail panel pieces: drain
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3])::"memory");
        tie_x(xn);
#if NMFMU_PP_DUP & (4 | 32)
        asm volatile("" : "+v"(xdupA[0]), "+v"(xdupA[1]), "+v"(xdupA[2]), "+v"(xdupA[3]), "+v"(xdupB[0]), "+v"(xdupB[1]), "+v"(xdupB[2]), "+v"(xdupB[3]));
#endif
      }
      else
        wait_x(xn);
    };
    auto tile_last = [&](int t, u32x4(&xc)[NX], auto tailc) {
      barrier();
      elementwise_segment(t, std::false_type{}, xc, tailc);
      barrier();
      matrix_segment(std::false_type{}, std::true_type{}, std::false_type{}, no_fill);
    };
    // static register buffers => the tile loop is unrolled by two; the host gives every workgroup an EVEN number of
    // tiles (tiles_per_split is rounded up to even, the padded contraction length is a multiple of 256), so there is
    // one straight-line tail and no join of differently shaped paths
    int t = 0;
    for (; t + 2 < nt; t += 2) {
      tile_full(t, xA, xB, std::false_type{});
      tile_full(t + 1, xB, xA, std::false_type{});
    }
    tile_full(t, xA, xB, std::true_type{});
    tile_last(t + 1, xB, std::true_type{});
    stamp(1);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // XDL write -> VALU read of the accumulators (asm MFMAs are not padded)
    if (!half) barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // clamped tail prefetches: nothing may land in LDS after this
#ifdef NMFMU_PP_ABLATE_STALE_OPERANDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(abl_sink));
#endif
  }
  __syncthreads();               // LDS is reused by the epilogue
  if constexpr (C::LACC) {       // riding loss: one partial per wave, [8 * workgroup + wave] (no LDS, no barrier)
    float lsum = (la[0] + la[1]) + (la[2] + la[3]), ssum = (ls[0] + ls[1]) + (ls[2] + ls[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64), ssum += __shfl_xor(ssum, o, 64);
    if (lane == 0) a.loss_part[16 * blockIdx.x + 2 * wave] = lsum, a.loss_part[16 * blockIdx.x + 2 * wave + 1] = ssum;
  }

  // The epilogue re-derives its lane coordinates from a laundered copy of the thread id: otherwise hipcc hoists the
  // epilogue's address arithmetic above the main loop and keeps it live across it (the loop has no registers to spare).
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, j_e = lane_e & 31, hl_e = lane_e >> 5;
  // ---------------- epilogue (same register -> element map as nmfmu_fused.h: accumulator register e of lane_e (j_e, hl_e)
  // is row (e&3) + 8*(e>>2) + 4*hl_e, column 32*rt + j_e)
  if constexpr (C::LOSS) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lacc += __shfl_xor(lacc, o, 64);
    float* red = reinterpret_cast<float*>(smem);
    if (lane_e == 0) red[wave] = lacc;
    __syncthreads();
    if (tid_e == 0)
      a.loss_part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  } else {
    if constexpr (SCALED) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rt][e] *= 8388608.f;
    }
    const int mrow0 = mb * C::BM + wave * 32;   // first owner row of this wave
    if (a.fuse_apply) {
      // ---- nmf.py:78-92 in the epilogue (nsplit == 1: the workgroup owns complete rows); re-emits the owner's images
      constexpr int LDT = R_PAD;
      constexpr int SP = R_PAD / 8;            // sixteen-byte image slots (8 ranks) per row
      constexpr int NCH = (32 * SP) / 64;      // (row, slot) chunks per lane
      float* tile = reinterpret_cast<float*>(smem) + wave * (32 * LDT);
      // every lane keeps one slot (8 consecutive ranks) for the whole pass: 16-byte master loads / stores, one
      // 16-byte row-major image store per chunk, denominators and column sums of those 8 ranks in registers
      const int slot_e = lane_e % SP, rl0 = lane_e / SP;
      float den8[8], csum8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { den8[k] = a.kl_den[slot_e * 8 + k]; csum8[k] = 0.f; }
      const bool vec = (a.rank & 3) == 0;
      bool clamped = false;   // fp16 images saturate at 65504 (MODE.FP16_OVFL): reported through a.status
      // numerators -> the wave's staging tile [32][R_PAD]
      static_for<RT>([&](auto rtc) {
        constexpr int rt = decltype(rtc)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * hl_e) * LDT + rt * 32 + j_e] = acc[rt][e];
      });
      __syncthreads();
      // The common case -- no regularisation (gamma == 1 follows from 1 <= beta <= 2), every row and rank of the tile
      // valid -- without the per-element branches, bounds tests and IEEE divisions of the general form below: the
      // denominators are the same for every row, so their reciprocals are taken once (the multiplier then differs from
      // the correctly rounded quotient by at most one ulp).
      const bool plain = NMFMU_PP_EPI_PLAIN && a.l1 <= 0.f && a.l2 <= 0.f && a.gamma == 1.f && a.rank == R_PAD &&
                         mb * C::BM + C::BM <= a.M;
      if (plain) {
        float rden8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rden8[k] = 1.f / den8[k];
#pragma unroll 4
        for (int i = 0; i < NCH; ++i) {
          const int rl = i * (64 / SP) + rl0, row = mrow0 + rl, r0 = slot_e * 8;
          float* trow = tile + rl * LDT + r0;
          float* frow = a.f + (size_t)row * R_PAD + r0;
          float fv[8], nm[8];
          *reinterpret_cast<float4*>(fv) = *reinterpret_cast<const float4*>(frow);
          *reinterpret_cast<float4*>(fv + 4) = *reinterpret_cast<const float4*>(frow + 4);
          *reinterpret_cast<float4*>(nm) = *reinterpret_cast<const float4*>(trow);
          *reinterpret_cast<float4*>(nm + 4) = *reinterpret_cast<const float4*>(trow + 4);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            fv[k] *= (fmaxf(nm[k], 0.f) + kEps) * rden8[k];
            csum8[k] += fv[k];
            if constexpr (OPT == kOpF16) clamped |= fv[k] > 65504.f;
          }
          *reinterpret_cast<float4*>(frow) = *reinterpret_cast<const float4*>(fv);
          *reinterpret_cast<float4*>(frow + 4) = *reinterpret_cast<const float4*>(fv + 4);
          *reinterpret_cast<float4*>(trow) = *reinterpret_cast<const float4*>(fv);
          *reinterpret_cast<float4*>(trow + 4) = *reinterpret_cast<const float4*>(fv + 4);
          u32x4 hi;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) hi[qq] = pack_op<OPT>(fv[2 * qq], fv[2 * qq + 1]);
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o1_hi) + p1_offset(row, r0, R_PAD)) = hi;
        }
      } else
#pragma unroll 2
      for (int i = 0; i < NCH; ++i) {
        const int rl = i * (64 / SP) + rl0, row = mrow0 + rl, r0 = slot_e * 8;
        float* trow = tile + rl * LDT + r0;
        float* frow = a.f + (size_t)row * a.rank + r0;
        float fv[8], nm[8];
        *reinterpret_cast<float4*>(nm) = *reinterpret_cast<const float4*>(trow);
        *reinterpret_cast<float4*>(nm + 4) = *reinterpret_cast<const float4*>(trow + 4);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool in = row < a.M && r0 + 4 * h < a.rank;
          if (in && vec) {
            *reinterpret_cast<float4*>(fv + 4 * h) = *reinterpret_cast<const float4*>(frow + 4 * h);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) fv[4 * h + k] = (in && r0 + 4 * h + k < a.rank) ? frow[4 * h + k] : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float neg = fmaxf(nm[k], 0.f) + kEps;
          float pos = den8[k];
          if (a.l1 > 0.f) pos += a.l1;
          if (a.l2 > 0.f) pos += a.l2 * fv[k];
          float mult = neg / pos;
          if (a.gamma != 1.f) mult = mu_pow(mult, a.gamma);
          // padding rows / ranks stay exactly 0 (their denominators may be 0: 0 * inf)
          fv[k] = (row < a.M && r0 + k < a.rank) ? fv[k] * mult : 0.f;
          csum8[k] += fv[k];
          if constexpr (OPT == kOpF16) clamped |= fv[k] > 65504.f;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool in = row < a.M && r0 + 4 * h < a.rank;
          if (in && vec) {
            *reinterpret_cast<float4*>(frow + 4 * h) = *reinterpret_cast<const float4*>(fv + 4 * h);
          } else if (in) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (r0 + 4 * h + k < a.rank) frow[4 * h + k] = fv[4 * h + k];
          }
        }
        *reinterpret_cast<float4*>(trow) = *reinterpret_cast<const float4*>(fv);
        *reinterpret_cast<float4*>(trow + 4) = *reinterpret_cast<const float4*>(fv + 4);
        u32x4 hi;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) hi[qq] = pack_op<OPT>(fv[2 * qq], fv[2 * qq + 1]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o1_hi) + p1_offset(row, r0, R_PAD)) = hi;
      }
      if constexpr (OPT == kOpF16) {
        if (a.status && __any(clamped) && lane_e == 0) atomicOr(a.status, 1u);
      }
      __syncthreads();
      // transposed image from the updated tile: 8 consecutive owner rows of one rank = one sixteen-byte slot
#pragma unroll
      for (int ii = 0; ii < R_PAD / 64 + (R_PAD < 64 ? 1 : 0); ++ii) {
        const int r = ii * 64 + lane_e;
        if (r < R_PAD) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float* src = tile + (g * 8) * LDT + r;
            u32x4 hi;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) hi[qq] = pack_op<OPT>(src[(2 * qq) * LDT], src[(2 * qq + 1) * LDT]);
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o2_hi) + p2_offset(mrow0 + g * 8, r, R_PAD)) = hi;
          }
        }
      }
      __syncthreads();
      // partial column sums of this workgroup's rows: the lanes sharing a slot, then the waves (fixed order)
      float* red = reinterpret_cast<float*>(smem);  // [WAVES][R_PAD]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float tot = csum8[k];
        if constexpr (SP <= 32) tot += __shfl_xor(tot, 32, 64);
        if constexpr (SP <= 16) tot += __shfl_xor(tot, 16, 64);
        if constexpr (SP <= 8) tot += __shfl_xor(tot, 8, 64);
        if constexpr (SP <= 4) tot += __shfl_xor(tot, 4, 64);
        if (lane_e < SP) red[wave * R_PAD + slot_e * 8 + k] = tot;
      }
      __syncthreads();
      for (int r = tid_e; r < R_PAD; r += C::THREADS) {
        float tot = (red[r] + red[R_PAD + r]) + (red[2 * R_PAD + r] + red[3 * R_PAD + r]);
        tot += (red[4 * R_PAD + r] + red[5 * R_PAD + r]) + (red[6 * R_PAD + r] + red[7 * R_PAD + r]);
        a.colsum_part[(size_t)mb * R_PAD + r] = tot;
      }
    } else {
      const size_t slab = ((size_t)ks * a.M_pad + (size_t)mrow0) * R_PAD;
      static_for<RT>([&](auto rtc) {
        constexpr int rt = decltype(rtc)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * hl_e;
          a.slab_num[slab + (size_t)row * R_PAD + rt * 32 + j_e] = acc[rt][e];
        }
      });
    }
  }
  if constexpr (MODE == kModeMU) {
    if (a.debug) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the epilogue's stores have left the CU
    stamp(3);
  }
}

template <int R_PAD, int OPT, int MODE, bool XR = false, bool LACC = false>
int launch_pp_one(const FusedArgs& a, int grid, hipStream_t s) {
  using C = PPCfg<R_PAD, OPT, MODE, XR, LACC>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
  auto kern = pp_kernel<R_PAD, OPT, MODE, XR, LACC>;
  static bool done[64] = {};
  bool* flag = attr_flag(done);
  if (!*flag) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       C::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    *flag = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), C::LDS_BYTES, s, a);
  return (int)hipGetLastError();
}

// Host-side launcher (nmfmu_inst_pp.hip).  opt = OperandType, mode = kModeMU | kModeLoss.
int launch_pp(int r_pad, int opt, int mode, const FusedArgs& a, int grid, hipStream_t s, bool xr = false, bool lacc = false);
bool pp_available(int r_pad, int opt, int mode);

}  // namespace nmfmu
