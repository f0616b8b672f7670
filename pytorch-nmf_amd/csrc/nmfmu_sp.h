// Software-pipelined, one-wave-per-SIMD form of the fused beta = 1 MU half-step at padded rank 256 (round 6).
//
// Same mathematics, HBM layouts and epilogue as nmfmu::fused_kernel<256, kKL, f16, kModeMU> (nmfmu_fused.h; reference
// seam nmf.py:376-378 / 389-391 + nmf.py:61-74, 122-131) -- the kernel of BASELINE configs[4]'s shard.  That kernel runs
// every wave through   panel DMA issue -> GEMM1(t) -> ratio stage(t) -> GEMM2(t) -> vmcnt(0) + barrier   strictly in
// sequence: 2 048 cycles of MFMA per tile inside ~4 200.  The two-waves-per-SIMD ping-pong of nmfmu_pp.h does not fit
// here (128 accumulator + 64 owner-fragment registers per wave), so the overlap is built INSIDE the one wave, across
// tiles, as a hand-placed instruction stream (every instruction of the loop is an asm statement; hipcc only allocates
// registers):
//
//   phase A(t): 32 MFMAs of GEMM1(t+1) -> S[(t+1) & 1]        fillers: their 32 ds_read_b128 (P1 slot of tile t+1) and the
//                                                              80 VALU instructions of the ratio stage of tile t
//                                                              (S[t & 1], X(t) -> 16-bit operands gn)
//               s_waitcnt vmcnt(4); s_barrier                  (panel tile t+2 has landed and is visible to every wave)
//   phase B(t): 32 MFMAs of GEMM2(t) (acc += gn x P1^T)        fillers: their 64 ds_read_b64_tr_b16 (P1 slot of tile t),
//                                                              the LDS-DMA of panel tile t+3 (8 pieces per wave), the
//                                                              X loads of tile t+2, the address-register upkeep
//               s_waitcnt vmcnt(12)                            (X(t+1) has landed)
//
// so that the matrix pipe sees 64 back-to-back MFMAs per tile with <= 5 single-issue fillers per MFMA gap
// (MI355X_MICROARCH.md: the budget of a 512-register kernel).  Two S tiles are live (the one being produced and the
// one being consumed); accumulators (128) and owner fragments (64) live in AGPRs -- MFMA takes them from there directly.
//
// Memory pipeline: ONE panel image (P1, row-major; GEMM2 gathers its k-contiguous operands from it with the
// transposing read, as FusedCfg::TR) in a four-slot LDS ring of 32 KiB tiles, tile u in slot u & 3, staged three tiles
// ahead by LDS-DMA; X in two register buffers, two tiles ahead, non-temporal.  The only waits are the two counted vmcnt
// above and one counted lgkmcnt in front of every MFMA (operand ring of depth 4, LDS returns in order).
// LDS addressing: ds_read offsets are 16-bit, the ring is 128 KiB -- slots 0/1 and 2/3 are reached from the same
// address registers by flipping bit 16 of the 8 (GEMM1) + 16 (GEMM2) per-lane bases twice per four tiles; the
// tile loop is unrolled by four so that ring slot, S buffer and X buffer are compile-time.  The operand prefetch does
// not cross the loop's back edge (a four-tile group drains its ring and the next one refills it: ~1 % of a group), so
// no in-flight LDS read is live where hipcc could insert a copy.
#pragma once
#include "nmfmu_fused.h"

namespace nmfmu {

template <int R_PAD, int OPT>
struct SPCfg {
  static constexpr int BM = 128, WAVES = 4, THREADS = 256;
  static constexpr int KS = R_PAD / 16;      // k-steps of GEMM1 (contraction over rank)
  static constexpr int RT = R_PAD / 32;      // 32-wide rank tiles of GEMM2's output
  static constexpr int ROWB = 2 * R_PAD;     // bytes per P1 row
  static constexpr int IMG = kBK * ROWB;     // bytes of one image tile (32 KiB)
  static constexpr int NSLOT = 4, PF = 4;
  static constexpr int N1 = 2 * KS, N2 = 4 * RT;          // MFMAs of GEMM1 / GEMM2 per tile and wave
  static constexpr int NPW = IMG / 1024 / WAVES;          // LDS-DMA pieces per wave and tile
  static constexpr int XTILE = BM * kBK * 2;              // one X tile: 128 rows x 64 columns x 2 bytes
  static constexpr int NEL = 80;                          // VALU instructions of one tile's ratio stage (fp16)
  static constexpr int LDS_MAIN = NSLOT * IMG;
  static constexpr int LDS_EPI = WAVES * 32 * R_PAD * 4;  // fused-apply staging tile per wave
  static constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static_assert(R_PAD == 256 && OPT == kOpF16, "built for padded rank 256, fp16 operands");
  static_assert(N1 == 32 && N2 == 32 && NPW == 8, "schedule tables below");
  static_assert(NSLOT * IMG == 2 * 65536, "two 64 KiB halves: bit 16 of the address registers selects the half");
};

// ---- the operand stream of a four-tile group (compile-time tables).  Iteration `it` (tile i0 + it) contributes N1
// GEMM1 entries (tile i0 + it + 1; one ds_read_b128 each) followed by N2 GEMM2 entries (tile i0 + it; two
// ds_read_b64_tr_b16 each); the final iteration of a workgroup's last group has no GEMM1 (no tile behind the last one).
struct SPEnt {
  int it;
  bool g1;
  int k;
};
template <int N1, int N2>
constexpr int sp_group_len(bool last) {
  return 3 * (N1 + N2) + (last ? N2 : N1 + N2);
}
template <int N1, int N2>
constexpr SPEnt sp_entry(int n, bool last) {
  const int it = n / (N1 + N2), l = n % (N1 + N2);
  if (last && it == 3) return SPEnt{3, false, l};
  return SPEnt{it, l < N1, l < N1 ? l : l - N1};
}
template <int N1, int N2, int PF>
constexpr int sp_younger(int n, bool last) {   // LDS read instructions issued after entry n's own at the time of its MFMA
  const int len = sp_group_len<N1, N2>(last);
  int c = 0;
  for (int m = n + 1; m < len && m < n + PF; ++m) c += sp_entry<N1, N2>(m, last).g1 ? 1 : 2;
  return c;
}

template <int R_PAD, int OPT>
__global__ void __launch_bounds__(256, 1) sp_kernel(const FusedArgs a) {
  using C = SPCfg<R_PAD, OPT>;
  constexpr int KS = C::KS, RT = C::RT, ROWB = C::ROWB, IMG = C::IMG, PF = C::PF, N1 = C::N1, N2 = C::N2;
  using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;   // MFMA column = owner row within the wave's 32
  const int hl = lane >> 5;  // lane half
  const int mb = blockIdx.x / a.nsplit;
  const int ks = blockIdx.x - mb * a.nsplit;
  const int t0 = ks * a.tiles_per_split;
  const int t1 = min(t0 + a.tiles_per_split, a.ktiles);
  const int nt = t1 - t0;    // a multiple of 4: the host rounds tiles_per_split, the padded contraction is a multiple of 256

  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");  // FP16_OVFL: conversions saturate at 65504
  // clock stamps (nmfmu_step.stamps; layout of nmfmu_pp.h): kernel entry (slot 2), loop start (0), loop end (1), exit (3)
  auto stamp = [&](int slot) {
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.debug);
    if (!dbg || wave != 0) return;
    const unsigned long long c = __builtin_amdgcn_s_memtime();
    const unsigned long long r = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      if (blockIdx.x == 0) {
        dbg[slot * 4 + 0] = c;
        dbg[slot * 4 + 1] = r;
        dbg[slot * 4 + 2] = (unsigned long long)(nt > 0 ? nt : 0);
      }
      dbg[64 + 5 * blockIdx.x + slot] = r;
      if (slot == 2)
        dbg[64 + 5 * blockIdx.x + 4] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |
                                       (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
  };
  stamp(2);

  // numerator accumulators: AGPRs for the whole kernel (every MFMA that touches them names them "+a")
  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
  // (a variable that appears ONLY as an asm operand inside a generic lambda is not implicitly captured by clang: every
  // emitter below binds its operands to local references first)
  static_for<RT>([&](auto rc) {
    f32x16& r = acc[decltype(rc)::value];
    asm volatile("" : "+a"(r));
  });

  if (nt > 0) {
    const unsigned lds_base =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const unsigned ldsw = lds_base + (unsigned)wave * (unsigned)(C::NPW * 1024);   // this wave's share of a ring slot: 8 KiB
    const char* xsrc = reinterpret_cast<const char*>(a.xp) + ((size_t)mb * a.ktiles + t0) * (size_t)C::XTILE + (size_t)wave * 4096;
    const char* p1src = reinterpret_cast<const char*>(a.p1_hi) + (size_t)t0 * IMG;
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned dv[C::NPW];   // LDS-DMA source offsets of this wave's pieces inside an image tile
#pragma unroll
    for (int p = 0; p < C::NPW; ++p) dv[p] = (unsigned)wave * (unsigned)(C::NPW * 1024) + (unsigned)p * 1024u + lane16;
    auto clampt = [&](int t) { return t < nt ? t : nt - 1; };   // tail prefetches re-read the last tile (never used)
    // source pointers are formed well ahead of the asm that reads them as an SGPR base (SALU write -> VMEM read of the
    // SGPR needs wait states hipcc does not insert for an asm consumer): laundered here, used a phase later
    auto panel_src = [&](int t) {
      const char* p = p1src + (size_t)clampt(t) * IMG;
      asm volatile("" : "+s"(p));
      return p;
    };
    auto x_src = [&](int t) {
      const char* p = xsrc + (size_t)clampt(t) * (size_t)C::XTILE;
      asm volatile("" : "+s"(p));
      return p;
    };

    // ---- owner fragments (B operand of GEMM1): row m0, rank slice 16 kk + 8 hl .. +7 -- AGPRs
    u32x4 q[KS];
    {
      const int m0 = mb * C::BM + wave * 32 + j;
      const int sw = P1Swz<R_PAD>::of(m0) << 4;
      const char* row = reinterpret_cast<const char*>(a.a1_hi) + (size_t)m0 * ROWB;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) q[kk] = ld16(row + ((kk * 32 + hl * 16) ^ sw));
    }

    // ---- per-lane LDS address registers (index algebra mirrored on the CPU in tests/test_layout_emulation.py)
    // GEMM1, entry (tt, kk): panel row pi_tt(j) = row0 + 16 tt, 16-byte slot (2 kk + hl) ^ swz(row).  The swizzle has 4 bits
    // (it does not reach slot bit 4) and does not depend on tt (16 tt leaves row & 3 and (row >> 2) & 3 alone), so
    //   address = ga[kk & 7] + 16 tt ROWB + 256 (kk >> 3) + ring-slot offset.
    const int row0 = 32 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);
    const int sw0 = P1Swz<R_PAD>::of(row0);
    int ga[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) ga[v] = row0 * ROWB + (((2 * v + hl) ^ sw0) << 4);
    // GEMM2 (transposing reads, lane map as FusedCfg::TR in nmfmu_fused.h): source lane (grp, s16) addresses panel row
    // 32 (grp >> 1) + 16 tt + 8 m2 + 4 h + (s16 >> 2), slot cslot ^ swz(row) ^ 4 rt.  With lr = (s16 >> 2) & 3 and
    // c = 2 m2 + h:  swz(row) = 4 lr | c, hence slot = (cslot ^ c) | 4 (lr ^ (rt & 3)) | 16 (rt >> 2) and
    //   address = gb[c][rt & 3] + (16 tt + 8 m2 + 4 h) ROWB + 256 (rt >> 2) + ring-slot offset.
    const int grp = lane >> 4, s16 = lane & 15;
    const int cslot = 2 * (grp & 1) + ((s16 & 3) >> 1), lr = (s16 >> 2) & 3;
    int gb[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int p = 0; p < 4; ++p)
        gb[c][p] = (32 * (grp >> 1) + (s16 >> 2)) * ROWB + (((cslot ^ c) | ((lr ^ p) << 2)) << 4) + 8 * (s16 & 1);

    // accumulator seed (eps of nmf.py:65), laundered so that hipcc keeps it resident (see nmfmu_pp.h)
    f32x16 epsv;
#pragma unroll
    for (int e = 0; e < 16; ++e) epsv[e] = kEps;
    asm volatile("" : "+v"(epsv));

    f32x16 S[2][2];           // [buffer][S^T tile]
    u32x4 xb[2][4];           // X(even tiles) / X(odd tiles)
    uint32_t gn[2][8];        // ratios of the tile whose GEMM2 comes next, packed fp16
    u32x4 r128[PF];           // operand ring, GEMM1 entries
    u32x2 rlo[PF], rhi[PF];   // operand ring, GEMM2 entries (two transposing reads each)
    float er[4];              // ratio-stage temporaries

    // ---- instruction emitters (one asm statement per hardware instruction: program order = issue order)
    auto dma_piece = [&](const char* src_tile, auto pc, auto slotc) {   // piece pc of this wave -> ring slot slotc
      constexpr int p = decltype(pc)::value, off = decltype(slotc)::value * IMG + p * 1024;
      const unsigned vo = dv[p], lw = ldsw;
      asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :
                   : "v"(vo), "s"(src_tile), "s"(lw), "n"(off)
                   : "memory", "m0", "scc");   // (s_add writes SCC: without the clobber hipcc carried its loop compare across this)
    };
    auto load_x1 = [&](const char* src, u32x4(&x)[4], auto qc) {   // one 16-byte chunk per lane (1 KiB per wave)
      constexpr int qi = decltype(qc)::value;
      const unsigned l16 = lane16;
      u32x4& dst = x[qi];
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=&v"(dst) : "v"(l16), "s"(src), "n"(qi * 1024) : "memory");
    };
    auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };
    // ratio-stage instruction k (0 .. 79) of the tile in S[sb] / xb[sb]: groups of four elements as in nmfmu_pp.h --
    // 4 x v_rcp_f32, 4 x v_fma_mix_f32 (fp16 half of the X word x fp32 reciprocal), 2 x v_cvt_pk_f16_f32.  In this order
    // no reciprocal is consumed by the instruction right behind it (trans -> VALU forwarding).
    auto ratio_op = [&](auto kc, auto sbc) {
      constexpr int k = decltype(kc)::value, sb = decltype(sbc)::value;
      constexpr int g = k / 10, pos = k % 10, tt = g >> 2, d = 2 * (g & 3);
      if constexpr (pos < 4) {
        float& r = er[pos];
        const float sv = S[sb][tt][2 * d + pos];
        asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(sv));
      } else if constexpr (pos < 8) {
        constexpr int p = pos - 4;
        float& r = er[p];
        const uint32_t w = xb[sb][2 * tt + (d >> 2)][(d & 3) + (p >> 1)];
        if constexpr ((p & 1) == 0) asm volatile("v_fma_mix_f32 %0, %1, %0, 0 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w));
        else asm volatile("v_fma_mix_f32 %0, %1, %0, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w));
      } else {
        constexpr int h = pos - 8;
        uint32_t& g2 = gn[tt][d + h];
        const float e0 = er[2 * h], e1 = er[2 * h + 1];
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(g2) : "v"(e0), "v"(e1));
      }
    };
    // the LDS reads of a stream entry into ring slot rs.  G1 entry k: tt = k & 1, kk = k >> 1, ring slot of tile i0+it+1;
    // G2 entry k: rt = k % RT, (tt, m2) = k / RT, ring slot of tile i0+it.  par = slot parity inside its 64 KiB half.
    auto issue_g1 = [&](auto kc, auto parc, auto rsc) {
      constexpr int k = decltype(kc)::value, tt = k & 1, kk = k >> 1, rs = decltype(rsc)::value;
      u32x4& dst = r128[rs];
      const int ad = ga[kk & 7];
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(tt * 16 * ROWB + (kk >> 3) * 256 + decltype(parc)::value * IMG));
    };
    auto issue_g2 = [&](auto kc, auto parc, auto rsc) {
      constexpr int k = decltype(kc)::value, rt = k % RT, c4 = k / RT, tt = c4 >> 1, m2 = c4 & 1, rs = decltype(rsc)::value;
      constexpr int o0 = (16 * tt + 8 * m2) * ROWB + (rt >> 2) * 256 + decltype(parc)::value * IMG;
      u32x2 &dlo = rlo[rs], &dhi = rhi[rs];
      const int a0 = gb[2 * m2][rt & 3], a1 = gb[2 * m2 + 1][rt & 3];
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dlo) : "v"(a0), "n"(o0));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dhi) : "v"(a1), "n"(o0 + 4 * ROWB));
    };
    auto mfma_g1 = [&](auto kc, auto nbc, auto rsc) {   // S[nb][tt] (+)= panel fragment x owner fragment kk
      constexpr int k = decltype(kc)::value, tt = k & 1, kk = k >> 1, nb = decltype(nbc)::value, rs = decltype(rsc)::value;
      f32x16& sd = S[nb][tt];
      const u32x4 pa = r128[rs], qo = q[kk];
      const f32x16 seed = epsv;
      if constexpr (kk == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(sd) : "v"(pa), "a"(qo), "v"(seed));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sd) : "v"(pa), "a"(qo));
    };
    auto mfma_g2 = [&](auto kc, auto rsc) {             // acc[rt] += ratios (tt, m2) x transposed panel fragment
      constexpr int k = decltype(kc)::value, rt = k % RT, c4 = k / RT, tt = c4 >> 1, m2 = c4 & 1, rs = decltype(rsc)::value;
      const u32x4 nh = {gn[tt][4 * m2], gn[tt][4 * m2 + 1], gn[tt][4 * m2 + 2], gn[tt][4 * m2 + 3]};
      const u32x4 bh = {rlo[rs][0], rlo[rs][1], rhi[rs][0], rhi[rs][1]};
      f32x16& ad = acc[rt];
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ad) : "v"(nh), "v"(bh));
    };
    auto issue_entry = [&](auto nc, auto lastc) {
      constexpr int n = decltype(nc)::value;
      constexpr SPEnt e = sp_entry<N1, N2>(n, decltype(lastc)::value);
      using RS = std::integral_constant<int, n % PF>;
      if constexpr (e.g1) issue_g1(std::integral_constant<int, e.k>{}, std::integral_constant<int, (e.it + 1) & 1>{}, RS{});
      else issue_g2(std::integral_constant<int, e.k>{}, std::integral_constant<int, e.it & 1>{}, RS{});
    };
    auto flip = [&](int& r) { asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(r)); };

    // ---- one group of four tiles i0 .. i0+3 (i0 a multiple of 4: ring slot = it, S / X buffer = it & 1)
    auto group = [&](int i0, auto lastc) {
      constexpr bool last = decltype(lastc)::value;
      constexpr int LEN = sp_group_len<N1, N2>(last);
      const char* dsrc = nullptr;   // panel tile i0+it+3 / X tile i0+it+2: formed at the start of iteration it's phase A,
      const char* xs = nullptr;     // read by its phase B (an SGPR base must be older than a few wait states at its VMEM use)
      static_for<PF>([&](auto pc) { issue_entry(pc, lastc); });
      static_for<LEN>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr SPEnt e = sp_entry<N1, N2>(n, last);
        constexpr bool final_it = last && e.it == 3;
        using RS = std::integral_constant<int, n % PF>;
        using SB = std::integral_constant<int, e.it & 1>;
        if constexpr (final_it && e.k == 0)   // last tile: no GEMM1 to hide the ratio stage behind
          static_for<C::NEL>([&](auto kc) { ratio_op(kc, SB{}); });
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(sp_younger<N1, N2, PF>(n, last)));
        if constexpr (e.g1) mfma_g1(std::integral_constant<int, e.k>{}, std::integral_constant<int, (e.it + 1) & 1>{}, RS{});
        else mfma_g2(std::integral_constant<int, e.k>{}, RS{});
        if constexpr (n + PF < LEN) issue_entry(std::integral_constant<int, n + PF>{}, lastc);
        if constexpr (e.g1) {
          // phase A: ratio stage of tile i0+it, 3 / 2 / 3 / 2 ... instructions per MFMA gap
          if constexpr (e.k == 0) {
            dsrc = panel_src(i0 + e.it + 3);
            xs = x_src(i0 + e.it + 2);
          }
          constexpr int lo = (e.k >> 1) * 5 + ((e.k & 1) ? 3 : 0), cnt = (e.k & 1) ? 2 : 3;
          static_for<cnt>([&](auto cc) { ratio_op(std::integral_constant<int, lo + decltype(cc)::value>{}, SB{}); });
          if constexpr (e.k == N1 - 1) {
            // panel tile i0+it+2 (issued a tile ago) has landed: leaves this wave's 4 youngest loads (X) in flight
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            barrier();
          }
        } else if constexpr (!final_it) {
          // phase B: stage panel tile i0+it+3 into the slot tile i0+it-1 has vacated, load X(i0+it+2) into the buffer
          // the ratio stage has just consumed, flip the address registers whose next tile sits in the other LDS half
          if constexpr ((e.k & 1) == 0 && e.k < 2 * C::NPW)
            dma_piece(dsrc, std::integral_constant<int, (e.k >> 1)>{}, std::integral_constant<int, (e.it + 3) & 3>{});
          if constexpr ((e.it & 1) == 0 && e.k >= 16 && e.k < 24) flip(ga[e.k - 16]);
          if constexpr (e.k >= 24 && e.k < 28) load_x1(xs, xb[e.it & 1], std::integral_constant<int, e.k - 24>{});
          if constexpr ((e.it & 1) == 1 && e.k >= 28) {
            constexpr int c = e.k - 28;
            flip(gb[c][0]), flip(gb[c][1]), flip(gb[c][2]), flip(gb[c][3]);
          }
          if constexpr (e.k == N2 - 1) {
            // X(i0+it+1) has landed: at most this iteration's 8 panel pieces + 4 X loads stay in flight
            u32x4(&xn)[4] = xb[(e.it + 1) & 1];
            asm volatile("s_waitcnt vmcnt(12)" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3])::"memory");
          }
        }
      });
    };

    // ---- prologue: panel tiles 0 .. 2, X(0), X(1); everything landed and visible; then GEMM1(0) -> S[0]
    {
      static_for<3>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const char* src = panel_src(t);
        asm volatile("s_nop 4" ::: "memory");
        static_for<C::NPW>([&](auto pc) { dma_piece(src, pc, tc); });
      });
      const char* x0 = x_src(0);
      const char* x1 = x_src(1);
      asm volatile("s_nop 4" ::: "memory");
      static_for<4>([&](auto qc) { load_x1(x0, xb[0], qc); });
      static_for<4>([&](auto qc) { load_x1(x1, xb[1], qc); });
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(xb[0][2]), "+v"(xb[0][3]), "+v"(xb[1][0]), "+v"(xb[1][1]), "+v"(xb[1][2]),
                     "+v"(xb[1][3])::"memory");
      static_for<KS>([&](auto kc) {   // owner fragments -> AGPRs
        u32x4& r = q[decltype(kc)::value];
        asm volatile("" : "+a"(r));
      });
      barrier();
      using Z = std::integral_constant<int, 0>;
      static_for<PF>([&](auto pc) { issue_g1(pc, Z{}, pc); });
      static_for<N1>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int younger = (N1 - 1 - k) < (PF - 1) ? (N1 - 1 - k) : (PF - 1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger));
        mfma_g1(kc, Z{}, std::integral_constant<int, k % PF>{});
        if constexpr (k + PF < N1) issue_g1(std::integral_constant<int, k + PF>{}, Z{}, std::integral_constant<int, k % PF>{});
      });
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // XDL write -> VALU read of S (asm MFMAs are not padded)
    }
    stamp(0);
    int i0 = 0;
    for (; i0 + 4 < nt; i0 += 4) group(i0, std::false_type{});
    group(i0, std::true_type{});
    stamp(1);
    // XDL write -> VALU read of the accumulators; the clamped tail prefetches: nothing may land in LDS / registers after this
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)"
                 : "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(xb[0][2]), "+v"(xb[0][3]), "+v"(xb[1][0]), "+v"(xb[1][1]), "+v"(xb[1][2]),
                   "+v"(xb[1][3])::"memory");
  }
  __syncthreads();               // LDS is reused by the epilogue

  // ---------------- epilogue: the ping-pong kernel's (nmfmu_pp.h), four waves.  Lane coordinates are re-derived from a
  // laundered thread id so that hipcc does not keep the epilogue's address arithmetic live across the main loop.
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, j_e = lane_e & 31, hl_e = lane_e >> 5;
  const int mrow0 = mb * C::BM + wave * 32;   // first owner row of this wave
  // accumulator register e of lane (j_e, hl_e): row (e & 3) + 8 (e >> 2) + 4 hl_e, column 32 rt + j_e
  if (a.fuse_apply) {
    // ---- nmf.py:78-92 in the epilogue (nsplit == 1: the workgroup owns complete rows); re-emits the owner's images
    constexpr int LDT = R_PAD;
    constexpr int SP = R_PAD / 8;            // sixteen-byte image slots (8 ranks) per row
    constexpr int NCH = (32 * SP) / 64;      // (row, slot) chunks per lane
    float* tile = reinterpret_cast<float*>(smem) + wave * (32 * LDT);
    const int slot_e = lane_e % SP, rl0 = lane_e / SP;
    float den8[8], csum8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { den8[k] = a.kl_den[slot_e * 8 + k]; csum8[k] = 0.f; }
    const bool vec = (a.rank & 3) == 0;
    bool clamped = false;   // fp16 images saturate at 65504 (MODE.FP16_OVFL): reported through a.status
    static_for<RT>([&](auto rtc) {
      constexpr int rt = decltype(rtc)::value;
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * hl_e) * LDT + rt * 32 + j_e] = acc[rt][e];
    });
    __syncthreads();
    const bool plain = a.l1 <= 0.f && a.l2 <= 0.f && a.gamma == 1.f && a.rank == R_PAD && mb * C::BM + C::BM <= a.M;
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
          clamped |= fv[k] > 65504.f;
        }
        *reinterpret_cast<float4*>(frow) = *reinterpret_cast<const float4*>(fv);
        *reinterpret_cast<float4*>(frow + 4) = *reinterpret_cast<const float4*>(fv + 4);
        if (a.o2_hi) {   // (the updated tile feeds the transposed image only)
          *reinterpret_cast<float4*>(trow) = *reinterpret_cast<const float4*>(fv);
          *reinterpret_cast<float4*>(trow + 4) = *reinterpret_cast<const float4*>(fv + 4);
        }
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
        clamped |= fv[k] > 65504.f;
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
    if (a.status && __any(clamped) && lane_e == 0) atomicOr(a.status, 1u);
    __syncthreads();
    // transposed image from the updated tile: 8 consecutive owner rows of one rank = one sixteen-byte slot
    // (a.o2_hi == nullptr -- NMFMU_STAGE_DMA_NOP2: nothing reads this factor's transposed image, this kernel's own GEMM2 gathers
    // from the row-major one -- skips the pass: 5.8 of the epilogue's 21 us, profiles/r06_sp_epilogue_ablations.txt)
    if (a.o2_hi)
#pragma unroll
    for (int ii = 0; ii < R_PAD / 64; ++ii) {
      const int r = ii * 64 + lane_e;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float* src = tile + (g * 8) * LDT + r;
        u32x4 hi;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) hi[qq] = pack_op<OPT>(src[(2 * qq) * LDT], src[(2 * qq + 1) * LDT]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o2_hi) + p2_offset(mrow0 + g * 8, r, R_PAD)) = hi;
      }
    }
    __syncthreads();
    // partial column sums of this workgroup's rows: the lanes sharing a slot, then the waves (fixed order)
    float* red = reinterpret_cast<float*>(smem);  // [WAVES][R_PAD]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float tot = csum8[k];
      tot += __shfl_xor(tot, 32, 64);
      if (lane_e < SP) red[wave * R_PAD + slot_e * 8 + k] = tot;
    }
    __syncthreads();
    for (int r = tid_e; r < R_PAD; r += C::THREADS)
      a.colsum_part[(size_t)mb * R_PAD + r] = (red[r] + red[R_PAD + r]) + (red[2 * R_PAD + r] + red[3 * R_PAD + r]);
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
  if (a.debug) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the epilogue's stores have left the CU
  stamp(3);
}

template <int R_PAD, int OPT>
int launch_sp_one(const FusedArgs& a, int grid, hipStream_t s) {
  using C = SPCfg<R_PAD, OPT>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
  auto kern = sp_kernel<R_PAD, OPT>;
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

// Host-side launcher (nmfmu_inst_sp.hip).  Serves beta == 1, fp16 operands and target, padded rank 256, 128-row tiles.
int launch_sp(int r_pad, int opt, const FusedArgs& a, int grid, hipStream_t s);
bool sp_available(int r_pad, int opt, int mode);

}  // namespace nmfmu
