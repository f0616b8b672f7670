// Software-pipelined, one-wave-per-SIMD form of the fused beta != 1 MU half-step at padded rank 128 (round 6).
//
// Same mathematics, HBM layouts and epilogues as nmfmu::fused_kernel<128, beta, f16, kModeMU> (nmfmu_fused.h; reference
// seam nmf.py:61-92) -- the kernel of BASELINE configs[2]'s beta < 1 legs.  That kernel keeps two accumulator sets
// (numerator and denominator) and runs GEMM1 (16 MFMAs) -> elementwise stage (160 .. 224 VALU instructions: reciprocal /
// rsqrt / log + exp, power-of-two scaling, multiplies, two conversions per element pair) -> GEMM2 (32 MFMAs) -> drain + barrier
// back to back, two workgroups per CU: ~3 950 SIMD cycles per wave and tile for 1 536 of MFMA.  Here, as in nmfmu_sp.h
// (padded rank 256, beta = 1), ONE wave per SIMD runs a hand-placed stream in which the matrix pipe never waits for the
// elementwise stage:
//
//   Q'(t): 16 MFMAs of GEMM1(t+1) -> S[(t+1) & 1]              fillers: their 16 ds_read_b128, the LAST third of the
//                                                                elementwise stage of tile t
//          s_waitcnt vmcnt(0); s_barrier                         (X(t+1) and panel tile t+2 have landed and are visible)
//   P'(t): 32 MFMAs of GEMM2(t): num += gn x P1^T, den += gp x P1^T (each transposed panel fragment feeds both)
//                                                                fillers: 32 ds_read_b64_tr_b16, the FIRST two thirds of
//                                                                the elementwise stage of tile t+1, the X loads of tile
//                                                                t+2, the LDS-DMA of panel tile t+3
//
// The packed operands gn / gp are SINGLE-buffered: the elementwise stage of tile t+1 overwrites a word only after the
// last MFMA of GEMM2(t) that reads it has issued (GEMM2 consumes the words in the order the stage produces them; the
// schedule below holds a conversion back until its word is free), and tests/test_layout_emulation.py replays the schedule
// against exactly that rule.  Two S tiles and two X buffers are live; both accumulator sets (128 registers) and the owner
// fragments (32) live in AGPRs.  The panel ring has four 16 KiB slots = 64 KiB: every operand is base register + 16-bit
// immediate, no address upkeep at all.
#pragma once
#include "nmfmu_fused.h"

namespace nmfmu {

template <int BETA>
constexpr int sp2_group_ops() {   // VALU instructions per group of four elements (see ratio_op below)
  return BETA == kIS ? 20 : BETA == kSqrt ? 24 : BETA == kSqrt3 ? 16 : 28;
}

// ---- where every instruction of a tile's elementwise stage goes (compile-time plan; mirrored in
// tests/test_layout_emulation.py).  Instructions keep their order.  The head runs in P'(t-1) from gap G0 on, CAP (+1 in
// even gaps when ALT) per MFMA gap; a conversion (the last four instructions of a group: they write the gn / gp words of
// GEMM2 entries 4 c4 .. 4 c4 + 3, c4 = group >> 1) waits for gap 8 c4 + 8 -- the MFMAs 8 c4 .. 8 c4 + 7 of GEMM2(t-1) have
// issued by then.  Whatever does not fit P' forms the tail in Q'(t), spread evenly over its 16 gaps.
template <int BETA>
struct SP2Plan {
  static constexpr int GOPS = sp2_group_ops<BETA>(), NE = 8 * GOPS;
  static constexpr int G0 = 3;
  static constexpr int CAP = BETA == kSqrt3 ? 3 : (BETA == kGen ? 5 : 4);
  static constexpr bool ALT = BETA == kSqrt;
  int n_head = 0;
  int gap[8 * 28] = {};   // gap of instruction k: P' gap (k < n_head) or Q' gap (k >= n_head)
  constexpr SP2Plan() {
    int k = 0;
    for (int g = G0; g < 32 && k < NE; ++g) {
      const int cap = CAP + ((ALT && (g & 1) == 0) ? 1 : 0);
      for (int c = 0; c < cap && k < NE; ++c) {
        const int grp = k / GOPS, pos = k % GOPS;
        if (pos >= GOPS - 4 && g < 8 * (grp >> 1) + 8) break;   // its words are still being read by GEMM2 of the tile before
        gap[k++] = g;
      }
    }
    n_head = k;
    const int nq = NE - n_head;
    for (int i = 0; i < nq; ++i) gap[n_head + i] = (i * 16) / (nq > 0 ? nq : 1);
  }
};

template <int BETA>
struct SP2PlanOf {
  static constexpr SP2Plan<BETA> value{};
};

template <int R_PAD, int OPT, int BETA>
struct SP2Cfg {
  static constexpr int BM = 128, WAVES = 4, THREADS = 256;
  static constexpr int KS = R_PAD / 16, RT = R_PAD / 32, ROWB = 2 * R_PAD, IMG = kBK * ROWB;
  static constexpr int NSLOT = 4, PF = 4;
  static constexpr int N1 = 2 * KS, N2 = 4 * RT;          // stream entries of GEMM1 / GEMM2 per tile (GEMM2: two MFMAs each)
  static constexpr int NPW = IMG / 1024 / WAVES;          // LDS-DMA pieces per wave and tile
  static constexpr int XTILE = BM * kBK * 2;
  static constexpr bool SCALE = BETA == kIS || BETA == kSqrt || BETA == kGen;   // as FusedCfg::SCALE for fp16 operands
  static constexpr int LDS_MAIN = NSLOT * IMG;
  static constexpr int LDS_EPI = WAVES * 32 * R_PAD * 4;
  static constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static_assert(R_PAD == 128 && OPT == kOpF16, "built for padded rank 128, fp16 operands");
  static_assert(BETA == kIS || BETA == kSqrt || BETA == kSqrt3 || BETA == kGen, "beta = 2 keeps the four-wave kernel / the Gram path");
  static_assert(N1 == 16 && N2 == 16 && NPW == 4 && NSLOT * IMG <= 65536, "schedule tables below; every LDS offset an immediate");
};

// ---- operand stream of a four-tile group: iteration it = N1 GEMM1 entries (tile i0+it+1; one ds_read_b128 each), then N2
// GEMM2 entries (tile i0+it; two ds_read_b64_tr_b16 and TWO MFMAs each); a workgroup's final iteration has no GEMM1
struct SP2Ent {
  int it;
  bool g1;
  int k;
};
template <int N1, int N2>
constexpr int sp2_group_len(bool last) {
  return 3 * (N1 + N2) + (last ? N2 : N1 + N2);
}
template <int N1, int N2>
constexpr SP2Ent sp2_entry(int n, bool last) {
  const int it = n / (N1 + N2), l = n % (N1 + N2);
  if (last && it == 3) return SP2Ent{3, false, l};
  return SP2Ent{it, l < N1, l < N1 ? l : l - N1};
}
template <int N1, int N2, int PF>
constexpr int sp2_younger(int n, bool last) {
  const int len = sp2_group_len<N1, N2>(last);
  int c = 0;
  for (int m = n + 1; m < len && m < n + PF; ++m) c += sp2_entry<N1, N2>(m, last).g1 ? 1 : 2;
  return c;
}

template <int R_PAD, int OPT, int BETA>
__global__ void __launch_bounds__(256, 1) sp2_kernel(const FusedArgs a) {
  using C = SP2Cfg<R_PAD, OPT, BETA>;
  using PL = SP2Plan<BETA>;
  constexpr int KS = C::KS, RT = C::RT, ROWB = C::ROWB, IMG = C::IMG, PF = C::PF, N1 = C::N1, N2 = C::N2;
  using PH = SP2PlanOf<BETA>;   // PH::value = the compile-time plan of the elementwise stage
  using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int hl = lane >> 5;
  const int mb = blockIdx.x / a.nsplit;
  const int ks = blockIdx.x - mb * a.nsplit;
  const int t0 = ks * a.tiles_per_split;
  const int t1 = min(t0 + a.tiles_per_split, a.ktiles);
  const int nt = t1 - t0;    // a multiple of 4 (the host rounds tiles_per_split)

  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");  // FP16_OVFL: conversions saturate at 65504

  // ---- fp16 operands, negative powers of S: scale 2^ki of Gn / Gp from the typical S (as nmfmu_fused.h; identical in every workgroup)
  int ki = 0;
  if constexpr (C::SCALE) {
    if (a.cs_owner && a.cs_panel) {
      float p = 0.f;
      for (int r = lane; r < R_PAD; r += 64) p += a.cs_owner[r] * a.cs_panel[r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
      const float styp = p / ((float)a.M * (float)a.K) + kEps;
      if (styp > 0.f && styp < 3.0e38f) {
        const float bexp = BETA == kIS ? -1.f : (BETA == kSqrt ? -0.5f : a.beta - 1.f);
        ki = (int)fminf(fmaxf(-rintf(bexp * log2f(styp)), -40.f), 40.f);
      }
      ki = __builtin_amdgcn_readfirstlane(ki);
    }
  }

  // numerator / denominator accumulators: AGPRs for the whole kernel
  f32x16 on[RT], op[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int e = 0; e < 16; ++e) on[rt][e] = 0.f, op[rt][e] = 0.f;
  static_for<RT>([&](auto rc) {
    f32x16 &r0 = on[decltype(rc)::value], &r1 = op[decltype(rc)::value];
    asm volatile("" : "+a"(r0));
    asm volatile("" : "+a"(r1));
  });

  if (nt > 0) {
    const unsigned lds_base =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const unsigned ldsw = lds_base + (unsigned)wave * (unsigned)(C::NPW * 1024);   // this wave's share of a ring slot: 4 KiB
    const char* xsrc = reinterpret_cast<const char*>(a.xp) + ((size_t)mb * a.ktiles + t0) * (size_t)C::XTILE + (size_t)wave * 4096;
    const char* p1src = reinterpret_cast<const char*>(a.p1_hi) + (size_t)t0 * IMG;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned dv = (unsigned)wave * (unsigned)(C::NPW * 1024) + lane16;   // LDS-DMA source offset of this wave's share
    auto clampt = [&](int t) { return t < nt ? t : nt - 1; };
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

    u32x4 q[KS];   // owner fragments -> AGPRs
    {
      const int m0 = mb * C::BM + wave * 32 + j;
      const int sw = P1Swz<R_PAD>::of(m0) << 4;
      const char* row = reinterpret_cast<const char*>(a.a1_hi) + (size_t)m0 * ROWB;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) q[kk] = ld16(row + ((kk * 32 + hl * 16) ^ sw));
    }

    // ---- per-lane LDS address registers (the algebra of nmfmu_sp.h at 16 slots per row: no slot bit above the swizzle)
    //   GEMM1 (tt, kk):        ga[kk] + 16 tt ROWB + slot IMG
    //   GEMM2 (tt, m2, h, rt): gb[2 m2 + h][rt] + (16 tt + 8 m2 + 4 h) ROWB + slot IMG
    const int row0 = 32 * ((j >> 2) & 1) + (j & 3) + 4 * (j >> 3);
    const int sw0 = P1Swz<R_PAD>::of(row0);
    int ga[KS];
#pragma unroll
    for (int v = 0; v < KS; ++v) ga[v] = row0 * ROWB + (((2 * v + hl) ^ sw0) << 4);
    const int grp = lane >> 4, s16 = lane & 15;
    const int cslot = 2 * (grp & 1) + ((s16 & 3) >> 1), lr = (s16 >> 2) & 3;
    int gb[4][RT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int p = 0; p < RT; ++p)
        gb[c][p] = (32 * (grp >> 1) + (s16 >> 2)) * ROWB + (((cslot ^ c) | ((lr ^ p) << 2)) << 4) + 8 * (s16 & 1);

    f32x16 epsv;
#pragma unroll
    for (int e = 0; e < 16; ++e) epsv[e] = kEps;
    asm volatile("" : "+v"(epsv));
    const float bm1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.beta - 1.f)));
    float kif = (float)ki;   // (generic beta: the scale rides in the exponent of exp2; a VGPR -- one SGPR operand per VOP3)
    asm volatile("" : "+v"(kif));

    f32x16 S[2][2];
    u32x4 xb[2][4];
    uint32_t gn[2][8], gp[2][8];   // packed fp16 operands of GEMM2 (single-buffered: see the header comment)
    u32x4 r128[PF];
    u32x2 rlo[PF], rhi[PF];
    float ea[4], eb[4], ec[4];

    auto dma_piece = [&](const char* src_tile, auto pc, auto slotc) {
      // the instruction offset moves the LDS destination AND the global source (tools/ubench/dma_off_probe.hip): one M0
      // value and one offset register per slot, the piece is the immediate
      constexpr int p = decltype(pc)::value, off = decltype(slotc)::value * IMG;
      const unsigned vo = dv, lw = ldsw;
      asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%4"
                   :
                   : "v"(vo), "s"(src_tile), "s"(lw), "n"(off), "n"(p * 1024)
                   : "memory", "m0", "scc");
    };
    auto load_x1 = [&](const char* src, u32x4(&x)[4], auto qc) {
      constexpr int qi = decltype(qc)::value;
      const unsigned l16 = lane16;
      u32x4& dst = x[qi];
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=&v"(dst) : "v"(l16), "s"(src), "n"(qi * 1024) : "memory");
    };
    auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };

    // ---- elementwise instruction k of the tile in S[sb] / xb[sb] (nmf.py:61-74; the arithmetic of mu_elem_scaled /
    // mu_elem with the fp16 target folded into one v_fma_mix_f32 per element, instruction by instruction).  Groups of
    // four elements g = (tt, d): transcendental(s) first, so that none is consumed by the instruction right behind it.
    auto mix = [&](float& r, uint32_t w, auto pc) {   // r = fp16 half of w (low: p even, high: p odd) * r
      if constexpr ((decltype(pc)::value & 1) == 0) asm volatile("v_fma_mix_f32 %0, %1, %0, 0 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w));
      else asm volatile("v_fma_mix_f32 %0, %1, %0, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w));
    };
    auto pack = [&](uint32_t& dst, float e0, float e1) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(dst) : "v"(e0), "v"(e1)); };
    auto ratio_op = [&](auto kc, auto sbc) {
      constexpr int k = decltype(kc)::value, sb = decltype(sbc)::value, GO = PL::GOPS;
      constexpr int g = k / GO, pos = k % GO, tt = g >> 2, d = 2 * (g & 3), i = pos & 3, stage = pos >> 2;
      const float sv = S[sb][tt][2 * d + i];
      const uint32_t w = xb[sb][2 * tt + (d >> 2)][(d & 3) + (i >> 1)];
      float &A = ea[i], &B = eb[i], &Cc = ec[i];
      const int kis = ki;
      using IC = std::integral_constant<int, i>;
      if constexpr (pos >= GO - 4) {          // the four conversions: gn words d, d+1, then gp words d, d+1
        constexpr int h = pos - (GO - 4), i0c = 2 * (h & 1);
        constexpr bool n_in_c = BETA == kSqrt || BETA == kGen;   // which temporaries hold the numerator terms
        if constexpr (h < 2) {
          if constexpr (n_in_c) pack(gn[tt][d + h], ec[i0c], ec[i0c + 1]);
          else pack(gn[tt][d + h], ea[i0c], ea[i0c + 1]);
        } else {
          pack(gp[tt][d + h - 2], eb[i0c], eb[i0c + 1]);
        }
      } else if constexpr (BETA == kIS) {     // r = rcp(s); gp = ldexp(r, ki); gn = gp r x
        if constexpr (stage == 0) asm volatile("v_rcp_f32 %0, %1" : "=v"(A) : "v"(sv));
        else if constexpr (stage == 1) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(B) : "v"(A), "s"(kis));
        else if constexpr (stage == 2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(A) : "v"(B));
        else mix(A, w, IC{});
      } else if constexpr (BETA == kSqrt) {   // r = rsq(s); gp = ldexp(r, ki); gn = gp r r x
        if constexpr (stage == 0) asm volatile("v_rsq_f32 %0, %1" : "=v"(A) : "v"(sv));
        else if constexpr (stage == 1) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(B) : "v"(A), "s"(kis));
        else if constexpr (stage == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(Cc) : "v"(B), "v"(A));
        else if constexpr (stage == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(Cc) : "v"(A));
        else mix(Cc, w, IC{});
      } else if constexpr (BETA == kSqrt3) {  // r = rsq(s); gp = s r; gn = r x
        if constexpr (stage == 0) asm volatile("v_rsq_f32 %0, %1" : "=v"(A) : "v"(sv));
        else if constexpr (stage == 1) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(B) : "v"(sv), "v"(A));
        else mix(A, w, IC{});
      } else {                                // lg = log2(s); gp = exp2((beta - 1) lg + ki); gn = gp rcp(s) x
        const float kf = kif, b1 = bm1;
        if constexpr (stage == 0) asm volatile("v_log_f32 %0, %1" : "=v"(A) : "v"(sv));
        else if constexpr (stage == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "s"(b1), "v"(kf));
        else if constexpr (stage == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(B) : "v"(A));
        else if constexpr (stage == 3) asm volatile("v_rcp_f32 %0, %1" : "=v"(Cc) : "v"(sv));
        else if constexpr (stage == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(Cc) : "v"(B));
        else mix(Cc, w, IC{});
      }
    };

    auto issue_g1 = [&](auto kc, auto slotc, auto rsc) {
      constexpr int k = decltype(kc)::value, tt = k & 1, kk = k >> 1, rs = decltype(rsc)::value;
      u32x4& dst = r128[rs];
      const int ad = ga[kk];
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(tt * 16 * ROWB + decltype(slotc)::value * IMG));
    };
    auto issue_g2 = [&](auto kc, auto slotc, auto rsc) {
      constexpr int k = decltype(kc)::value, rt = k % RT, c4 = k / RT, tt = c4 >> 1, m2 = c4 & 1, rs = decltype(rsc)::value;
      constexpr int o0 = (16 * tt + 8 * m2) * ROWB + decltype(slotc)::value * IMG;
      u32x2 &dlo = rlo[rs], &dhi = rhi[rs];
      const int a0 = gb[2 * m2][rt], a1 = gb[2 * m2 + 1][rt];
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dlo) : "v"(a0), "n"(o0));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dhi) : "v"(a1), "n"(o0 + 4 * ROWB));
    };
    auto mfma_g1 = [&](auto kc, auto nbc, auto rsc) {
      constexpr int k = decltype(kc)::value, tt = k & 1, kk = k >> 1, nb = decltype(nbc)::value, rs = decltype(rsc)::value;
      f32x16& sd = S[nb][tt];
      const u32x4 pa = r128[rs], qo = q[kk];
      const f32x16 seed = epsv;
      if constexpr (kk == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(sd) : "v"(pa), "a"(qo), "v"(seed));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sd) : "v"(pa), "a"(qo));
    };
    auto mfma_g2 = [&](auto kc, auto rsc, auto denc) {   // num (denc = 0) or den (1) MFMA of GEMM2 entry k
      constexpr int k = decltype(kc)::value, rt = k % RT, c4 = k / RT, tt = c4 >> 1, m2 = c4 & 1, rs = decltype(rsc)::value;
      const u32x4 bh = {rlo[rs][0], rlo[rs][1], rhi[rs][0], rhi[rs][1]};
      if constexpr (decltype(denc)::value == 0) {
        const u32x4 nh = {gn[tt][4 * m2], gn[tt][4 * m2 + 1], gn[tt][4 * m2 + 2], gn[tt][4 * m2 + 3]};
        f32x16& ad = on[rt];
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ad) : "v"(nh), "v"(bh));
      } else {
        const u32x4 ph = {gp[tt][4 * m2], gp[tt][4 * m2 + 1], gp[tt][4 * m2 + 2], gp[tt][4 * m2 + 3]};
        f32x16& ad = op[rt];
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ad) : "v"(ph), "v"(bh));
      }
    };
    auto issue_entry = [&](auto nc, auto lastc) {
      constexpr int n = decltype(nc)::value;
      constexpr SP2Ent e = sp2_entry<N1, N2>(n, decltype(lastc)::value);
      using RS = std::integral_constant<int, n % PF>;
      if constexpr (e.g1) issue_g1(std::integral_constant<int, e.k>{}, std::integral_constant<int, (e.it + 1) & 3>{}, RS{});
      else issue_g2(std::integral_constant<int, e.k>{}, std::integral_constant<int, e.it & 3>{}, RS{});
    };
    // the instructions of the elementwise plan that belong to MFMA gap `gapc` of a phase (head: P', tile sb; tail: Q')
    auto ratio_gap = [&](auto headc, auto gapc, auto sbc) {
      constexpr bool head = decltype(headc)::value;
      constexpr int gi = decltype(gapc)::value;
      constexpr int lo = head ? 0 : PH::value.n_head, hi = head ? PH::value.n_head : PL::NE;
      static_for<hi - lo>([&](auto kc) {
        constexpr int k = lo + decltype(kc)::value;
        if constexpr (PH::value.gap[k] == gi) ratio_op(std::integral_constant<int, k>{}, sbc);
      });
    };

    // ---- one group of four tiles i0 .. i0+3 (ring slot = it, S / X buffer = it & 1)
    auto group = [&](int i0, auto lastc) {
      constexpr bool last = decltype(lastc)::value;
      constexpr int LEN = sp2_group_len<N1, N2>(last);
      const char* dsrc = nullptr;
      const char* xs = nullptr;
      static_for<PF>([&](auto pc) { issue_entry(pc, lastc); });
      static_for<LEN>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr SP2Ent e = sp2_entry<N1, N2>(n, last);
        constexpr bool final_it = last && e.it == 3;
        using RS = std::integral_constant<int, n % PF>;
        using K = std::integral_constant<int, e.k>;
        if constexpr (final_it && e.k == 0)   // last tile: no GEMM1 to carry the tail of its elementwise stage
          static_for<16>([&](auto gc) { ratio_gap(std::false_type{}, gc, std::integral_constant<int, 1>{}); });
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(sp2_younger<N1, N2, PF>(n, last)));
        if constexpr (e.g1) {
          // ---- Q'(it): GEMM1 of tile i0+it+1; the tail of the elementwise stage of tile i0+it
          mfma_g1(K{}, std::integral_constant<int, (e.it + 1) & 1>{}, RS{});
          if constexpr (n + PF < LEN) issue_entry(std::integral_constant<int, n + PF>{}, lastc);
          if constexpr (e.k == 0) {
            dsrc = panel_src(i0 + e.it + 3);
            xs = x_src(i0 + e.it + 2);
          }
          ratio_gap(std::false_type{}, K{}, std::integral_constant<int, e.it & 1>{});
          if constexpr (e.k == N1 - 1) {
            // Everything P'(it-1) issued -- X(i0+it+1), whose elementwise stage starts below, and the panel pieces of tile
            // i0+it+2 -- has landed, and after the barrier every wave's pieces are visible: the first reads of that slot are
            // the operand prefetch in the tail of P'(it), and the LDS-DMA of P'(it) overwrites the slot every wave finished
            // reading in P'(it-1).  vmcnt(0), not a counted wait: the first build left "the 4 youngest" (the LDS-DMA pieces) in
            // flight behind the X loads -- and ran non-deterministic: LDS-DMA pieces (L2 hits) complete BEFORE older register
            // loads (HBM), so a count says nothing about which loads are still out (tools/sp_bitcompare.py against itself;
            // the drain costs nothing measurable).  The barrier sits HERE and not behind P': the prefetch for Q'(it+1) is
            // issued four entries before P' ends.
            u32x4(&xn)[4] = xb[(e.it + 1) & 1];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3])::"memory");
            barrier();
          }
        } else {
          // ---- P'(it): GEMM2 of tile i0+it, numerator and denominator from the same panel fragment; head of the
          // elementwise stage of tile i0+it+1; X loads of tile i0+it+2, panel pieces of tile i0+it+3
          using GapN = std::integral_constant<int, 2 * e.k>;        // the MFMA gap behind the numerator MFMA ...
          using GapD = std::integral_constant<int, 2 * e.k + 1>;    // ... and behind the denominator MFMA
          using SBN = std::integral_constant<int, (e.it + 1) & 1>;
          mfma_g2(K{}, RS{}, std::integral_constant<int, 0>{});
          if constexpr (!final_it) {
            if constexpr (e.k < 2) load_x1(xs, xb[e.it & 1], std::integral_constant<int, 2 * e.k>{});
            if constexpr (e.k >= 2 && e.k < 2 + C::NPW)
              dma_piece(dsrc, std::integral_constant<int, e.k - 2>{}, std::integral_constant<int, (e.it + 3) & 3>{});
            ratio_gap(std::true_type{}, GapN{}, SBN{});
          }
          mfma_g2(K{}, RS{}, std::integral_constant<int, 1>{});
          if constexpr (n + PF < LEN) issue_entry(std::integral_constant<int, n + PF>{}, lastc);
          if constexpr (!final_it) {
            if constexpr (e.k < 2) load_x1(xs, xb[e.it & 1], std::integral_constant<int, 2 * e.k + 1>{});
            ratio_gap(std::true_type{}, GapD{}, SBN{});
          }
        }
      });
    };

    // ---- prologue: panel tiles 0 .. 2, X(0), X(1); GEMM1(0) -> S[0]; the head of tile 0's elementwise stage on its own
    {
      static_for<3>([&](auto tc) {
        const char* src = panel_src(decltype(tc)::value);
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
      static_for<KS>([&](auto kc) {
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
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // XDL write -> VALU read of S
      static_for<32>([&](auto gc) { ratio_gap(std::true_type{}, gc, Z{}); });
    }
    int i0 = 0;
    for (; i0 + 4 < nt; i0 += 4) group(i0, std::false_type{});
    group(i0, std::true_type{});
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)"
                 : "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(xb[0][2]), "+v"(xb[0][3]), "+v"(xb[1][0]), "+v"(xb[1][1]), "+v"(xb[1][2]),
                   "+v"(xb[1][3])::"memory");
  }
  __syncthreads();               // LDS is reused by the epilogue

  // ---------------- epilogue: the four-wave kernel's (nmfmu_fused.h), both accumulator sets
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, j_e = lane_e & 31, hl_e = lane_e >> 5;
  const int mrow0 = mb * C::BM + wave * 32;   // first owner row of this wave
  const float unsc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((127 - ki) << 23));   // 2^-ki, exact
  // accumulator register e of lane (j_e, hl_e): row (e & 3) + 8 (e >> 2) + 4 hl_e, column 32 rt + j_e
  if (a.fuse_apply) {
    // ---- nmf.py:78-92 in the epilogue (nsplit == 1): relu(num) + eps over relu(den) + eps (+ l1 + l2 f), ^gamma
    constexpr int LDT = R_PAD;
    float* tile = reinterpret_cast<float*>(smem) + wave * (32 * LDT);
    float csum[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) csum[rt] = 0.f;
    static_for<RT>([&](auto rtc) {
      constexpr int rt = decltype(rtc)::value;
      const int r = rt * 32 + j_e;
      float fold[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mrow0 + (e & 3) + 8 * (e >> 2) + 4 * hl_e;
        fold[e] = (row < a.M && r < a.rank) ? a.f[(size_t)row * a.rank + r] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mrow0 + (e & 3) + 8 * (e >> 2) + 4 * hl_e;
        float fv = fold[e];
        if (row < a.M && r < a.rank) {
          const float neg = fmaxf(C::SCALE ? on[rt][e] * unsc : on[rt][e], 0.f) + kEps;
          float pos = fmaxf(C::SCALE ? op[rt][e] * unsc : op[rt][e], 0.f) + kEps;
          if (a.l1 > 0.f) pos += a.l1;
          if (a.l2 > 0.f) pos += a.l2 * fv;
          float mult = neg / pos;
          if (a.gamma != 1.f) mult = mu_pow(mult, a.gamma);
          fv *= mult;
          a.f[(size_t)row * a.rank + r] = fv;
        }
        fold[e] = fv;
        csum[rt] += fv;
        tile[((e & 3) + 8 * (e >> 2) + 4 * hl_e) * LDT + r] = fv;
      }
      // transposed image: 4 consecutive owner rows of column r = 8 bytes
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const uint32_t h0 = pack_op<OPT>(fold[4 * q4], fold[4 * q4 + 1]), h1 = pack_op<OPT>(fold[4 * q4 + 2], fold[4 * q4 + 3]);
        const int64_t off = p2_offset(mrow0 + 8 * q4 + 4 * hl_e, r, R_PAD);
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.o2_hi) + off) = make_uint2(h0, h1);
      }
    });
    __syncthreads();
    // row-major image from the LDS tile: 32 rows x R_PAD/8 sixteen-byte slots per wave
    constexpr int SP = R_PAD / 8;
    bool clamped = false;
#pragma unroll
    for (int i = 0; i < (32 * SP) / 64; ++i) {
      const int chunk = i * 64 + lane_e, rl = chunk / SP, slot = chunk % SP;
      const float* src = tile + rl * LDT + slot * 8;
      u32x4 hi;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const float x0 = src[2 * qq], x1 = src[2 * qq + 1];
        clamped |= fmaxf(x0, x1) > 65504.f;
        hi[qq] = pack_op<OPT>(x0, x1);
      }
      *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o1_hi) + p1_offset(mrow0 + rl, slot * 8, R_PAD)) = hi;
    }
    if (a.status && __any(clamped) && lane_e == 0) atomicOr(a.status, 1u);
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [WAVES][R_PAD]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float tot = csum[rt] + __shfl_xor(csum[rt], 32, 64);
      if (hl_e == 0) red[wave * R_PAD + rt * 32 + j_e] = tot;
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
        const size_t idx = slab + (size_t)row * R_PAD + rt * 32 + j_e;
        a.slab_num[idx] = C::SCALE ? on[rt][e] * unsc : on[rt][e];
        a.slab_den[idx] = C::SCALE ? op[rt][e] * unsc : op[rt][e];
      }
    });
  }
}

template <int R_PAD, int OPT, int BETA>
int launch_sp2_one(const FusedArgs& a, int grid, hipStream_t s) {
  using C = SP2Cfg<R_PAD, OPT, BETA>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS budget");
  auto kern = sp2_kernel<R_PAD, OPT, BETA>;
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

// Host-side launcher (nmfmu_inst_sp.hip).  beta_kind = BetaKind (kIS, kSqrt, kSqrt3, kGen); fp16 operands and target; padded rank 128.
int launch_sp2(int r_pad, int beta_kind, const FusedArgs& a, int grid, hipStream_t s);

}  // namespace nmfmu
