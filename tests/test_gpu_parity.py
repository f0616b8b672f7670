"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Run on the MI355X box:  python -m pytest tests -x -q -m gpu

Tolerances (relative Frobenius error unless stated):
  * bf16x3 (split-precision MFMA, the default): 1e-4 on factors, reconstruction and loss -- the bar
    BASELINE.json's north_star sets; measured errors are ~1e-6..1e-5.
  * f16 (fp16 operands and target, bf16's MFMA rate): 1e-4 as well wherever the contraction lengths are those of
    the BASELINE configs (the per-step operand rounding averages down with them); short contractions are held to
    what 11 significant bits give there, stated per test.  A target that fp16 does not hold exactly is rounded when
    packed; 'auto' therefore takes this mode only for fp16-exact targets.
  * bf16 (operands carry 8 significant bits): factors are compared at 5e-3 .. 2e-2 and the objective (loss) at 2e-3;
    this mode is NOT claimed to meet 1e-4 on the factors and nothing selects it implicitly.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, record, rel_err

_ORACLE_CACHE = {}   # (test, shape, beta) -> oracle factors, shared by the parametrizations that differ only in the operand mode

pytestmark = pytest.mark.gpu

TOL = 1e-4
NO_STOP = -1e9


def t(x):
    return torch.from_numpy(np.asarray(x))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from torchnmf_amd import _capi
    _capi.load()  # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ----------------------------------------------------------------------------------------------------------
# hardware assumptions
# ----------------------------------------------------------------------------------------------------------
def test_probe_mfma_lane_maps(dev):
    """v_mfma_f32_32x32x16_bf16 operand/result lane maps are the ones the fused kernel assumes."""
    from torchnmf_amd import _capi
    lib = _capi.load()
    g = torch.Generator().manual_seed(3)
    A = torch.randn(32, 16, generator=g).bfloat16()
    B = torch.randn(16, 32, generator=g).bfloat16()  # asymmetric on purpose
    d = torch.zeros(32, 32, device=dev)
    Ad, Bd = A.to(dev), B.to(dev)
    _capi.check(lib.nmfmu_probe_mfma(Ad.data_ptr(), Bd.data_ptr(), d.data_ptr(), _stream()), 'probe_mfma')
    torch.cuda.synchronize()
    want = A.float() @ B.float()
    assert rel_err(d.cpu(), want) < 1e-6


def test_probe_lds_dma_is_lane_linear(dev):
    from torchnmf_amd import _capi
    lib = _capi.load()
    n = 4096
    src = torch.arange(n, dtype=torch.int32, device=dev) * 7 + 1
    dst = torch.zeros(n, dtype=torch.int32, device=dev)
    _capi.check(lib.nmfmu_probe_lds_dma(src.data_ptr(), dst.data_ptr(), n, _stream()), 'probe_lds_dma')
    torch.cuda.synchronize()
    assert torch.equal(src.cpu(), dst.cpu())


# ----------------------------------------------------------------------------------------------------------
# packing layouts
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('prec', ['bf16', 'bf16x3', 'f16', 'f16x'])
@pytest.mark.parametrize('transpose', [False, True])
@pytest.mark.parametrize('block_rows', [128, 256])
def test_pack_x_layout_and_flags(dev, prec, transpose, block_rows):
    from test_layout_emulation import xp_index
    from torchnmf_amd import _capi
    from torchnmf_amd.engine import HipBackend
    be = HipBackend()
    g = torch.Generator().manual_seed(5)
    V = torch.rand(150, 200, generator=g).bfloat16().float()
    V[3, 7] = 0.0
    M, K = (200, 150) if transpose else (150, 200)
    m_pad, k_pad = be.pad_rows(M), be.pad_rows(K)
    flags = torch.tensor([0, 0x7f800000], dtype=torch.int32, device=dev)
    xp = be.pack_x(V.to(dev), transpose, _capi.PRECISIONS[prec], block_rows, m_pad, k_pad, flags)
    torch.cuda.synchronize()
    fp32 = prec in ('bf16x3', 'f16x')
    got = (xp.view(torch.float32) if fp32 else xp.view(torch.float16 if prec == 'f16' else torch.bfloat16).float()).cpu().numpy()
    X = (V.t() if transpose else V).numpy()
    want = np.zeros(m_pad * k_pad, dtype=np.float32)
    mm, kk = np.meshgrid(np.arange(M), np.arange(K), indexing='ij')
    idx = np.vectorize(lambda a, b: xp_index(int(a), int(b), k_pad // 64, fp32, block_rows // 128))(mm, kk)
    want[idx.reshape(-1)] = X.reshape(-1)
    np.testing.assert_array_equal(got, want)
    assert flags.tolist() == [0, 0]
    # a negative entry and a NaN must both raise the "bad" flag
    for badval in (-1.0, float('nan')):
        V2 = V.clone()
        V2[5, 5] = badval
        flags = torch.tensor([0, 0x7f800000], dtype=torch.int32, device=dev)
        be.pack_x(V2.to(dev), transpose, _capi.PRECISIONS[prec], block_rows, m_pad, k_pad, flags)
        assert flags.tolist()[0] == 1


@pytest.mark.parametrize('rank', [5, 16, 40, 100])
def test_pack_factor_images(dev, rank):
    from test_layout_emulation import p1_offset, p2_offset
    from torchnmf_amd import _capi
    from torchnmf_amd.engine import FactorBuf, HipBackend
    be = HipBackend()
    g = torch.Generator().manual_seed(rank)
    F = torch.rand(130, rank, generator=g)
    r_pad = be.pad_rank(rank)
    fb = FactorBuf(F.to(dev).contiguous(), r_pad, _capi.PREC_BF16X3, be)
    be.pack_factor(fb, rank, r_pad, _capi.PREC_BF16X3)
    torch.cuda.synchronize()
    hi = F.bfloat16().float()
    lo = (F - hi).bfloat16().float()
    p1h = fb.p1_hi.view(torch.bfloat16).float().cpu().numpy()
    p1l = fb.p1_lo.view(torch.bfloat16).float().cpu().numpy()
    p2h = fb.p2_hi.view(torch.bfloat16).float().cpu().numpy()
    p2l = fb.p2_lo.view(torch.bfloat16).float().cpu().numpy()
    w1h = np.zeros_like(p1h); w1l = np.zeros_like(p1h); w2h = np.zeros_like(p1h); w2l = np.zeros_like(p1h)
    for row in range(130):
        for r in range(rank):
            w1h[p1_offset(row, r, r_pad)] = hi[row, r]
            w1l[p1_offset(row, r, r_pad)] = lo[row, r]
            w2h[p2_offset(row, r, r_pad)] = hi[row, r]
            w2l[p2_offset(row, r, r_pad)] = lo[row, r]
    np.testing.assert_array_equal(p1h, w1h)
    np.testing.assert_array_equal(p1l, w1l)
    np.testing.assert_array_equal(p2h, w2h)
    np.testing.assert_array_equal(p2l, w2l)
    want_cs = torch.zeros(r_pad)
    want_cs[:rank] = F.sum(0)
    assert rel_err(fb.colsum.cpu(), want_cs) < 1e-6


# ----------------------------------------------------------------------------------------------------------
# single half-steps against the oracle (every beta branch, both precisions, both staging modes)
# ----------------------------------------------------------------------------------------------------------
def _one_iter(dev, V, W0, H0, beta, prec, stage, alpha=0.0, l1r=0.0, block_rows=None, allow_gram=False):
    from torchnmf_amd.engine import DenseMU
    W = W0.clone().to(dev).contiguous()
    H = H0.clone().to(dev).contiguous()
    eng = DenseMU(V.to(dev), W, H, beta, alpha * l1r, alpha * (1 - l1r), precision=prec, stage=stage,
                  block_rows=block_rows, allow_gram=allow_gram)
    # (the path without reconstruction exists for every single-plane mode except 'f16x' at padded rank 256)
    assert eng.gram_path == (allow_gram and beta == 2 and not (prec == 'f16x' and W0.shape[1] > 128))
    loss0 = eng.divergence()
    eng.w_step()
    torch.cuda.synchronize()
    W1 = W.cpu().clone()
    eng.h_step()
    torch.cuda.synchronize()
    return W1, H.cpu().clone(), loss0, eng.divergence()


@pytest.mark.parametrize('beta', [1, 2, 0, 0.5, 1.5, 3, -1])
@pytest.mark.parametrize('stage', [1])
def test_half_steps_bf16x3(dev, beta, stage):
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(11)
    N, C, R = 200, 330, 24   # ragged: not multiples of 128 / 64 / 32
    V = torch.rand(N, C, generator=g) + (1e-3 if beta <= 0 else 0)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, 'bf16x3', stage, alpha=0.1, l1r=0.5)
    gam = O.gamma_of(beta)
    Wr = O.nmf_w_step(V, W0, H0, beta, gam, 0.05, 0.05)
    Hr = O.nmf_h_step(V, Wr, H0, beta, gam, 0.05, 0.05)
    assert rel_err(W1, Wr) < TOL, rel_err(W1, Wr)
    assert rel_err(H1, Hr) < TOL, rel_err(H1, Hr)
    want0 = float(O.beta_div(O.nmf_reconstruct(H0, W0), V, beta))
    want1 = float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, beta))
    assert l0 == pytest.approx(want0, rel=TOL) and l1 == pytest.approx(want1, rel=TOL)


@pytest.mark.parametrize('beta', [1, 2, 0.5])
@pytest.mark.parametrize('stage', [1])
@pytest.mark.parametrize('block_rows', [128, None])   # None = the engine's choice (256-row tiles for beta == 1)
def test_half_steps_bf16(dev, beta, stage, block_rows):
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(12)
    N, C, R = 384, 1100, 64
    V = torch.rand(N, C, generator=g).bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, 'bf16', stage, block_rows=block_rows)
    gam = O.gamma_of(beta)
    Wr = O.nmf_w_step(V, W0, H0, beta, gam)
    Hr = O.nmf_h_step(V, Wr, H0, beta, gam)
    assert rel_err(W1, Wr) < 5e-3 and rel_err(H1, Hr) < 5e-3, (rel_err(W1, Wr), rel_err(H1, Hr))
    assert l0 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(H0, W0), V, beta)), rel=2e-3)


@pytest.mark.parametrize('rank', [5, 40, 100])
def test_pack_factor_images_f16(dev, rank):
    """fp16 images (precision 'f16'): same P1 / P2 layouts, values rounded to fp16 (RNE) and clamped to 65504."""
    from test_layout_emulation import p1_offset, p2_offset
    from torchnmf_amd import _capi
    from torchnmf_amd.engine import FactorBuf, HipBackend
    be = HipBackend()
    g = torch.Generator().manual_seed(rank)
    F = torch.rand(130, rank, generator=g)
    F[0, 0], F[1, 0], F[2, 0] = 1e5, 3e-6, 0.0        # clamps to 65504, fp16 subnormal, zero
    r_pad = be.pad_rank(rank)
    fb = FactorBuf(F.to(dev).contiguous(), r_pad, _capi.PREC_F16, be)
    be.pack_factor(fb, rank, r_pad, _capi.PREC_F16)
    torch.cuda.synchronize()
    want = F.clamp(max=65504.0).half().float()
    p1 = fb.p1_hi.view(torch.float16).float().cpu().numpy()
    p2 = fb.p2_hi.view(torch.float16).float().cpu().numpy()
    w1 = np.zeros_like(p1); w2 = np.zeros_like(p1)
    for row in range(130):
        for r in range(rank):
            w1[p1_offset(row, r, r_pad)] = want[row, r]
            w2[p2_offset(row, r, r_pad)] = want[row, r]
    np.testing.assert_array_equal(p1, w1)
    np.testing.assert_array_equal(p2, w2)


@pytest.mark.parametrize('shape', [(384, 1100, 64), (520, 2300, 100), (200, 330, 24), (300, 700, 128)])
@pytest.mark.parametrize('regs', [(0.0, 0.0), (0.05, 0.05)])
def test_half_steps_f16(dev, shape, regs):
    """precision='f16' (fp16 operands in the ping-pong kernel): one iteration against the fp32 oracle.  One rounding
    to 11 significant bits per operand: a few 1e-5 per half-step."""
    from oracle import mu_oracle as O
    N, C, R = shape
    g = torch.Generator().manual_seed(N + R)
    V = torch.rand(N, C, generator=g)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, 1, 'f16', 1, alpha=sum(regs), l1r=0.5)
    Wr = O.nmf_w_step(V, W0, H0, 1, 1.0, *regs)
    Hr = O.nmf_h_step(V, Wr, H0, 1, 1.0, *regs)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('half_steps_f16', shape=shape, regs=regs, relW=ew, relH=eh)
    assert ew < TOL and eh < TOL, (ew, eh)
    assert l0 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(H0, W0), V, 1)), rel=TOL)
    assert l1 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, 1)), rel=TOL)


@pytest.mark.parametrize('beta', [2, 0, 0.5, 1.5, 3, -1])
@pytest.mark.parametrize('shape', [(384, 1100, 64), (520, 2300, 128), (200, 330, 24), (300, 700, 200)])
def test_half_steps_f16_every_beta(dev, beta, shape):
    """precision='f16' on the four-wave kernel (beta != 1: two accumulator sets; padded rank 256): one iteration against
    the fp32 oracle, regularised.  For beta < 1 the elementwise terms are negative powers of S; the kernel scales them
    by a power of two taken from the factors' column sums, so that they stay in fp16's normal range."""
    from oracle import mu_oracle as O
    N, C, R = shape
    g = torch.Generator().manual_seed(N + R + int(10 * beta))
    V = torch.rand(N, C, generator=g) + (2.0 ** -7 if beta <= 0 else 0)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, 'f16', 1, alpha=0.1, l1r=0.5)
    gam = O.gamma_of(beta)
    Wr = O.nmf_w_step(V, W0, H0, beta, gam, 0.05, 0.05)
    Hr = O.nmf_h_step(V, Wr, H0, beta, gam, 0.05, 0.05)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('half_steps_f16_every_beta', beta=beta, shape=shape, relW=ew, relH=eh)
    assert ew < TOL and eh < TOL, (ew, eh)
    assert l0 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(H0, W0), V, beta)), rel=2e-4)
    assert l1 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, beta)), rel=2e-4)


@pytest.mark.parametrize('beta', [0, 0.5])
@pytest.mark.parametrize('scale', [1e-3, 30.0])
def test_f16_scaled_terms_small_and_large_reconstructions(dev, beta, scale):
    """beta < 1 in fp16: with S ~ 1e3 the terms S^(beta-2) V sit ~1e-6 (fp16-subnormal without the kernel's power-of-two
    scale), with S ~ 1e-4 they overflow 65504 without it.  Both must stay at the single-rounding error level."""
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(int(scale * 1000) + int(10 * beta))
    N, C, R = 520, 1300, 64
    V = torch.rand(N, C, generator=g) + 2.0 ** -7
    W0 = torch.randn(C, R, generator=g).abs() * scale
    H0 = torch.randn(N, R, generator=g).abs() * scale
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, 'f16', 1)
    gam = O.gamma_of(beta)
    Wr = O.nmf_w_step(V, W0, H0, beta, gam)
    Hr = O.nmf_h_step(V, Wr, H0, beta, gam)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('f16_scaled_terms', beta=beta, scale=scale, relW=ew, relH=eh)
    assert torch.isfinite(W1).all() and torch.isfinite(H1).all()
    assert ew < 2e-4 and eh < 2e-4, (ew, eh)


@pytest.mark.parametrize('beta', [1, 2, 0, 0.5, 1.5, 3, -1])
@pytest.mark.parametrize('shape', [(384, 1100, 64), (520, 2300, 128), (200, 330, 24), (300, 700, 200)])
def test_half_steps_f16x_every_beta(dev, beta, shape):
    """precision='f16x' (round 4: fp16 operands, the target stays fp32 -- four-wave kernel for every beta, padded rank up
    to 256): one regularised iteration against the fp32 oracle on a target fp16 does NOT hold exactly (plain floats).
    For beta = 2 the target is the second GEMM's operand and goes in as an fp16 hi + lo pair."""
    from oracle import mu_oracle as O
    N, C, R = shape
    g = torch.Generator().manual_seed(N + R + int(10 * beta) + 7)
    V = torch.rand(N, C, generator=g) + (2.0 ** -7 if beta <= 0 else 0)
    assert not torch.equal(V.half().float(), V)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, 'f16x', 1, alpha=0.1, l1r=0.5)
    gam = O.gamma_of(beta)
    Wr = O.nmf_w_step(V, W0, H0, beta, gam, 0.05, 0.05)
    Hr = O.nmf_h_step(V, Wr, H0, beta, gam, 0.05, 0.05)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('half_steps_f16x_every_beta', beta=beta, shape=shape, relW=ew, relH=eh)
    assert ew < TOL and eh < TOL, (ew, eh)
    assert l0 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(H0, W0), V, beta)), rel=2e-4)
    assert l1 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, beta)), rel=2e-4)


def test_f16x_keeps_the_target_unrounded(dev):
    """The point of 'f16x': a target made of values that fp16 rounds badly (x = 1 + 2^-13 multiples) -- the 'f16' mode
    sees V rounded to 11 bits, 'f16x' must not: one beta = 2 half-step, whose numerator V^T H is linear in V, is compared
    with the oracle on V and on fp16(V)."""
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(77)
    N, C, R = 384, 1100, 64
    V = 1.0 + torch.randint(0, 8, (N, C), generator=g).float() * 2.0 ** -13     # fp16 spacing at 1.0 is 2^-10
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs().half().float()                  # operands exact in fp16: isolates the target
    W0 = W0.half().float()
    W1, _, _, _ = _one_iter(dev, V, W0, H0, 2, 'f16x', 1)
    Wr = O.nmf_w_step(V, W0, H0, 2, 1.0)
    Wq = O.nmf_w_step(V.half().float(), W0, H0, 2, 1.0)
    e_true, e_rounded = rel_err(W1, Wr), rel_err(W1, Wq)
    record('f16x_unrounded_target', e_true=e_true, e_vs_rounded_target=e_rounded)
    assert e_true < 2e-5 and e_rounded > 3 * e_true, (e_true, e_rounded)


@pytest.mark.parametrize('prec,tol', [('bf16', 5e-3), ('f16', TOL), ('f16x', TOL)])
@pytest.mark.parametrize('shape', [(384, 1100, 64), (520, 2300, 128), (200, 330, 24), (300, 700, 200), (3000, 260, 100)])
@pytest.mark.parametrize('regs', [(0.0, 0.0), (0.05, 0.05)])
def test_half_steps_beta2_without_reconstruction(dev, prec, tol, shape, regs):
    """Round 4, the beta == 2 path of fit(): numerator = X @ panel (one streaming MFMA GEMM, kModeXB), denominator =
    owner @ (panel^T panel) through the MFMA Gram kernel -- in the kernel's epilogue when the contraction is not split
    (rank pad <= 128), in the apply kernel otherwise (split contraction, rank pad 256).  One iteration against the fp32
    oracle, which follows the reference's own formulation (reconstruction + two backward products, nmf.py:61-63, 77-83)."""
    from oracle import mu_oracle as O
    N, C, R = shape
    g = torch.Generator().manual_seed(N + R + 2)
    V = torch.rand(N, C, generator=g)
    if prec != 'f16x':
        V = V.bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, 2, prec, 1, alpha=sum(regs), l1r=0.5, allow_gram=True)
    Wr = O.nmf_w_step(V, W0, H0, 2, 1.0, *regs)
    Hr = O.nmf_h_step(V, Wr, H0, 2, 1.0, *regs)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('half_steps_beta2_gram', prec=prec, shape=shape, regs=regs, relW=ew, relH=eh)
    assert ew < tol and eh < tol, (ew, eh)
    assert l1 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, 2)), rel=20 * tol)


def test_gram_panel_matches_fp32(dev):
    """nmfmu_gram_panel: G = F^T F from the 16-bit transposed image (MFMA, deterministic two-stage sum): the fp32 matrix
    against torch on the image's own (rounded) values, the hi + lo images times their row scales against the fp32 matrix."""
    from torchnmf_amd import _capi
    from torchnmf_amd.engine import FactorBuf, HipBackend
    be = HipBackend()
    g = torch.Generator().manual_seed(4)
    for rows, rank, prec in [(5000, 100, 'f16'), (700, 128, 'bf16'), (70000, 40, 'f16'), (300, 200, 'f16')]:
        F = (torch.randn(rows, rank, generator=g).abs() * 3).to(dev)
        r_pad, P = be.pad_rank(rank), _capi.PRECISIONS[prec]
        fb = FactorBuf(F, r_pad, P, be)
        be.pack_factor(fb, rank, r_pad, P)
        gm = be.gram_alloc(r_pad, dev)
        be.gram_panel(fb, r_pad, P, gm)
        be.gram_panel(fb, r_pad, P, gm)                  # deterministic: a second run must reproduce the bits
        torch.cuda.synchronize()
        _, gram, hi, lo, scale = gm
        Fq = (F.half() if prec == 'f16' else F.bfloat16()).double()
        want = torch.zeros(r_pad, r_pad, dtype=torch.float64, device=dev)
        want[:rank, :rank] = Fq.t() @ Fq
        G = gram.view(r_pad, r_pad)
        assert float((G.double() - want).norm() / want.norm()) < 2e-6
        assert torch.equal(G, G.t().contiguous()) or float((G - G.t()).abs().max() / G.abs().max()) < 1e-6
        dt = torch.float16 if prec == 'f16' else torch.bfloat16
        img = (hi.view(dt).float() + lo.view(dt).float()).view(r_pad, r_pad) * scale.view(r_pad, 1)
        assert float((img - G).norm() / G.norm()) < (2e-6 if prec == 'f16' else 3e-5)
        g2 = gram.clone()
        be.gram_panel(fb, r_pad, P, gm)
        torch.cuda.synchronize()
        assert torch.equal(g2, gram)


def test_register_staging_is_gone(dev):
    """NMFMU_STAGE_REG (ABI < 4) is no longer built: the library says so instead of silently taking the DMA path."""
    from torchnmf_amd.engine import DenseMU
    g = torch.Generator().manual_seed(1)
    V, W, H = torch.rand(64, 80, generator=g), torch.rand(80, 8, generator=g), torch.rand(64, 8, generator=g)
    with pytest.raises(NotImplementedError):
        DenseMU(V.to(dev), W.to(dev), H.to(dev), 1.0, precision='bf16x3', stage=0).divergence()


def test_f16_range_handling(dev):
    """fp16 operands: values below the fp16 normal range degrade gracefully (subnormals), ratios above 65504 saturate
    (MODE.FP16_OVFL) instead of turning into inf / NaN, exact zeros stay exact."""
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(3)
    N, C, R = 300, 520, 32
    V = torch.rand(N, C, generator=g)
    V[:, :40] *= 1e-6                      # a block of tiny targets (fp16 subnormal / flushed)
    V[5, :] = 0.0
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    H0[7, :] = 1e-7                        # S ~ 1e-6 on that row while V ~ 0.5: ratio ~ 5e5 > 65504
    W0[10, :] = 0.0                        # an exactly-zero factor row stays exactly zero
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, 1, 'f16', 1)
    assert torch.isfinite(W1).all() and torch.isfinite(H1).all()
    assert float(W1[10].abs().max()) == 0.0
    Wr = O.nmf_w_step(V, W0, H0, 1, 1.0)
    Hr = O.nmf_h_step(V, Wr, H0, 1, 1.0)
    keep = torch.ones(N, dtype=torch.bool); keep[7] = False
    # away from the saturating row the update is the reference's (the clamped ratios of row 7 and the flushed tiny
    # targets move W by ~1e-3, which H inherits)
    assert rel_err(H1[keep], Hr[keep]) < 5e-3, rel_err(H1[keep], Hr[keep])


def test_f16_factor_leaving_the_range_is_reported(dev):
    """the fp16 range gate looks at the initial data only.  A factor that grows beyond 65504 during the fit is
    clamped in its fp16 image; every kernel that clamps sets bit 0 of nmfmu_step.status, fit() reads it at its loss
    checkpoints and warns (the update no longer follows the reference from there on)."""
    import warnings
    from torchnmf_amd.engine import DenseMU
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(4)
    N, C, R = 300, 520, 8
    V = (torch.rand(N, C, generator=g) * 2.0e4).half().float()
    W0 = torch.rand(C, R, generator=g) + 0.5
    H0 = (torch.rand(N, R, generator=g) + 0.5) * 1e-3          # W <- W * (V / (H W^T)) H / sum(H)  ~ 2e4 / 1e-3: far outside
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = DenseMU(V.to(dev), W, H, 1.0, precision='f16')
    assert not eng.left_f16_range()
    eng.w_step()
    torch.cuda.synchronize()
    assert float(W.max()) > 65504.0 and eng.left_f16_range()
    # in range: the flag stays clear
    W2, H2 = W0.clone().to(dev), (H0 * 1e3).to(dev)
    eng2 = DenseMU((V / 2.0e4).to(dev), W2, H2, 1.0, precision='f16')
    eng2.w_step(); eng2.h_step()
    torch.cuda.synchronize()
    assert not eng2.left_f16_range()
    m = NMF(W=W0, H=H0).to(dev)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        m.fit(V.to(dev), 1, NO_STOP, 10, precision='f16')
    assert any("fp16's range" in str(w.message) for w in rec), [str(w.message) for w in rec]


def test_nmfd_f16_factor_leaving_the_range_is_reported(dev):
    """The same report for NMFD in 'f16' (its planes and window tables clamp at 65504): fit() warns at a loss checkpoint."""
    import warnings
    from torchnmf_amd.nmf import NMFD
    from torchnmf_amd.nmfd_engine import ConvMU
    g = torch.Generator().manual_seed(6)
    B, Cc, L, R, T = 1, 24, 328, 2, 136
    V = torch.rand(B, Cc, L, generator=g) * 2.0e4
    W0 = torch.rand(Cc, R, T, generator=g) + 0.5
    H0 = (torch.rand(B, R, L - T + 1, generator=g) + 0.5) * 1e-3     # W <- W * (V / S (*) H) / sum(H): far beyond 65504
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = ConvMU(V.to(dev), W, H, 1, precision='f16')
    assert not eng.left_f16_range()
    eng.w_step()
    assert float(W.max()) > 65504.0 and eng.left_f16_range()
    W2, H2 = W0.clone().to(dev), (H0 * 1e3).to(dev)
    eng2 = ConvMU((V / 2.0e4).to(dev), W2, H2, 1, precision='f16')
    eng2.w_step(); eng2.h_step()
    assert not eng2.left_f16_range()
    m = NMFD(W=W0, H=H0).to(dev)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        m.fit(V.to(dev), 1, NO_STOP, 10, precision='f16')
    assert any("fp16's range" in str(w.message) for w in rec), [str(w.message) for w in rec]


def test_fit_f16_meets_parity_bar(dev):
    """north_star's bar (1e-4 relative on the factors after N iterations) in the single-plane fp16 mode at a size where
    the per-step rounding errors average down: 2048 x 4096, rank 64, 50 iterations."""
    from torchnmf_amd.nmf import NMF
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(21)
    N, C, R = 2048, 4096, 64
    V = torch.rand(N, C, generator=g).bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    m = NMF(W=W0, H=H0).to(dev)
    n = m.fit(V.to(dev), 1, NO_STOP, 50, precision='f16')
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 50)
    ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
    print(f'f16 fit 2048x4096 r64, 50 iterations: relW={ew:.2e} relH={eh:.2e}')
    assert n == nr == 50
    assert ew < TOL and eh < TOL, (ew, eh)


def test_auto_f16_at_its_threshold_200_iterations(dev):
    """What precision='auto' promises where it takes the fp16 mode: the smallest shape it admits
    (4096 x 4096), a target that fp16 holds exactly, the default max_iter = 200 -- factors within 1e-4 of the
    reference's fp32 iteration (oracle.aten_port: the reference's own op sequence)."""
    from oracle import aten_port
    from torchnmf_amd.engine import DenseMU
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(33)
    N, C, R = DenseMU.F16_MIN_DIM, DenseMU.F16_MIN_DIM, 64
    V = torch.rand(N, C, generator=g).half().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    m = NMF(W=W0, H=H0).to(dev)
    Vd = V.to(dev)
    assert DenseMU(Vd, m.W.data.clone(), m.H.data.clone(), 1.0, precision='auto', allow_f16=True).precision_name == 'f16'
    n = m.fit(Vd, 1, NO_STOP, 200)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    Wr, Hr = aten_port.mu_iterations(V, W0, H0, 1, 200)
    ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
    record('auto_f16_threshold_200', shape=(N, C, R), relW=ew, relH=eh)
    assert n == 200 and ew < TOL and eh < TOL, (ew, eh)


def test_auto_f16x_real_data_200_iterations(dev):
    """what precision='auto' promises on data fp16 does not hold exactly (plain U[0,1) floats): the
    smallest shape it admits, the default max_iter = 200, through fit() -- 'f16x' (fp16 operands, fp32 target), factors
    within 1e-4 of the reference's fp32 iteration.  (The 'f16' mode drifts to 3.4e-4 on such a target.)"""
    from oracle import aten_port
    from torchnmf_amd import engine
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(34)
    N, C, R = engine.DenseMU.F16_MIN_DIM, engine.DenseMU.F16_MIN_DIM, 64
    V = torch.rand(N, C, generator=g)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    m = NMF(W=W0, H=H0).to(dev)
    Vd = V.to(dev)
    # (round 6: the 3-byte form of the unrounded target, 'f16r'; 'f16x' through round 5)
    assert engine.DenseMU(Vd, m.W.data.clone(), m.H.data.clone(), 1.0, precision='auto', allow_f16=True).precision_name == 'f16r'
    n = m.fit(Vd, 1, NO_STOP, 200)
    assert m.last_precision == 'f16r'
    torch.set_num_threads(min(16, torch.get_num_threads()))
    Wr, Hr = aten_port.mu_iterations(V, W0, H0, 1, 200)
    ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
    record('auto_f16x_real_data_200', shape=(N, C, R), relW=ew, relH=eh)
    assert n == 200 and ew < TOL and eh < TOL, (ew, eh)


@pytest.mark.parametrize('cols,nsplit', [(256, 1), (320, 1), (576, 3), (1100, 2), (2300, 8), (4100, 3)])
@pytest.mark.parametrize('regs', [(0.0, 0.0), (0.05, 0.05)])
def test_rank128_bf16_kl_tile_variants(dev, cols, nsplit, regs, monkeypatch):
    """The beta == 1 / bf16 / rank-pad-128 kernel in both workgroup shapes (128-row tiles; 256-row tiles = the
    eight-wave software-pipelined loop when the library is built with it).  The contraction split is forced so that
    workgroups get 1, 2, 3, ... tiles (prologue / odd-even tail paths of the pipelined loop), and the W half-step runs
    both with the apply fused in the epilogue (nsplit 1) and through slabs."""
    from oracle import mu_oracle as O
    from torchnmf_amd import engine
    g = torch.Generator().manual_seed(cols + nsplit)
    N, C, R = 520, cols, 100
    V = torch.rand(N, C, generator=g).bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    monkeypatch.setattr(engine.HipBackend, 'choose_nsplit', lambda self, m, k, br, d: min(nsplit, max(1, k // 64)))
    out = {}
    for br in (128, 256):
        out[br] = _one_iter(dev, V, W0, H0, 1, 'bf16', 1, alpha=sum(regs), l1r=0.5, block_rows=br)
    Wr = O.nmf_w_step(V, W0, H0, 1, 1.0, *regs)
    Hr = O.nmf_h_step(V, Wr, H0, 1, 1.0, *regs)
    for br in (128, 256):
        W1, H1, l0, l1 = out[br]
        assert rel_err(W1, Wr) < 5e-3 and rel_err(H1, Hr) < 5e-3, (br, rel_err(W1, Wr), rel_err(H1, Hr))
        assert l0 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(H0, W0), V, 1)), rel=2e-3)
    # same operands, same products: the two tile shapes differ only in summation order
    assert rel_err(out[256][0], out[128][0]) < 2e-5 and rel_err(out[256][1], out[128][1]) < 2e-5


@pytest.mark.parametrize('shape', [(128, 64, 32), (1, 1, 1), (129, 65, 33), (700, 5000, 128), (3000, 260, 100)])
def test_shapes_and_ksplit(dev, shape):
    """Ragged and degenerate sizes; 5000 columns forces a contraction split (several slabs per owner block)."""
    from oracle import mu_oracle as O
    N, C, R = shape
    g = torch.Generator().manual_seed(N + C)
    V = torch.rand(N, C, generator=g)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W1, H1, _, l1 = _one_iter(dev, V, W0, H0, 1, 'bf16x3', 1)
    Wr = O.nmf_w_step(V, W0, H0, 1, 1.0)
    Hr = O.nmf_h_step(V, Wr, H0, 1, 1.0)
    assert rel_err(W1, Wr) < TOL and rel_err(H1, Hr) < TOL
    assert l1 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, 1)), rel=TOL, abs=1e-6)


# ----------------------------------------------------------------------------------------------------------
# fit() through the module surface against the golden vectors the reference produced
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
@pytest.mark.parametrize('reg', [(0, 0), (0.1, 0), (0.1, 0.5), (0.1, 1.0)])
def test_fit_g1_golden(dev, beta, reg):
    from torchnmf_amd.nmf import NMF
    g = load_golden('g1_nmf_small')
    alpha, l1r = reg
    V = t(g['V']) + (float(g['v_shift_nonpos_beta']) if beta <= 0 else 0.0)
    tag = f'b{beta}_a{alpha}_l{l1r}'
    m = NMF(W=t(g['W0']), H=t(g['H0'])).to(dev)
    n = m.fit(V.to(dev), beta, NO_STOP, 50, False, alpha, l1r, precision='bf16x3')
    assert n == 50
    assert rel_err(m.W.data.cpu(), g[f'{tag}_W50']) < TOL, rel_err(m.W.data.cpu(), g[f'{tag}_W50'])
    assert rel_err(m.H.data.cpu(), g[f'{tag}_H50']) < TOL
    assert rel_err(m().cpu(), t(g[f'{tag}_H50']) @ t(g[f'{tag}_W50']).t()) < TOL


def test_fit_g2_cfg1(dev):
    """BASELINE configs[0]: NMF 256x512 rank 16 beta=1, 50 iterations -- both precision modes."""
    from torchnmf_amd.nmf import NMF
    from oracle import mu_oracle as O
    g = load_golden('g2_cfg1')
    V = t(g['V_bf16_bits']).view(torch.bfloat16).float()
    for prec, wtol, ltol in (('bf16x3', TOL, TOL), ('bf16', 2e-2, 2e-3)):
        m = NMF(W=t(g['W0']), H=t(g['H0'])).to(dev)
        n = m.fit(V.to(dev), 1, NO_STOP, 50, precision=prec)
        assert n == 50
        ew, eh = rel_err(m.W.data.cpu(), g['W50']), rel_err(m.H.data.cpu(), g['H50'])
        loss = O.fit_loss(O.nmf_reconstruct(m.H.data.cpu(), m.W.data.cpu()), V, 1)
        print(f'cfg1 {prec}: relW={ew:.2e} relH={eh:.2e} loss={loss:.6f} ref={float(g["losses50"][-1]):.6f}')
        assert ew < wtol and eh < wtol
        assert loss == pytest.approx(float(g['losses50'][-1]), rel=ltol)


@pytest.mark.parametrize('beta', [0.5, 1, 2])
def test_fit_g3_early_stop(dev, beta):
    from torchnmf_amd.nmf import NMF
    g = load_golden('g3_early_stop')
    m = NMF(W=t(g['W0']), H=t(g['H0'])).to(dev)
    n = m.fit(t(g['V']).to(dev), beta, 1e-4, 200, precision='bf16x3')
    assert n == int(g[f'b{beta}_n_iter'])
    assert rel_err(m.W.data.cpu(), g[f'b{beta}_W']) < TOL and rel_err(m.H.data.cpu(), g[f'b{beta}_H']) < TOL


@pytest.mark.parametrize('prec,beta,shape', [('f16', 1, (512, 768, 20)), ('bf16x3', 0.5, (300, 520, 12)), ('f16r', 1, (301, 333, 5)),
                                             ('f16', 2, (512, 768, 20))])
def test_fused_loss_checkpoint_equals_the_generic_sequence(dev, monkeypatch, prec, beta, shape):
    """Round 6: a loss checkpoint of fit() is the loss kernel + ONE launch (nmfmu_loss_checkpoint: fixed-order sum of the
    partials, fp16-range flag, snapshots of both factors) instead of the loss's two launches, five small torch launches and
    two copies.  Same loss bit for bit, same flag, same snapshots, same rollback; factors whose size is not a multiple of
    four floats take the generic sequence (third case); fit() returns the same count and factors either way."""
    from torchnmf_amd.engine import DenseMU
    from torchnmf_amd.nmf import NMF
    N, C, R = shape
    g = torch.Generator().manual_seed(N + R)
    V = (torch.rand(N, C, generator=g) + 0.01).to(dev)
    W0, H0 = torch.rand(C, R, generator=g) + 0.1, torch.rand(N, R, generator=g) + 0.1
    out = {}
    monkeypatch.setenv('TORCHNMF_AMD_RIDING_LOSS', '0')     # (beta == 1 on the ping-pong kernel would not make a loss pass at all)
    for mode in ('1', '0'):
        monkeypatch.setenv('TORCHNMF_AMD_FUSED_CHECKPOINT', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = DenseMU(V, W, H, float(beta), precision=prec, allow_gram=(beta == 2))
        took = []
        orig = eng._checkpoint_fused
        eng._checkpoint_fused = lambda *a: took.append(orig(*a)) or took[-1]
        for _ in range(3):
            eng.w_step(), eng.h_step()
        eng.checkpoint_begin()
        Wk, Hk = W.clone(), H.clone()
        for _ in range(2):
            eng.w_step(), eng.h_step()
        div, left = eng.checkpoint_result()
        sync = eng.divergence()
        eng.rollback()
        torch.cuda.synchronize()
        assert torch.equal(W, Wk) and torch.equal(H, Hk)
        assert took == [mode == '1' and (C * R) % 4 == 0 and (N * R) % 4 == 0]
        assert eng.divergence() == div and sync < div and not left
        out[mode] = (div, W.cpu().clone(), H.cpu().clone())
    assert out['1'][0] == out['0'][0] and torch.equal(out['1'][1], out['0'][1]) and torch.equal(out['1'][2], out['0'][2])
    fits = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('TORCHNMF_AMD_FUSED_CHECKPOINT', mode)
        m = NMF(W=W0.clone(), H=H0.clone()).to(dev)
        fits[mode] = (m.fit(V, beta, 2e-3, 120, precision=prec), m.W.data.cpu().clone(), m.H.data.cpu().clone())
    assert fits['1'][0] == fits['0'][0] and fits['1'][0] < 120
    assert torch.equal(fits['1'][1], fits['0'][1]) and torch.equal(fits['1'][2], fits['0'][2])


def test_rank256_kl_does_not_keep_the_transposed_images(dev, monkeypatch):
    """Round 6 (NMFMU_STAGE_DMA_NOP2): beta == 1 at padded rank 256 with fp16 operands runs the software-pipelined kernel in
    both half-steps and the four-wave loss kernel -- one image per factor, GEMM2's operand gathered from it.  Nothing reads the
    transposed images, so the engine stops refreshing them (5.8 of the fused-apply epilogue's 21 us).  Proof by poison: with both
    transposed images overwritten by NaN bit patterns after packing, three iterations and the loss give bit-identical results to
    an engine that keeps them (TORCHNMF_AMD_NO_P2=0); the poisoned images are still poison afterwards, the kept ones are not.
    Both the unsplit (fused apply) and the split (apply kernel) W half-step."""
    from torchnmf_amd import _capi
    from torchnmf_amd.engine import DenseMU
    for N, C, R in ((300, 2300, 200), (2048, 1200, 256)):
        g = torch.Generator().manual_seed(N + R)
        V = torch.rand(N, C, generator=g).half().float().to(dev)
        W0, H0 = torch.rand(C, R, generator=g) + 0.1, torch.rand(N, R, generator=g) + 0.1
        out = {}
        for mode in ('1', '0'):
            monkeypatch.setenv('TORCHNMF_AMD_NO_P2', mode)
            W, H = W0.clone().to(dev), H0.clone().to(dev)
            eng = DenseMU(V, W, H, 1.0, precision='f16')
            assert (eng.step_h.struct.stage == _capi.STAGE_DMA_NOP2) == (mode == '1')
            for fac in (eng.fW, eng.fH):
                fac.p2_hi.view(torch.int16).fill_(0x7e00)          # fp16 NaN in every slot
            for _ in range(3):
                eng.w_step(), eng.h_step()
            loss = eng.divergence()
            poisoned = [bool((fac.p2_hi.view(torch.int16) == 0x7e00).all()) for fac in (eng.fW, eng.fH)]
            assert poisoned == [mode == '1'] * 2, (mode, poisoned)
            out[mode] = (W.cpu().clone(), H.cpu().clone(), loss)
            assert bool(torch.isfinite(W).all()) and bool(torch.isfinite(H).all())
        assert torch.equal(out['1'][0], out['0'][0]) and torch.equal(out['1'][1], out['0'][1]) and out['1'][2] == out['0'][2]


def test_target_stats_one_pass_matches_the_torch_passes(dev):
    """Round 6: the admission test of precision='auto' asks its questions about V (max, fp16-exactness, mean) with ONE kernel
    pass (nmfmu_target_sums) instead of ~7 GB of torch temporaries at configs[1]; the same pass leaves the two sums the riding
    loss needs.  Same answers as the torch passes on exact / inexact / out-of-range / strided targets; sums against float64."""
    from torchnmf_amd.engine import DenseMU, DEFAULT_BACKEND_FACTORY, target_stats, TARGET_STATS
    be = DEFAULT_BACKEND_FACTORY()
    g = torch.Generator().manual_seed(3)
    W, H = (torch.rand(500, 8, generator=g) + 0.1).to(dev), (torch.rand(300, 8, generator=g) + 0.1).to(dev)
    V = torch.rand(300, 500, generator=g)
    big = torch.rand(300, 700, generator=g)
    cases = {'plain': V, 'exact': V.half().float(), 'bf16': V.bfloat16().float(), 'counts': torch.randint(0, 2048, (300, 500), generator=g).float(),
             'beyond': V.half().float() * 2.0 ** 17, 'tiny': V.half().float() * 2.0 ** -20, 'one_bad': V.half().float().index_put((torch.tensor([7]), torch.tensor([9])), torch.tensor(0.1)),
             'strided': big[:, :500]}
    for name, v in cases.items():
        vd = v.to(dev) if name != 'strided' else big.to(dev)[:, :500]
        assert (vd.stride(0) == 700) == (name == 'strided')
        got = DenseMU.f16_stats(vd, W, H, be)
        want = DenseMU.f16_stats(vd, W, H)
        assert got == want, (name, got, want)
        out4 = target_stats(vd, be).cpu()
        v64 = v.double()
        assert float(out4[1]) == pytest.approx(float(v64.sum()), rel=1e-9)
        ref_a = float((v64 * (v.float() + torch.finfo(torch.float32).eps).double().log()).sum())      # (x + eps is an fp32 sum in the reference too)
        assert float(out4[0]) == pytest.approx(ref_a, rel=2e-6, abs=1e-3), (name, float(out4[0]), ref_a)
        assert float(out4[2]) == float(v.max())
    # the memo follows the version counter: an in-place edit of V is seen
    vd = cases['exact'].to(dev)
    assert DenseMU.f16_stats(vd, W, H, be)[1] is True and TARGET_STATS['ref']() is vd
    # ... and the tensor OBJECT: another tensor at the same address (caching allocator) is not a hit
    for _ in range(4):                                  # (whether or not the allocator hands the address out again)
        del vd
        vd = cases['plain'].to(dev)
        assert DenseMU.f16_stats(vd, W, H, be)[1] is False
        del vd
        vd = cases['exact'].to(dev)
        assert DenseMU.f16_stats(vd, W, H, be)[1] is True
    vd = cases['exact'].to(dev)
    vd[3, 3] += 2.0 ** -15
    assert DenseMU.f16_stats(vd, W, H, be)[1] is False


@pytest.mark.parametrize('prec,shape', [('f16', (300, 1000, 20)), ('f16r', (300, 1000, 20)), ('f16', (301, 999, 7)), ('f16', (1024, 2048, 64)),
                                        ('f16r', (2048, 4096, 128)), ('f16', (4096, 16384, 100))])
def test_riding_loss_equals_the_loss_pass(dev, monkeypatch, prec, shape):
    """Round 6: on unsharded beta == 1 fits served by the ping-pong kernel the loss of a checkpoint is not a
    pass over V any more -- the W half-step that FOLLOWS the checkpoint accumulates sum x log2(s + eps) (its reconstruction is
    the one nmf.py:400-401 evaluates), the other three sums of metrics.py:22 come from the target (once) and the factors'
    column sums.  Same value as the loss pass on the checkpoint's factors (<= 2e-6 relative: fp32 partial sums in another
    order), bit-identical factors from the carrying half-step, same stop iteration and factors from fit()."""
    from torchnmf_amd.engine import DenseMU
    from torchnmf_amd.nmf import NMF
    N, C, R = shape
    g = torch.Generator().manual_seed(N + R)
    V = torch.rand(N, C, generator=g) * torch.rand(N, 1, generator=g)
    V = (V.half().float() if prec == 'f16' else V).to(dev)
    W0, H0 = torch.rand(C, R, generator=g) + 0.1, torch.rand(N, R, generator=g) + 0.1
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('TORCHNMF_AMD_RIDING_LOSS', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = DenseMU(V, W, H, 1.0, precision=prec)
        assert (eng._riding is not None) == (mode == '1')
        for _ in range(3):
            eng.w_step(), eng.h_step()
        eng.checkpoint_begin()
        assert eng._riding_pending == (mode == '1')
        eng.w_step(), eng.h_step()
        W4, H4 = W.clone(), H.clone()
        div, left = eng.checkpoint_result()
        eng.rollback()
        ref = eng.divergence()                     # the loss pass on the checkpoint's (restored) factors
        assert abs(div - ref) <= 2e-6 * abs(ref) and not left, (div, ref)
        res[mode] = (div, W4.cpu(), H4.cpu(), W.cpu().clone(), H.cpu().clone())
    for i in (1, 2, 3, 4):
        assert torch.equal(res['1'][i], res['0'][i])
    record('riding_loss', prec=prec, shape=shape, riding=res['1'][0], loss_pass=res['0'][0])
    fits = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('TORCHNMF_AMD_RIDING_LOSS', mode)
        m = NMF(W=W0.clone(), H=H0.clone()).to(dev)
        fits[mode] = (m.fit(V, 1, 1e-3, 150, precision=prec), m.W.data.cpu().clone(), m.H.data.cpu().clone())
    assert fits['1'][0] == fits['0'][0] and 10 < fits['1'][0] < 150
    assert torch.equal(fits['1'][1], fits['0'][1]) and torch.equal(fits['1'][2], fits['0'][2])


@pytest.mark.parametrize('dtype', [torch.float64, torch.bfloat16])
@pytest.mark.parametrize('beta', [1, 0.5, 2])
def test_fit_module_cast_to_another_dtype(dev, dtype, beta):
    """VERDICT r5, missing 6: the reference fits in whatever dtype the module was cast to (`m.double()`, nmf.py:216-221); this
    engine's masters are fp32, so fit() runs on fp32 working copies and stores the result in the module's dtype.  float64:
    W, H and the iteration count against the oracle run in float64; bfloat16: the parameters keep their dtype and hold the
    fp32 result rounded once.  forward() answers in the module's dtype."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(41)
    V = torch.rand(90, 130, generator=g, dtype=torch.float64) + 0.01
    W0 = torch.rand(130, 7, generator=g, dtype=torch.float64) + 0.1
    H0 = torch.rand(90, 7, generator=g, dtype=torch.float64) + 0.1
    m = NMF(W=W0.float(), H=H0.float()).to(dev).to(dtype)
    assert m.W.dtype == dtype
    Vd = V.to(dev).to(dtype)
    n = m.fit(Vd, beta, 1e-4, 60, precision='bf16x3')
    assert m.W.dtype == dtype and m.H.dtype == dtype and m.W.requires_grad and m.H.requires_grad
    start = (W0, H0) if dtype == torch.float64 else (W0.float().to(dtype).double(), H0.float().to(dtype).double())
    Vr = V if dtype == torch.float64 else V.to(dtype).double()
    Wr, Hr, nr, _, _ = O.fit(Vr, start[0], start[1], beta, 1e-4, 60)
    assert n == nr
    tol = TOL if dtype == torch.float64 else 6e-3            # bf16 storage: 8 significant bits, rounded once at the end
    assert rel_err(m.W.data.double().cpu(), Wr) < tol and rel_err(m.H.data.double().cpu(), Hr) < tol
    out = m()
    assert out.dtype == dtype and out.shape == (90, 130)
    assert rel_err(out.double().cpu(), Hr @ Wr.t()) < tol


@pytest.mark.parametrize('beta', [1, 2])
@pytest.mark.parametrize('name,tW,tH', [('frozenW', False, True), ('frozenH', True, False)])
def test_fit_g4_frozen(dev, beta, name, tW, tH):
    from torchnmf_amd.nmf import NMF
    g = load_golden('g4_frozen')
    m = NMF(W=t(g['W0']), H=t(g['H0']), trainable_W=tW, trainable_H=tH).to(dev)
    m.fit(t(g['V']).to(dev), beta, NO_STOP, 20, precision='bf16x3')
    assert rel_err(m.W.data.cpu(), g[f'b{beta}_{name}_W']) < TOL
    assert rel_err(m.H.data.cpu(), g[f'b{beta}_{name}_H']) < TOL
    if not tW:
        assert torch.equal(m.W.data.cpu(), t(g['W0']))
    if not tH:
        assert torch.equal(m.H.data.cpu(), t(g['H0']))


@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
def test_metrics_beta_div_g6(dev, beta):
    from torchnmf_amd.metrics import beta_div
    g = load_golden('g6_beta_div')
    xs = {'rand': t(g['x_rand']), 'zero': torch.zeros(100)}
    ys = {'rand': t(g['y_rand']), 'zero': torch.zeros(100)}
    for xn, x in xs.items():
        for yn, y in ys.items():
            got = float(beta_div(x.to(dev), y.to(dev), beta))
            want = float(g[f'b{beta}_x{xn}_y{yn}'])
            assert got == pytest.approx(want, rel=1e-4, abs=1e-4), (beta, xn, yn)
            assert not np.isnan(got) and got >= -1e-6   # reference tests/test_metrics.py:6-14


# ----------------------------------------------------------------------------------------------------------
# reference-style behaviour tests (tests/test_nmf.py of the reference, on the device)
# ----------------------------------------------------------------------------------------------------------
def test_forward_shapes(dev):
    from torchnmf_amd.nmf import NMF
    m = NMF((100, 50)).to(dev)
    y = m()
    assert y.shape == (100, 50)
    assert rel_err(y.cpu(), m.H.data.cpu() @ m.W.data.cpu().t()) < 1e-6


@pytest.mark.parametrize('shape', [(1, 1, 1), (33, 130, 7), (128, 128, 32), (300, 257, 33), (513, 1000, 128), (200, 90, 256)])
def test_reconstruct_shapes(dev, shape):
    """NMF.reconstruct (nmf.py:691-693): ragged tiles, ranks that are not multiples of 4 / 32, explicit factors."""
    from torchnmf_amd.nmf import NMF
    N, C, R = shape
    g = torch.Generator().manual_seed(sum(shape))
    H, W = torch.rand(N, R, generator=g), torch.rand(C, R, generator=g)
    y = NMF.reconstruct(H.to(dev), W.to(dev))
    assert y.shape == (N, C) and y.is_contiguous()
    assert rel_err(y.cpu(), (H.double() @ W.double().t()).float()) < 1e-6
    m = NMF(W=W, H=H).to(dev)
    assert torch.equal(m(), y) and torch.equal(m(H=H.to(dev) * 2), NMF.reconstruct(H.to(dev) * 2, W.to(dev)))


@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
@pytest.mark.parametrize('tol', [0, 1e-4])
@pytest.mark.parametrize('alpha,l1_ratio', [(0, 0), (0.1, 0.5)])
def test_fit_smoke_like_reference(dev, beta, tol, alpha, l1_ratio):
    """tests/test_nmf.py:104-120: terminates within max_iter and produces no NaN."""
    from torchnmf_amd.nmf import NMF
    V = torch.rand(100, 50)
    m = NMF(V.shape, 8).to(dev)
    n = m.fit(V.to(dev), beta, tol, 100, False, alpha, l1_ratio)
    assert n <= 100
    assert not torch.any(torch.isnan(m.W)) and not torch.any(torch.isnan(m.H))


def test_fit_accepts_other_floating_dtypes_and_betamu_says_no(dev):
    """Round 6: a module cast with .double() fits (test_fit_module_cast_to_another_dtype has the parity side) -- an fp32 target
    is accepted beside float64 factors, as any target dtype is; BetaMu, which updates the parameters' own storage in place, says
    clearly that it needs float32."""
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    m = NMF((20, 30), 4).double().to(dev)
    n = m.fit(torch.rand(20, 30, device=dev), max_iter=20)
    assert 1 <= n <= 20 and m.W.dtype == torch.float64 and m.H.dtype == torch.float64
    assert bool(torch.isfinite(m.W).all()) and bool((m.W >= 0).all())
    V = torch.rand(20, 30, device=dev, dtype=torch.float64)
    with pytest.raises(NotImplementedError, match='float32'):
        BetaMu(m.parameters()).step(lambda: (V, m))


def test_fit_error_behaviour(dev):
    from torchnmf_amd.nmf import NMF
    m = NMF((20, 30), 4).to(dev)
    V = torch.rand(20, 30)
    V[0, 0] = -1
    with pytest.raises(AssertionError):
        m.fit(V.to(dev))
    V[0, 0] = 0
    with pytest.raises(ValueError):
        m.fit(V.to(dev), beta=0)
    with pytest.raises(ValueError):
        m.fit(V.to(dev), beta=-1)
    assert m.fit(V.to(dev), beta=1, max_iter=5) <= 5


# ----------------------------------------------------------------------------------------------------------
# size-independent properties at a size where the oracle is no longer cheap
# ----------------------------------------------------------------------------------------------------------
def test_large_slice_property_and_fixed_point(dev):
    """(a) W-update rows depend only on their own columns of V: compare a 256-column slice of a
    2048 x 16384 W half-step with the oracle on that slice.  (b) exact factorisation is a fixed point."""
    from oracle import mu_oracle as O
    from torchnmf_amd.engine import DenseMU
    g = torch.Generator().manual_seed(99)
    N, C, R = 2048, 16384, 128
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    V = torch.rand(N, C, generator=g)
    W = W0.to(dev).contiguous()
    H = H0.to(dev).contiguous()
    eng = DenseMU(V.to(dev), W, H, 1, precision='bf16x3')
    eng.w_step()
    torch.cuda.synchronize()
    sl = slice(5000, 5256)
    Wr = O.nmf_w_step(V[:, sl], W0[sl], H0, 1, 1.0)
    assert rel_err(W.cpu()[sl], Wr) < TOL
    # fixed point: V = H W^T exactly (fp32) => multiplicative factor == 1
    Vx = (H0.double() @ W0.double().t()).float()
    W = W0.to(dev).contiguous()
    H = H0.to(dev).contiguous()
    eng = DenseMU(Vx.to(dev), W, H, 1, precision='bf16x3')
    eng.w_step()
    eng.h_step()
    torch.cuda.synchronize()
    assert rel_err(W.cpu(), W0) < TOL and rel_err(H.cpu(), H0) < TOL


# ----------------------------------------------------------------------------------------------------------
# NMFD (1-D convolutive NMF): golden vectors from the reference + oracle
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['doc', 'mid', 'batch'])
@pytest.mark.parametrize('beta', [0.5, 1, 2])
def test_nmfd_fit_g5_golden(dev, name, beta):
    from torchnmf_amd.nmf import NMFD
    g = load_golden('g5_nmfd')
    V, W0, H0 = t(g[f'{name}_V']), t(g[f'{name}_W0']), t(g[f'{name}_H0'])
    m = NMFD(W=W0, H=H0).to(dev)
    n = m.fit(V.to(dev), beta, NO_STOP, 30, precision='bf16x3')
    assert n == 30
    ew, eh = rel_err(m.W.data.cpu(), g[f'{name}_b{beta}_W30']), rel_err(m.H.data.cpu(), g[f'{name}_b{beta}_H30'])
    assert ew < TOL and eh < TOL, (ew, eh)


@pytest.mark.parametrize('name', ['doc', 'mid', 'batch'])
def test_nmfd_regularised_and_reconstruct(dev, name):
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMFD
    g = load_golden('g5_nmfd')
    V, W0, H0 = t(g[f'{name}_V']), t(g[f'{name}_W0']), t(g[f'{name}_H0'])
    m = NMFD(W=W0, H=H0).to(dev)
    assert rel_err(m().cpu(), O.nmfd_reconstruct(H0, W0)) < 1e-5          # NMFD.reconstruct (nmf.py:776-779)
    m.fit(V.to(dev), 1, NO_STOP, 10, alpha=0.1, l1_ratio=0.5, precision='bf16x3')
    assert rel_err(m.W.data.cpu(), g[f'{name}_reg_W10']) < TOL and rel_err(m.H.data.cpu(), g[f'{name}_reg_H10']) < TOL


@pytest.mark.parametrize('prec,tol', [('bf16x3', TOL), ('bf16', 2e-2)])
def test_nmfd_medium_against_oracle(dev, prec, tol):
    """A spectrogram-like shape (C=257, L=1000, R=8, T=40): 3 iterations + early-stop bookkeeping vs the oracle."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMFD
    g = torch.Generator().manual_seed(21)
    B, Cc, L, R, T = 1, 257, 1000, 8, 40
    V = torch.rand(B, Cc, L, generator=g)
    if prec == 'bf16':
        V = V.bfloat16().float()
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    m = NMFD(W=W0, H=H0).to(dev)
    n = m.fit(V.to(dev), 1, NO_STOP, 3, precision=prec)
    Wr, Hr, nr, losses, _ = O.fit(V, W0, H0, 1, NO_STOP, 3, kind='nmfd')
    ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
    print(f'nmfd {prec}: relW={ew:.2e} relH={eh:.2e}')
    assert n == nr and ew < tol and eh < tol


@pytest.mark.parametrize('shape', [(1, 65, 304, 4, 8), (2, 40, 200, 3, 24), (1, 130, 1000, 5, 40), (3, 33, 96, 2, 16)])
@pytest.mark.parametrize('beta', [1, 2, 0.5])
def test_nmfd_implicit_toeplitz_operands(dev, shape, beta, monkeypatch):
    """Taps and frames that are multiples of 8 take the implicit path (GEMM operands fetched from the 8x window
    tables of H, nmfmu_conv_tables); it must agree with the explicit-unfold path operand for operand (same bf16
    products, same k order) and with the oracle.  Shapes cover k tiles that straddle rank boundaries (R*T = 72),
    batches, the minimum tap count and padding rows / columns on every side."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMFD
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_EXPLICIT', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, beta, precision='bf16x3')
        assert eng.implicit == (mode == '0')
        l0 = eng.divergence()
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        res[mode] = (W.cpu(), H.cpu(), l0, eng.divergence())
    assert rel_err(res['0'][0], res['1'][0]) < 1e-6 and rel_err(res['0'][1], res['1'][1]) < 1e-6
    assert res['0'][2] == pytest.approx(res['1'][2], rel=1e-6) and res['0'][3] == pytest.approx(res['1'][3], rel=1e-6)
    Wr, Hr, _, _, _ = O.fit(V, W0, H0, beta, NO_STOP, 2, kind='nmfd')
    assert rel_err(res['0'][0], Wr) < TOL and rel_err(res['0'][1], Hr) < TOL


@pytest.mark.parametrize('shape', [(1, 40, 520, 3, 136), (2, 33, 335, 2, 130), (1, 70, 600, 2, 400), (3, 20, 300, 1, 128)])
@pytest.mark.parametrize('beta,prec', [(1, 'bf16x3'), (2, 'bf16x3'), (0.5, 'bf16x3'), (1, 'bf16')])
def test_nmfd_fold_from_tile_diagonal_sums(dev, shape, beta, prec, monkeypatch):
    """With >= 128 taps the H numerator GEMM does not store Y[(r,t)][(b,l)]: its epilogue emits the diagonal sums of
    every 128 x 128 tile (NMFMU_EPI_FOLD) and nmfmu_conv_fold_parts_apply_h gathers them.  Same products, another
    summation order: must agree with the store-then-fold path to fp32 rounding, and with the oracle.  Shapes put rank
    boundaries (T = 130, 136, 400) and batch boundaries (L = 335, 300) inside tiles, B*L and R*T off the tile grid."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    res = {}
    monkeypatch.setenv('TORCHNMF_AMD_NMFD_H_ROWS', '0')      # the store-then-fold path is the comparison here
    for mode in ('0', '1'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_FOLD_PARTS', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, beta, 0.01, 0.02, precision=prec)
        assert eng.fold_parts == (mode == '1')
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        res[mode] = (W.cpu(), H.cpu(), eng.divergence())
    assert rel_err(res['0'][0], res['1'][0]) < 2e-6 and rel_err(res['0'][1], res['1'][1]) < 2e-6
    assert (res['0'][1] - res['1'][1]).abs().max() <= 1e-5 * res['0'][1].abs().max()
    assert res['0'][2] == pytest.approx(res['1'][2], rel=1e-5)
    if prec == 'bf16x3':
        Wr, Hr, _, _, _ = O.fit(V, W0, H0, beta, NO_STOP, 2, alpha=0.03, l1_ratio=1.0 / 3.0, kind='nmfd')
        assert rel_err(res['1'][0], Wr) < TOL and rel_err(res['1'][1], Hr) < TOL


@pytest.mark.parametrize('shape,tail', [((1, 520, 600, 2, 136), '1,3'), ((2, 600, 335, 3, 130), '2,2'), ((1, 1025, 776, 1, 400), '4,6')])
@pytest.mark.parametrize('beta,prec', [(1, 'bf16x3'), (2, 'bf16x3'), (1, 'f16')])
def test_nmfd_h_numerator_tail_round_split(dev, shape, tail, beta, prec, monkeypatch):
    """Tail-round split of the H-numerator GEMM (round 3): its last tile rows run contraction-split, their per-tile
    diagonal sums arrive as partial slabs and the gather adds them in a fixed order.  Forced here on small shapes
    (configs[3] selects it by itself: 1600 tiles on 512 slots): must agree with the unsplit launch to fp32 rounding and
    with the oracle."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    if prec == 'f16' and (T % 8 or L % 8):
        pytest.skip('fp16 operands need implicit Toeplitz operands (taps and frames multiples of 8)')
    res = {}
    for mode in ('0', tail):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_TAIL_SPLIT', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, beta, 0.01, 0.02, precision=prec)
        assert eng.fold_parts and (eng.h_tail_rows, eng.h_tail_split) == ((0, 1) if mode == '0' else tuple(int(v) for v in tail.split(',')))
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        res[mode] = (W.cpu(), H.cpu())
    assert rel_err(res[tail][0], res['0'][0]) < 2e-6 and rel_err(res[tail][1], res['0'][1]) < 2e-6
    if prec == 'bf16x3':
        Wr, Hr, _, _, _ = O.fit(V, W0, H0, beta, NO_STOP, 2, alpha=0.03, l1_ratio=1.0 / 3.0, kind='nmfd')
        assert rel_err(res[tail][0], Wr) < TOL and rel_err(res[tail][1], Hr) < TOL


@pytest.mark.parametrize('shape', [(1, 129, 304, 4, 8), (2, 257, 200, 3, 24), (1, 136, 600, 2, 400), (1, 1025, 520, 3, 136),
                                   (3, 130, 96, 9, 5)])
@pytest.mark.parametrize('beta,prec', [(1, 'bf16x3'), (2, 'bf16x3'), (0.5, 'bf16x3'), (1, 'bf16')])
def test_nmfd_ragged_channels(dev, shape, beta, prec, monkeypatch):
    """C = 128 k + (1..8) channels (the 1025 bins of configs[3]): the reconstruction GEMMs cover the first 128 k channels,
    nmfmu_conv_ragged_rows the rest by direct summation, for both half-steps and the loss.  Must agree with the all-GEMM
    path (same rounded operands, another summation order for the ragged rows) and with the oracle; ragged counts 1, 2
    and 8, batches, implicit and explicit Toeplitz operands, more ranks than rank groups (R = 9)."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_RAGGED', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, beta, 0.01, 0.02, precision=prec)
        assert eng.ragged == (mode == '1')
        l0 = eng.divergence()
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        res[mode] = (W.cpu(), H.cpu(), l0, eng.divergence())
    tol = 2e-6 if prec == 'bf16x3' else 2e-5     # bf16: a ratio that rounds the other way moves an element by 2^-8
    assert rel_err(res['0'][0], res['1'][0]) < tol and rel_err(res['0'][1], res['1'][1]) < tol
    # the ragged rows themselves (not drowned in the norm of the other 128 k)
    cm = (Cc // 128) * 128
    assert rel_err(res['0'][0][cm:], res['1'][0][cm:]) < (1e-5 if prec == 'bf16x3' else 5e-3)
    assert res['0'][2] == pytest.approx(res['1'][2], rel=1e-5) and res['0'][3] == pytest.approx(res['1'][3], rel=1e-5)
    if prec == 'bf16x3':
        Wr, Hr, _, _, _ = O.fit(V, W0, H0, beta, NO_STOP, 2, alpha=0.03, l1_ratio=1.0 / 3.0, kind='nmfd')
        assert rel_err(res['1'][0], Wr) < TOL and rel_err(res['1'][1], Hr) < TOL
        assert rel_err(res['1'][0][cm:], Wr[cm:]) < TOL


@pytest.mark.parametrize('shape', [(1, 1025, 304, 4, 8), (2, 1026, 200, 3, 24), (1, 1025, 520, 3, 136), (1, 1032, 328, 2, 40),
                                   (1, 1153, 328, 2, 136)])
@pytest.mark.parametrize('beta,prec', [(1, 'f16'), (1, 'bf16'), (1, 'bf16x3'), (2, 'bf16x3'), (0.5, 'bf16x3'), (0, 'bf16')])
def test_nmfd_ragged_channels_inside_the_gemm_grid(dev, shape, beta, prec, monkeypatch):
    """The 1 .. 8 channels beyond the last whole 128-channel tile ride inside the reconstruction GEMMs' grids
    (nmfmu_gemm_desc.rag_c0 / rag_channels: one extra 16 x 16 x 32 MFMA block per workgroup, eight workgroups sharing out
    the 128 frames of an implicit-operand tile) instead of in nmfmu_conv_ragged_rows launches.  Same rounded operands,
    MFMA instead of fp32 FMA summation: must agree with the separate-launch path and with the oracle -- one, two and
    eight ragged channels, eight and nine channel tiles (the ninth takes no part), batches, frame counts that leave
    padding columns, every beta branch and operand mode, both half-steps."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    if prec == 'f16' and T < 128:
        pytest.skip("NMFD 'f16' needs >= 128 taps")
    g = torch.Generator().manual_seed(sum(shape) + 1)
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_RAGGED_IN_GRID', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, beta, 0.01, 0.02, precision=prec)
        assert eng.ragged and eng.ragged_in_grid == (mode == '1')
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        res[mode] = (W.cpu(), H.cpu(), eng.divergence())
    single = prec != 'bf16x3'
    tol = 2e-5 if single else 2e-6      # single plane: a ratio that rounds the other way moves an element by 2^-8 / 2^-11
    assert rel_err(res['0'][0], res['1'][0]) < tol and rel_err(res['0'][1], res['1'][1]) < tol
    cm = (Cc // 128) * 128
    assert rel_err(res['0'][0][cm:], res['1'][0][cm:]) < (5e-3 if single else 1e-5)
    assert res['0'][2] == pytest.approx(res['1'][2], rel=1e-5)
    key = ('nmfd_ragged_in_grid', shape, beta)          # the same reference serves every operand mode of a (shape, beta)
    if key not in _ORACLE_CACHE:                         # (the CPU oracle is most of this suite's wall clock on the GPU box)
        _ORACLE_CACHE[key] = O.fit(V, W0, H0, beta, NO_STOP, 2, alpha=0.03, l1_ratio=1.0 / 3.0, kind='nmfd')[:2]
    Wr, Hr = _ORACLE_CACHE[key]
    bar = TOL if prec == 'bf16x3' else 1e-4 if prec == 'f16' else 3e-3
    assert rel_err(res['1'][0], Wr) < bar and rel_err(res['1'][1], Hr) < bar
    # (the ragged rows alone: a few hundred values, no averaging over the factor -- fp16's bar is the whole-factor one)
    assert rel_err(res['1'][0][cm:], Wr[cm:]) < {'bf16x3': TOL, 'f16': 3e-4, 'bf16': 1e-2}[prec]
    record('nmfd_ragged_in_grid', shape=list(shape), beta=beta, precision=prec, rel_W=rel_err(res['1'][0], Wr),
           rel_H=rel_err(res['1'][1], Hr), rel_W_ragged_rows=rel_err(res['1'][0][cm:], Wr[cm:]))


@pytest.mark.parametrize('shape', [(2, 40, 520, 3, 136), (1, 129, 1208, 2, 400), (1, 33, 392, 2, 128), (3, 16, 1096, 1, 136)])
@pytest.mark.parametrize('prec', ['bf16x3', 'bf16', 'f16'])
def test_nmfd_h_update_rewrites_the_window_tables(dev, shape, prec, monkeypatch):
    """beta == 1 on implicit operands: the H update's launch also rewrites the window tables of the new H
    (nmfmu_conv_fold_parts_apply_h_tables) instead of a nmfmu_conv_tables launch per iteration.  Blocks own 242 positions
    and recompute a 7-wide halo from a read-only shadow of the old H (two shadows alternate).  After every iteration the
    tables must equal, byte for byte, what nmfmu_conv_tables builds from the new H; the factors must agree with the
    separate-launch path (only the partition of the rank sums differs).  One to five blocks per
    (batch, rank) row, batches, an H changed from outside between iterations (refresh_images)."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    g = torch.Generator().manual_seed(sum(shape) + 2)
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_FUSED_TABLES', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, 1, 0.01, 0.02, precision=prec)
        assert eng.fused_sums and eng.fused_tables == (mode == '1')
        for it in range(3):
            eng.w_step()
            eng.h_step()
            if mode == '1':
                got = [eng.hu.hi.clone(), eng.hut.hi.clone()] + ([eng.hu.lo.clone(), eng.hut.lo.clone()] if prec == 'bf16x3' else [])
                eng._pack_h(sums=False)          # the standalone kernel on the same H
                want = [eng.hu.hi, eng.hut.hi] + ([eng.hu.lo, eng.hut.lo] if prec == 'bf16x3' else [])
                for a_, b_ in zip(got, want):
                    assert torch.equal(a_, b_), f'table bytes differ after iteration {it}'
            if it == 1:                          # H changed from outside: the images and the shadow must follow
                H.mul_(1.25)
                eng.refresh_images()
        res[mode] = (W.cpu(), H.cpu(), eng.divergence())
    tol = 3e-6 if prec == 'bf16x3' else 3e-5
    assert rel_err(res['0'][0], res['1'][0]) < tol and rel_err(res['0'][1], res['1'][1]) < tol
    assert res['0'][2] == pytest.approx(res['1'][2], rel=1e-5)


@pytest.mark.parametrize('shape', [(1, 40, 520, 3, 136), (2, 33, 335, 2, 130), (1, 129, 600, 2, 400), (3, 70, 300, 5, 128)])
@pytest.mark.parametrize('prec', ['bf16x3', 'bf16'])
def test_nmfd_rank_sums_ride_in_their_producers(dev, shape, prec, monkeypatch):
    """beta == 1 with >= 128 taps: the denominators sum_{c,t} W and sum_{b,j} H (nmf.py:122-131) are not separate
    reduction launches -- conv_apply_pack_w leaves per-channel-tile column sums of W that the H update finishes, the H
    update leaves per-block sums that the next W update finishes.  Same values, another summation order: must agree with
    the nmfmu_rank_sums path to fp32 rounding over three iterations (the hand-over crosses iteration boundaries)."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_FUSED_SUMS', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, 1, 0.01, 0.02, precision=prec)
        assert eng.fused_sums == (mode == '1')
        for _ in range(3):
            eng.w_step()
            eng.h_step()
        res[mode] = (W.cpu(), H.cpu(), eng.divergence())
    tol = 3e-6 if prec == 'bf16x3' else 3e-5
    assert rel_err(res['0'][0], res['1'][0]) < tol and rel_err(res['0'][1], res['1'][1]) < tol
    assert res['0'][2] == pytest.approx(res['1'][2], rel=1e-5)
    if prec == 'bf16x3':
        Wr, Hr, _, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 3, alpha=0.03, l1_ratio=1.0 / 3.0, kind='nmfd')
        assert rel_err(res['1'][0], Wr) < TOL and rel_err(res['1'][1], Hr) < TOL


@pytest.mark.parametrize('shape', [(1, 130, 2100, 2, 136), (2, 40, 1100, 3, 128)])
@pytest.mark.parametrize('prec', ['bf16x3', 'bf16'])
def test_nmfd_w_numerator_split_k(dev, shape, prec, monkeypatch):
    """The W numerator GEMM (few tiles, contraction over all frames) runs as two half-contractions whose partials the
    apply kernel adds (nmfmu_gemm_desc.k_split, implicit HuT operand starting mid-way): same products, one more rounding
    per element."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_KSPLIT', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, 1, precision=prec)
        assert eng.w_ksplit == (2 if mode == '1' else 1)
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        res[mode] = (W.cpu(), H.cpu(), eng.divergence())
    tol = 2e-6 if prec == 'bf16x3' else 2e-5
    assert rel_err(res['0'][0], res['1'][0]) < tol and rel_err(res['0'][1], res['1'][1]) < tol
    assert res['0'][2] == pytest.approx(res['1'][2], rel=1e-5)
    if prec == 'bf16x3':
        Wr, Hr, _, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 2, kind='nmfd')
        assert rel_err(res['1'][0], Wr) < TOL and rel_err(res['1'][1], Hr) < TOL


@pytest.mark.parametrize('shape', [(1, 257, 1096, 4, 136), (2, 130, 640, 3, 128), (1, 64, 2200, 2, 400)])
def test_nmfd_f16_mode(dev, shape):
    """precision='f16' of the NMFD engine (fp16 operand planes, window tables and ratio planes; beta == 1, >= 128 taps,
    implicit operands): three iterations against the fp32 oracle -- one rounding to 11 significant bits per operand, so
    an order of magnitude closer than the bf16 single-plane mode -- through the ragged-channel, fold-parts, fused-sums and
    split-K paths; exact zeros stay exact; unsupported configurations refuse loudly; 'auto' only picks it when the
    problem is large."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, L, R, T = shape
    g = torch.Generator().manual_seed(sum(shape))
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    W0[3] = 0.0                                   # a silent channel template stays exactly zero
    Wr, Hr, _, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 3, kind='nmfd')
    err = {}
    for prec in ('f16', 'bf16'):
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, 1, precision=prec)
        assert eng.precision_name == prec and eng.fused_sums
        l0 = eng.divergence()
        for _ in range(3):
            eng.w_step()
            eng.h_step()
        err[prec] = (rel_err(W.cpu(), Wr), rel_err(H.cpu(), Hr))
        assert bool(torch.isfinite(W).all()) and bool(torch.isfinite(H).all())
        assert float(W[3].abs().max()) == 0.0
        assert l0 == pytest.approx(float(O.beta_div(O.nmfd_reconstruct(H0, W0), V, 1)), rel=2e-3 if prec == 'bf16' else 3e-4)
    print(f'nmfd f16 {shape}: f16 relW={err["f16"][0]:.2e} relH={err["f16"][1]:.2e}; bf16 relW={err["bf16"][0]:.2e}')
    assert max(err['f16']) < 6e-4 and max(err['f16']) < 0.4 * max(err['bf16'])
    # fewer than 128 taps: the window-operand path (round 4) has fp16 planes too; beta != 1 / unaligned frames: not built
    short = ConvMU(V[:, :, :256].to(dev).contiguous(), W0[:, :, :16].clone().to(dev).contiguous(),
                   torch.rand(B, R, 241).to(dev), 1, precision='f16')
    assert short.h_rows and not short.fold_parts
    with pytest.raises(ValueError):
        ConvMU(V.to(dev), W0.clone().to(dev), H0.clone().to(dev), 2, precision='f16')
    with pytest.raises(ValueError):
        ConvMU(V[:, :, :251].to(dev).contiguous(), W0[:, :, :16].clone().to(dev).contiguous(),
               torch.rand(B, R, 236).to(dev), 1, precision='f16')
    small = ConvMU(V.to(dev), W0.clone().to(dev), H0.clone().to(dev), 1, precision='auto')
    assert small.precision_name == ('f16' if min(Cc, B * L, R * T) >= 1024 else 'bf16x3')


@pytest.mark.parametrize('shape', [
    (1, 64, (64, 128), 8, (8, 16)),          # B, C, ls, R, ks: NMF2D, every contraction >= 1024 -> 'auto' takes fp16
    (2, 96, (2048,), 16, (64,)),             # NMFD below 128 taps
    (1, 40, (12, 24, 32), 4, (2, 4, 8)),     # NMF3D (contractions too short for 'auto')
])
def test_fp16_operands_on_the_window_operand_path(dev, shape):
    """precision 'f16' beyond the fold-parts path (round 4): fp16 window tables (1 - 3 shift axes), fp16 ratio planes,
    fp16 W planes for the H numerator GEMM; beta == 1.  Same bar as the 1-D fold path: closer to the oracle than bf16 by a
    wide margin, and 'auto' takes it exactly when every contraction has >= 1024 terms."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, ls, R, ks = shape
    g = torch.Generator().manual_seed(sum(ls) + R)
    V = torch.rand(B, Cc, *ls, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, *ks, generator=g).abs()
    H0 = torch.randn(B, R, *[l - k + 1 for l, k in zip(ls, ks)], generator=g).abs()
    kind = 'nmfd' if len(ls) == 1 else 'convnd'
    Wr, Hr, _, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 3, kind=kind)
    recon0 = O.nmfd_reconstruct(H0, W0) if len(ls) == 1 else O.convnd_reconstruct(H0, W0)
    err = {}
    for prec in ('f16', 'bf16'):
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, 1, precision=prec)
        assert eng.precision_name == prec and eng.h_rows and eng.implicit
        l0 = eng.divergence()
        for _ in range(3):
            eng.w_step()
            eng.h_step()
        err[prec] = (rel_err(W.cpu(), Wr), rel_err(H.cpu(), Hr))
        assert bool(torch.isfinite(W).all()) and bool(torch.isfinite(H).all())
        assert l0 == pytest.approx(float(O.beta_div(recon0, V, 1)), rel=2e-3 if prec == 'bf16' else 3e-4)
    print(f'f16 window path {shape}: f16 relW={err["f16"][0]:.2e} relH={err["f16"][1]:.2e}; bf16 relW={err["bf16"][0]:.2e} relH={err["bf16"][1]:.2e}')
    assert max(err['f16']) < 6e-4 and max(err['f16']) < 0.4 * max(err['bf16'])
    T, L = int(np.prod(ks)), int(np.prod(ls))
    auto = ConvMU(V.to(dev), W0.clone().to(dev), H0.clone().to(dev), 1, precision='auto')
    assert auto.precision_name == ('f16' if min(Cc * T, B * L, R * T) >= 1024 else 'bf16x3')
    if auto.precision_name == 'f16':
        assert max(err['f16']) < 2e-4


def test_nmf2d_default_fit_takes_fp16_operands(dev):
    """NMF2D.fit with its default precision on a problem whose contractions all have >= 1024 terms: 'auto' -> fp16
    operands (1x the MFMA work), 20 iterations through the asynchronous checkpoint loop, within 1e-4 of the oracle's
    20 iterations (nmf.py:297-409 on the conv2d model)."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF2D
    g = torch.Generator().manual_seed(77)
    V = torch.rand(1, 64, 64, 128, generator=g) + 1e-3
    W0 = torch.randn(64, 8, 8, 16, generator=g).abs()
    H0 = torch.randn(1, 8, 57, 113, generator=g).abs()
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 20, kind='convnd')
    m = NMF2D(W=W0, H=H0).to(dev)
    n = m.fit(V.to(dev), beta=1, tol=NO_STOP, max_iter=20)
    assert n == nr == 20 and m.last_precision == 'f16'
    ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
    print(f'NMF2D default fit, 20 iterations: rel W={ew:.2e} H={eh:.2e}')
    assert ew < TOL and eh < TOL


@pytest.mark.parametrize('cls,ls', [('NMF2D', (12, 10)), ('NMF3D', (5, 6, 7))])
def test_convnd_default_kernel_size_one(dev, cls, ls):
    """The reference's default kernel_size is 1 (nmf.py:838, 920; its own construct tests use it): the conv model is then
    plain NMF over (batch x positions) -- one tap, no window-operand path, explicit operands."""
    from oracle import mu_oracle as O
    from torchnmf_amd import nmf as anmf
    g = torch.Generator().manual_seed(len(ls))
    V = torch.rand(2, 5, *ls, generator=g) + 1e-3
    m = getattr(anmf, cls)(V.shape, rank=4)
    assert tuple(m.W.shape) == (5, 4) + (1,) * len(ls) and tuple(m.H.shape) == (2, 4) + ls
    W0, H0 = m.W.data.clone(), m.H.data.clone()
    m = m.to(dev)
    n = m.fit(V.to(dev), beta=1, tol=NO_STOP, max_iter=10)
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 10, kind='convnd')
    assert n == nr and rel_err(m.W.data.cpu(), Wr) < TOL and rel_err(m.H.data.cpu(), Hr) < TOL
    assert tuple(m().shape) == tuple(V.shape)


def test_nmfd_auto_warns_when_alignment_costs_the_fp16_mode(dev):
    """a spectrogram whose frame count is not a multiple of 8 cannot take the implicit operands, so
    'auto' runs split bf16 at 3x the matrix work -- the user is told."""
    from torchnmf_amd.nmfd_engine import ConvMU
    g = torch.Generator().manual_seed(3)
    V = (torch.rand(1, 1024, 1027, generator=g) + 1e-3).to(dev)
    W, H = torch.rand(1024, 8, 128, generator=g).to(dev), torch.rand(1, 8, 900, generator=g).to(dev)
    with pytest.warns(UserWarning, match='multiples of 8'):
        eng = ConvMU(V, W, H, 1, precision='auto')
    assert eng.precision_name == 'bf16x3' and not eng.implicit
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')               # aligned: fp16 mode, no warning
        eng = ConvMU(V[:, :, :1024].contiguous(), W, H[:, :, :897].contiguous(), 1, precision='auto')
    assert eng.precision_name == 'f16'


@pytest.mark.parametrize('shape', [
    # (B, C, ls, R, ks)
    (2, 70, (24, 40), 5, (3, 8)),        # two batches, 70 channels = two 64-channel k-tiles per tap, rank -> 32-wide tile
    (1, 64, (40, 33), 40, (4, 5)),       # rank 40 -> 64-wide tile, odd extents (no alignment rule on this path)
    (1, 130, (10, 12, 14), 100, (2, 3, 2)),   # three shift axes, rank 100 -> 128-wide tile, three k-tiles per tap
    (3, 20, (150,), 7, (30,)),           # one axis (NMFD below the fold-parts path's 128 taps)
])
@pytest.mark.parametrize('beta', [1, 0.5])
def test_h_numerator_from_shifted_ratio_rows(dev, shape, beta, monkeypatch):
    """H half-step with the numerator as the window-operand GEMM (NMFMU_OPS_A_WIN: no Y matrix, no fold) against the
    oracle's conv backward pass (nmf.py:380-391 on the convolutive models) and against the engine's own Y + fold path."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, ls, R, ks = shape
    g = torch.Generator().manual_seed(sum(ls) + R)
    V = torch.rand(B, Cc, *ls, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, *ks, generator=g).abs()
    H0 = torch.randn(B, R, *[l - k + 1 for l, k in zip(ls, ks)], generator=g).abs()
    kind = {1: 'nmfd', 2: 'convnd', 3: 'convnd'}[len(ls)]
    Wr, Hr, _, _, _ = O.fit(V, W0, H0, beta, NO_STOP, 2, kind=kind)
    res = {}
    for rows in ('1', '0'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_H_ROWS', rows)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, beta, precision='bf16x3')
        assert eng.h_rows == (rows == '1') and (eng.y is None) == (rows == '1')
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        res[rows] = (W.cpu(), H.cpu())
        assert rel_err(res[rows][0], Wr) < TOL and rel_err(res[rows][1], Hr) < TOL
    assert rel_err(res['1'][1], res['0'][1]) < 2e-5
    monkeypatch.setenv('TORCHNMF_AMD_NMFD_H_ROWS', '1')
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = ConvMU(V.to(dev), W, H, beta, precision='bf16')
    for _ in range(2):
        eng.w_step()
        eng.h_step()
    assert rel_err(H.cpu(), Hr) < 2e-2 and bool(torch.isfinite(H).all())


@pytest.mark.parametrize('shape', [
    (2, 70, (24, 40), 5, (3, 8)),            # B, C, ls, R, ks: last axis 40 frames / 8 taps
    (1, 130, (9, 32), 3, (4, 16)),           # R * T = 192: k padding inside the last k-tile pair
    (2, 9, (6, 7, 16), 4, (2, 3, 8)),        # three shift axes
])
@pytest.mark.parametrize('beta', [1, 0.5, 2, 0])
def test_implicit_operands_with_several_shift_axes(dev, shape, beta, monkeypatch):
    """NMF2D / NMF3D with taps and frames of the last axis multiples of 8: the GEMMs fetch Hu / HuT from the window
    tables of nmfmu_convnd_tables (no unfold kernel, no T-times-H planes).  Same products as the explicit operands:
    must agree with them to fp32 rounding, and with the oracle (nmf.py:857-865, 937-942)."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    B, Cc, ls, R, ks = shape
    g = torch.Generator().manual_seed(sum(ls) + R)
    V = torch.rand(B, Cc, *ls, generator=g) + 1e-3
    W0 = torch.randn(Cc, R, *ks, generator=g).abs()
    H0 = torch.randn(B, R, *[l - k + 1 for l, k in zip(ls, ks)], generator=g).abs()
    Wr, Hr, _, _, _ = O.fit(V, W0, H0, beta, NO_STOP, 2, kind='convnd')
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_EXPLICIT', mode)
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = ConvMU(V.to(dev), W, H, beta, precision='bf16x3')
        assert eng.implicit == (mode == '0')
        l0 = eng.divergence()
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        res[mode] = (W.cpu(), H.cpu(), l0, eng.divergence())
    assert rel_err(res['0'][0], res['1'][0]) < 2e-6 and rel_err(res['0'][1], res['1'][1]) < 2e-6
    assert res['0'][2] == pytest.approx(res['1'][2], rel=1e-6) and res['0'][3] == pytest.approx(res['1'][3], rel=1e-6)
    assert rel_err(res['0'][0], Wr) < TOL and rel_err(res['0'][1], Hr) < TOL
    assert res['0'][2] == pytest.approx(float(O.beta_div(O.convnd_reconstruct(H0, W0), V, beta)), rel=1e-4)


@pytest.mark.parametrize('name,cls', [('2d_a', 'NMF2D'), ('2d_b', 'NMF2D'), ('3d_a', 'NMF3D')])
@pytest.mark.parametrize('beta', [0.5, 1, 2])
def test_nmf2d_nmf3d_fit_g8_golden(dev, name, cls, beta):
    """NMF2D / NMF3D (nmf.py:782-942) against the reference's own outputs (g8_convnd)."""
    from torchnmf_amd import nmf as anmf
    g = load_golden('g8_convnd')
    V, W0, H0 = t(g[f'{name}_V']), t(g[f'{name}_W0']), t(g[f'{name}_H0'])
    m = getattr(anmf, cls)(W=W0, H=H0).to(dev)
    assert rel_err(m().cpu(), g[f'{name}_recon']) < 1e-5 and tuple(m().shape) == tuple(V.shape)
    n = m.fit(V.to(dev), beta, NO_STOP, 20, precision='bf16x3')
    assert n == 20
    assert rel_err(m.W.data.cpu(), g[f'{name}_b{beta}_W20']) < TOL and rel_err(m.H.data.cpu(), g[f'{name}_b{beta}_H20']) < TOL
    if beta == 1:
        m = getattr(anmf, cls)(W=W0, H=H0).to(dev)
        m.fit(V.to(dev), 1, NO_STOP, 10, alpha=0.1, l1_ratio=0.5, precision='bf16x3')
        assert rel_err(m.W.data.cpu(), g[f'{name}_reg_W10']) < TOL and rel_err(m.H.data.cpu(), g[f'{name}_reg_H10']) < TOL


def test_nmf2d_nmf3d_constructors_and_docstring_shapes(dev):
    """The reference docstrings' examples (nmf.py:826-836, 909-918) and its constructor rules."""
    from torchnmf_amd.nmf import NMF2D, NMF3D
    V = torch.rand(1, 1, 33, 50)
    m = NMF2D(V.shape, 16, 3)
    assert tuple(m.W.shape) == (1, 16, 3, 3) and tuple(m.H.shape) == (1, 16, 31, 48) and m.kernel_size == (3, 3)
    m = m.to(dev)
    assert tuple(m().shape) == (1, 1, 33, 50)
    assert m.fit(V.to(dev), max_iter=15) <= 15 and bool(torch.all(m.W >= 0)) and bool(torch.all(m.H >= 0))
    V3 = torch.rand(1, 3, 16, 16, 20)
    m3 = NMF3D(V3.shape, 8, (5, 5, 6))
    assert tuple(m3.W.shape) == (3, 8, 5, 5, 6) and tuple(m3.H.shape) == (1, 8, 12, 12, 15) and m3.rank == 8
    m3 = m3.to(dev)
    assert tuple(m3().shape) == tuple(V3.shape)
    assert m3.fit(V3.to(dev), beta=2, max_iter=10) <= 10
    assert NMF2D((1, 2, 7, 9), kernel_size=(2, 3)).rank == 7 and NMF3D((1, 2, 5, 7, 9), kernel_size=2).rank == 7


# ----------------------------------------------------------------------------------------------------------
# column-sharded path on the real backend (RCCL, world_size 1: the same kernels and collectives as N > 1)
# ----------------------------------------------------------------------------------------------------------
def test_sharded_path_world1_rccl(dev):
    import os
    import socket
    import torch.distributed as dist
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        g = load_golden('g1_nmf_small')
        V, W0, H0 = t(g['V']), t(g['W0']), t(g['H0'])
        for beta, alpha in ((1, 0.0), (2, 0.1), (0.5, 0.0)):
            m = NMF(W=W0, H=H0).to(dev)
            n = m.fit(V.to(dev), beta, 1e-4, 60, alpha=alpha, l1_ratio=0.5, precision='bf16x3',
                      process_group=dist.group.WORLD)
            Wr, Hr, nr, _, _ = O.fit(V, W0, H0, beta, 1e-4, 60, alpha, 0.5)
            assert n == nr
            assert rel_err(m.W.data.cpu(), Wr) < TOL and rel_err(m.H.data.cpu(), Hr) < TOL
        # the throughput kernels (bf16, rank pad 128: eight-wave pipelined H half-step) on the sharded path:
        # slab_reduce -> packed buffer -> all_reduce -> apply must reproduce the single-device path
        from torchnmf_amd.engine import DenseMU
        gg = torch.Generator().manual_seed(9)
        Vb = torch.rand(700, 2100, generator=gg).bfloat16().float().to(dev)
        Wb, Hb = torch.randn(2100, 100, generator=gg).abs(), torch.randn(700, 100, generator=gg).abs()
        res = []
        for grp, overlap in ((None, '1'), (dist.group.WORLD, '1'), (dist.group.WORLD, '0'), (dist.group.WORLD, 'direct')):
            os.environ['TORCHNMF_AMD_AR_OVERLAP'] = '0' if overlap == 'direct' else overlap
            W, H = Wb.clone().to(dev), Hb.clone().to(dev)
            # 'direct' (round 4): the whole sharded H half-step as one C call with the library's own RCCL communicator
            eng = DenseMU(Vb, W, H, 1.0, precision='bf16', group=grp, ar_direct=(overlap == 'direct') if grp is not None else None)
            if grp is not None:     # 700 rows pad to 768: the overlapped form runs rows [0, 256) and [256, 768) separately
                assert (eng._comm is not None) == (overlap == 'direct')
                assert (eng._h_rows is not None) == (overlap == '1')
                if overlap == '1':
                    assert [(v.r0, v.owner.rows, v.owner.rows_pad) for v in eng._h_rows] == [(0, 256, 256), (256, 444, 512)]
            for _ in range(3):
                eng.w_step()
                eng.h_step()
            res.append((W.cpu(), H.cpu(), eng.divergence()))
        # (the paths sum the column sums in different orders; a 1e-7 difference flips a few bf16 roundings)
        for r in res[1:]:
            assert rel_err(r[0], res[0][0]) < 1e-4 and rel_err(r[1], res[0][1]) < 1e-4
            assert r[2] == pytest.approx(res[0][2], rel=1e-5)
        # (round 6) the 3-byte target on the sharded path: the same partial-sum kernels (ping-pong instance with six X pieces per
        # tile), row halves, packed all-reduce and apply -- against the single-device engine on the same plain-float target
        Vr = torch.rand(700, 2100, generator=gg).to(dev)
        res = []
        for grp, overlap in ((None, '1'), (dist.group.WORLD, '1'), (dist.group.WORLD, '0'), (dist.group.WORLD, 'direct')):
            os.environ['TORCHNMF_AMD_AR_OVERLAP'] = '0' if overlap == 'direct' else overlap
            W, H = Wb.clone().to(dev), Hb.clone().to(dev)
            eng = DenseMU(Vr, W, H, 1.0, precision='f16r', group=grp, ar_direct=(overlap == 'direct') if grp is not None else None)
            assert eng.precision_name == 'f16r' and eng.step_h.block_rows == 256
            for _ in range(3):
                eng.w_step()
                eng.h_step()
            res.append((W.cpu(), H.cpu(), eng.divergence()))
        for r in res[1:]:
            assert rel_err(r[0], res[0][0]) < 1e-4 and rel_err(r[1], res[0][1]) < 1e-4
            assert r[2] == pytest.approx(res[0][2], rel=1e-5)
        # the same row halves with numerator AND denominator slabs (beta = 2), fp32-grade mode, against the oracle
        os.environ['TORCHNMF_AMD_AR_OVERLAP'] = '1'
        Vc, Wc, Hc = Vb.cpu()[:, :500], Wb[:500, :24], Hb[:, :24]
        W, H = Wc.clone().to(dev), Hc.clone().to(dev)
        eng = DenseMU(Vc.to(dev).contiguous(), W, H, 2.0, 0.05, 0.05, precision='bf16x3', group=dist.group.WORLD)
        assert eng._h_rows is not None
        for _ in range(3):
            eng.w_step()
            eng.h_step()
        Wr, Hr = Wc, Hc
        for _ in range(3):
            Wr = O.nmf_w_step(Vc, Wr, Hr, 2, 1.0, 0.05, 0.05)
            Hr = O.nmf_h_step(Vc, Wr, Hr, 2, 1.0, 0.05, 0.05)
        assert rel_err(W.cpu(), Wr) < TOL and rel_err(H.cpu(), Hr) < TOL
        # ... and through fit(allreduce='direct'): numerator and denominator in ONE all-reduce from the C entry
        m = NMF(W=Wc, H=Hc).to(dev)
        assert m.fit(Vc.to(dev).contiguous(), 2, NO_STOP, 3, alpha=0.1, l1_ratio=0.5, precision='bf16x3',
                     process_group=dist.group.WORLD, allreduce='direct') == 3
        assert rel_err(m.W.data.cpu(), Wr) < TOL and rel_err(m.H.data.cpu(), Hr) < TOL
        # ---- configs[4]'s kernel family on the sharded path: rank 256, fp16 operands (the parity-grade
        # single-plane mode), an 8192 x 8192 slice of the per-GPU shard, two iterations through NMF.fit with
        # precision='auto' -- which must pick 'f16' here (fp16-exact target, both dimensions >= 4096) -- and the two
        # row halves of the H half-step with their own all-reduces
        gg = torch.Generator().manual_seed(19)
        N5, C5, R5 = 8192, 8192, 256
        V5 = torch.rand(N5, C5, generator=gg).half().float()
        W5, H5 = torch.randn(C5, R5, generator=gg).abs(), torch.randn(N5, R5, generator=gg).abs()
        picked = []
        orig_init = DenseMU.__init__

        def spy(self, *a, **k):
            orig_init(self, *a, **k)
            picked.append((self.precision_name, self._h_rows is not None))
        DenseMU.__init__ = spy
        try:
            m = NMF(W=W5, H=H5).to(dev)
            n = m.fit(V5.to(dev), 1, NO_STOP, 2, process_group=dist.group.WORLD, allreduce='overlap')
            # nothing chosen (no keyword, no environment): ONE all-reduce per iteration, north_star's form (round 5 default)
            os.environ.pop('TORCHNMF_AMD_AR_OVERLAP', None)
            m1 = NMF(W=W5, H=H5).to(dev)
            assert m1.fit(V5.to(dev), 1, NO_STOP, 2, process_group=dist.group.WORLD) == 2
        finally:
            DenseMU.__init__ = orig_init
        assert n == 2 and picked == [('f16', True), ('f16', False)], picked
        assert rel_err(m1.W.data, m.W.data) < 2e-5 and rel_err(m1.H.data, m.H.data) < 2e-5
        Wr, Hr = W5, H5
        for _ in range(2):
            Wr = O.nmf_w_step(V5, Wr, Hr, 1, 1.0)
            Hr = O.nmf_h_step(V5, Wr, Hr, 1, 1.0)
        ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
        record('sharded_world1_rank256_f16', relW=ew, relH=eh)
        assert ew < TOL and eh < TOL, (ew, eh)
        # sharded 'auto' at rank 129..256 without an admissible fp16 mode raises instead of falling to plain bf16
        Vs = torch.rand(300, 500, generator=gg)
        with pytest.raises(NotImplementedError):
            NMF(Vs.shape, 200).to(dev).fit(Vs.to(dev), max_iter=2, process_group=dist.group.WORLD)
        assert NMF(Vs.shape, 200).to(dev).fit(Vs.to(dev), max_iter=2, precision='bf16', process_group=dist.group.WORLD) == 2
    finally:
        os.environ.pop('TORCHNMF_AMD_AR_OVERLAP', None)
        dist.destroy_process_group()


def test_comm_entries_world1_rccl(dev):
    """The C entries of the sharded path's collective (include/nmfmu.h: nmfmu_comm_*, RCCL resolved by dlopen): a
    communicator of one rank by unique id, the in-place fp32 sum of a packed [numerator | denominator] buffer on a side
    stream, the one-process-many-devices form with one device, teardown."""
    import ctypes as C
    from torchnmf_amd import _capi
    lib = _capi.load()
    assert lib.nmfmu_comm_available() == 1
    uid = (C.c_char * 128)()
    _capi.check(lib.nmfmu_comm_unique_id(uid), 'nmfmu_comm_unique_id')
    comm = C.c_void_p()
    _capi.check(lib.nmfmu_comm_init_rank(C.byref(comm), 1, uid, 0), 'nmfmu_comm_init_rank')
    assert lib.nmfmu_comm_nranks(comm) == 1
    buf = torch.arange(4096 * 128 + 128, dtype=torch.float32, device=dev)
    want = buf.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    _capi.check(lib.nmfmu_comm_allreduce_sum_f32(comm, buf.data_ptr(), buf.numel(), side.cuda_stream), 'allreduce')
    side.synchronize()
    assert torch.equal(buf, want)
    _capi.check(lib.nmfmu_comm_destroy(comm), 'nmfmu_comm_destroy')
    comms = (C.c_void_p * 1)()
    _capi.check(lib.nmfmu_comm_init_all(comms, 1, None), 'nmfmu_comm_init_all')
    bufs, streams = (C.c_void_p * 1)(buf.data_ptr()), (C.c_void_p * 1)(torch.cuda.current_stream().cuda_stream)
    _capi.check(lib.nmfmu_comm_allreduce_sum_f32_multi(comms, bufs, buf.numel(), streams, 1), 'allreduce_multi')
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    _capi.check(lib.nmfmu_comm_destroy(comms[0]), 'nmfmu_comm_destroy')
    assert lib.nmfmu_comm_allreduce_sum_f32(None, buf.data_ptr(), 4, None) == _capi.ERR_ARG


# ----------------------------------------------------------------------------------------------------------
# wide ranks (padded rank 256, the configs[4] kernel family) and unsupported combinations
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('rank', [200, 256])
@pytest.mark.parametrize('beta', [1, 2, 0.5])
@pytest.mark.parametrize('prec,tol', [('bf16', 5e-3), ('f16', TOL)])
def test_rank_above_128_single_plane(dev, rank, beta, prec, tol):
    """Padded rank 256 (the configs[4] kernel family) with one operand plane: bf16, and fp16 at the parity bar."""
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(rank)
    N, C = 300, 700
    V = torch.rand(N, C, generator=g).bfloat16().float()
    W0 = torch.randn(C, rank, generator=g).abs()
    H0 = torch.randn(N, rank, generator=g).abs()
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, prec, 1)
    gam = O.gamma_of(beta)
    Wr = O.nmf_w_step(V, W0, H0, beta, gam)
    Hr = O.nmf_h_step(V, Wr, H0, beta, gam)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('rank_above_128', rank=rank, beta=beta, prec=prec, relW=ew, relH=eh)
    assert ew < tol and eh < tol, (ew, eh)
    assert l1 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, beta)), rel=50 * tol)


def test_unsupported_combinations_raise(dev):
    from torchnmf_amd.nmf import NMF
    V = torch.rand(64, 80)
    from torchnmf_amd.engine import DenseMU
    m = NMF(V.shape, 200).to(dev)
    with pytest.raises(NotImplementedError):       # the fused kernel's split precision stops at padded rank 128 ...
        DenseMU(V.to(dev), m.W.data, m.H.data, 1.0, precision='bf16x3')
    assert m.fit(V.to(dev), max_iter=3, precision='bf16x3') == 3   # ... so fit() takes the GEMM engine there,
    assert m.fit(V.to(dev), max_iter=3) == 3                       # which is also what 'auto' means above rank 128
    assert m.fit(V.to(dev), max_iter=3, precision='bf16') == 3     # the fast fused kernel on request
    assert NMF(V.shape, 300).to(dev).fit(V.to(dev), max_iter=3) == 3   # rank > 256: GEMM engine (WideRankMU)


# ----------------------------------------------------------------------------------------------------------
# trainer.BetaMu on one NMF layer -- reference outputs in g7_betamu
# ----------------------------------------------------------------------------------------------------------
def _g7_cases():
    return [str(c) for c in load_golden('g7_betamu')['cases']]


@pytest.mark.parametrize('case', _g7_cases())
def test_betamu_g7_golden(dev, case):
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    g = load_golden('g7_betamu')
    b, pen, which = case.split('_')
    l1, l2, ortho = {'plain': (0, 0, 0), 'pen': (1e-3, 1e-3, 1e-2)}[pen]
    m = NMF(W=t(g['W0']), H=t(g['H0'])).to(dev)
    params = list(m.parameters()) if which == 'both' else [getattr(m, which)]
    trainer = BetaMu(params, float(b[1:]), l1, l2, ortho, precision='bf16x3')
    V = t(g['V']).to(dev)

    def closure():
        trainer.zero_grad()
        return V, m()          # the reference's closure form (tests/test_trainer.py:66-68)
    for it in range(1, 6):
        trainer.step(closure)
        if it in (1, 5):
            assert rel_err(m.W.data.cpu(), g[f'{case}_W{it}']) < TOL
            assert rel_err(m.H.data.cpu(), g[f'{case}_H{it}']) < TOL
        if it == 1:
            last = 'H' if which in ('both', 'H') else 'W'
            # p.grad = pos - neg is a difference of two nearly equal sums near a fixed point: compare on the scale
            # of its terms (|pos| + |neg|), which is what 1e-4 relative on pos and neg separately amounts to
            got, want = getattr(m, last).grad.cpu(), t(g[f'{case}_grad{last}1'])
            assert float((got - want).norm() / want.norm()) < 2e-3
    assert m.W.requires_grad and m.H.requires_grad


def _g14_cases():
    return [str(c) for c in load_golden('g14_betamu_conv')['cases']]


@pytest.mark.parametrize('case', _g14_cases())
def test_betamu_conv_g14_golden(dev, case):
    """trainer.BetaMu over ONE convolutive layer (NMFD / NMF2D / NMF3D; VERDICT r5 Missing 3): the reference's outputs
    (tools/make_golden.py g14) after 1 and 3 steps of both parameters, beta in {0.5, 1, 2}, plain and with l1 / l2 /
    orthogonality penalties; p.grad of the step's last parameter on the scale of its terms."""
    from torchnmf_amd import nmf as anmf
    from torchnmf_amd.trainer import BetaMu
    g = load_golden('g14_betamu_conv')
    name, bs, pen = case.split('_')
    l1, l2, ortho = {'plain': (0, 0, 0), 'pen': (1e-3, 1e-3, 1e-2)}[pen]
    cls = {'1d': anmf.NMFD, '2d': anmf.NMF2D, '3d': anmf.NMF3D}[name]
    m = cls(W=t(g[f'{name}_W0']), H=t(g[f'{name}_H0'])).to(dev)
    trainer = BetaMu(m.parameters(), float(bs[1:]), l1, l2, ortho)
    V = t(g[f'{name}_V']).to(dev)

    def closure():
        trainer.zero_grad()
        return V, m()
    for it in range(1, 4):
        trainer.step(closure)
        if it in (1, 3):
            ew, eh = rel_err(m.W.data.cpu(), g[f'{case}_W{it}']), rel_err(m.H.data.cpu(), g[f'{case}_H{it}'])
            record('betamu_conv_g14', case=case, it=it, relW=ew, relH=eh)
            assert ew < TOL and eh < TOL, (it, ew, eh)
        if it == 1:
            got, want = m.H.grad.cpu(), t(g[f'{case}_gradH1'])
            assert float((got - want).norm() / want.norm()) < 2e-3
    assert trainer.last_precision == 'bf16x3' and m.W.requires_grad and m.H.requires_grad


def test_betamu_conv_layer_forms(dev):
    """The deferred closure form (``return V, m``), one parameter only, an outside edit of a factor between steps (the
    binding refreshes its operand images), and the refusal of a convolutive layer inside a chain."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF, NMFD
    from torchnmf_amd.trainer import BetaMu
    g = torch.Generator().manual_seed(14)
    V = torch.rand(2, 40, 96, generator=g)
    W0, H0 = torch.randn(40, 5, 8, generator=g).abs(), torch.randn(2, 5, 89, generator=g).abs()
    m = NMFD(W=W0.clone(), H=H0.clone()).to(dev)
    trainer = BetaMu([m.W], 1, 0, 1e-3, 0)
    Vd = V.to(dev)
    trainer.step(lambda: (Vd, m))
    Wr, _, _ = O.betamu_conv_step(V, W0, H0, 1, 0, 1e-3, 0, params=('W',))
    assert rel_err(m.W.data.cpu(), Wr) < TOL and torch.equal(m.H.data.cpu(), H0)
    with torch.no_grad():
        m.H.mul_(1.5)                                   # outside edit: bumps H._version
    trainer.step(lambda: (Vd, m))
    Wr2, _, _ = O.betamu_conv_step(V, Wr, H0 * 1.5, 1, 0, 1e-3, 0, params=('W',))
    assert rel_err(m.W.data.cpu(), Wr2) < TOL
    chain = torch.nn.Sequential(NMF((6, 4), 3), NMF((4, 5), 4)).to(dev)
    mixed = BetaMu(m.parameters(), 1)
    pred = m()
    pred._nmf_source = (m, chain(None), m.W)            # a convolutive layer fed by another layer's output
    with pytest.raises(NotImplementedError):
        mixed.step(lambda: (Vd, pred))


@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
@pytest.mark.parametrize('attr', ['W', 'H'])
def test_betamu_grad_is_beta_div_gradient(dev, beta, attr):
    """tests/test_trainer.py:54-73 of the reference: after one step p.grad equals the gradient of beta_div(m(), V)
    at the factors the step STARTED from; here the gradient comes from the oracle's closed form."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    g = torch.Generator().manual_seed(77)
    m = NMF((100, 50)).to(dev)                   # rank defaults to K = 50 (nmf.py:685)
    W0, H0 = m.W.data.cpu().clone(), m.H.data.cpu().clone()
    V = torch.rand(100, 50, generator=g) + 1e-3
    trainer = BetaMu([getattr(m, attr)], beta, precision='bf16x3')
    Vd = V.to(dev)

    def closure():
        trainer.zero_grad()
        return Vd, m
    trainer.step(closure)
    Wn, Hn, grads = O.betamu_step(V, W0, H0, beta, params=(attr,))
    got = getattr(m, attr).grad.cpu()
    scale = float(grads[attr].abs().max())
    assert float((got - grads[attr]).abs().max()) < 1e-4 * max(scale, 1.0) * 50
    assert rel_err(getattr(m, attr).data.cpu(), Wn if attr == 'W' else Hn) < TOL
    assert bool(torch.all(getattr(m, attr).data >= 0))


@pytest.mark.parametrize('rank', [200, 300])
@pytest.mark.parametrize('beta', [1, 0.5])
def test_betamu_default_precision_above_rank_128(dev, rank, beta):
    """BetaMu's default precision ('auto') has no fused mode above rank 128 (split bf16 stops there and the
    fp16 modes are not admitted for the trainer); such a layer -- and one wider than the kernels' 256 -- must take the
    exact chain path, not raise: the reference's BetaMu has no rank limit (trainer.py:72-112)."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    g = torch.Generator().manual_seed(rank)
    V = torch.rand(260, 340, generator=g) + 1e-3
    W0 = torch.randn(340, rank, generator=g).abs()
    H0 = torch.randn(260, rank, generator=g).abs()
    m = NMF(W=W0, H=H0).to(dev)
    trainer = BetaMu(m.parameters(), beta)
    Vd = V.to(dev)

    def closure():
        trainer.zero_grad()
        return Vd, m()
    trainer.step(closure)
    Wn, Hn, _ = O.betamu_step(V, W0, H0, beta)
    ew, eh = rel_err(m.W.data.cpu(), Wn), rel_err(m.H.data.cpu(), Hn)
    assert ew < TOL and eh < TOL, (ew, eh)


@pytest.mark.parametrize('exact,rank,beta,want', [(False, 128, 1, 'f16r'), (True, 128, 1, 'f16'), (False, 128, 2, 'f16x'),
                                                  (False, 200, 1, 'f16r'), (True, 64, 0.5, 'f16')])
def test_betamu_auto_takes_the_1x_modes(dev, exact, rank, beta, want):
    """BetaMu's 'auto' resolves like NMF.fit's -- fp16 operands at 1x MFMA work where both dimensions
    reach 4096 and the data fit fp16's range ('f16' for an fp16-exact target, 'f16x' otherwise), also at rank 129..256
    where the alternative is the exact chain path.  Three trainer steps at 4096 x 4096 against the oracle's
    trainer.py:72-112, 1e-4 bar; p.grad = pos - neg (trainer.py:98) checked on the last step."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    g = torch.Generator().manual_seed(4096 + rank)
    V = torch.rand(4096, 4096, generator=g) + 1e-3
    if exact:
        V = V.half().float()
    W0 = torch.randn(4096, rank, generator=g).abs()
    H0 = torch.randn(4096, rank, generator=g).abs()
    m = NMF(W=W0, H=H0).to(dev)
    trainer = BetaMu(m.parameters(), beta)
    Vd = V.to(dev)

    def closure():
        trainer.zero_grad()
        return Vd, m
    Wn, Hn = W0, H0
    for _ in range(3):
        trainer.step(closure)
        Wn, Hn, grads = O.betamu_step(V, Wn, Hn, beta)
    assert trainer.last_precision == want, trainer.last_precision
    ew, eh = rel_err(m.W.data.cpu(), Wn), rel_err(m.H.data.cpu(), Hn)
    assert ew < TOL and eh < TOL, (ew, eh)
    # p.grad of the LAST parameter (the closure's zero_grad() clears the earlier ones, as in the reference); the gradient is
    # a difference of two near-equal contractions: compare on the scale of its own magnitude
    scale = float(grads['H'].abs().mean()) + 1e-30
    assert float((m.H.grad.cpu() - grads['H']).abs().mean()) / scale < 5e-2


def test_betamu_rejects_general_graphs_and_cpu_tensors(dev):
    from torchnmf_amd import _capi
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    m = NMF((30, 20), 4).to(dev)
    V = torch.rand(30, 20, device=dev)
    tr = BetaMu(m.parameters())
    with pytest.raises(NotImplementedError):
        tr.step(lambda: (V, m() * 2.0))          # arithmetic on the prediction: not a single-layer graph
    with pytest.raises(_capi.NmfmuError):
        tr.step(lambda: (V.cpu(), m))            # no CPU fallback
    tr.step(lambda: (V, m()))
    assert bool(torch.all(m.W >= 0)) and bool(torch.all(m.H >= 0))


# ----------------------------------------------------------------------------------------------------------
# sparse-COO target: nmf.py:351-398, 602-638
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('beta', [1, 2])
@pytest.mark.parametrize('tag,args', [('run', (NO_STOP, 25, 0.0, 0.0)), ('reg', (NO_STOP, 10, 0.1, 0.5)),
                                      ('stop', (1e-3, 200, 0.0, 0.0))])
def test_sparse_fit_g9_golden(dev, beta, tag, args):
    from torchnmf_amd.nmf import NMF
    g = load_golden('g9_sparse')
    V = torch.sparse_coo_tensor(t(g['indices']), t(g['values']), tuple(g['shape'])).to(dev)
    m = NMF(W=t(g['W0']), H=t(g['H0'])).to(dev)
    tol, it, alpha, l1r = args
    n = m.fit(V, beta, tol, it, alpha=alpha, l1_ratio=l1r)
    assert n == int(g[f'b{beta}_{tag}_n'])
    assert rel_err(m.W.data.cpu(), g[f'b{beta}_{tag}_W']) < TOL and rel_err(m.H.data.cpu(), g[f'b{beta}_{tag}_H']) < TOL


@pytest.mark.parametrize('beta', [1, 2])
@pytest.mark.parametrize('rank', [16, 100, 200])
def test_fit_sparse_equals_dense(dev, beta, rank):
    """The reference's own sparse test (tests/test_nmf_sparse.py:8-37) on the device: same factors from the sparse
    and the dense representation of one matrix (dense path in its fp32-grade mode where available)."""
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(rank)
    Vd = torch.rand(500, 700, generator=g)
    Vd = torch.where(Vd > 0.93, Vd, torch.zeros(()))
    W0, H0 = torch.randn(700, rank, generator=g).abs(), torch.randn(500, rank, generator=g).abs()
    ms, md = NMF(W=W0, H=H0).to(dev), NMF(W=W0, H=H0).to(dev)
    ms.fit(Vd.to_sparse().to(dev), beta, 0, 5, alpha=0.1, l1_ratio=0.5)
    md.fit(Vd.to(dev), beta, 0, 5, alpha=0.1, l1_ratio=0.5, precision='bf16x3' if rank <= 128 else 'bf16')
    tol = TOL if rank <= 128 else 2e-2
    assert rel_err(ms.W.data.cpu(), md.W.data.cpu()) < tol and rel_err(ms.H.data.cpu(), md.H.data.cpu()) < tol


def test_sparse_fit_errors(dev):
    from torchnmf_amd.nmf import NMF, NMFD
    V = torch.rand(30, 20)
    Vs = torch.where(V > 0.5, V, torch.zeros(())).to_sparse().to(dev)
    m = NMF((30, 20), 4).to(dev)
    with pytest.raises(ValueError):                # a sparse target always contains zeros (nmf.py:332-336)
        m.fit(Vs, beta=0)
    with pytest.raises(AssertionError):
        m.fit((-Vs).coalesce())
    with pytest.raises(NotImplementedError):
        NMFD((1, 20, 30), 3, 2).to(dev).fit(torch.rand(1, 20, 30).to_sparse().to(dev))
    assert m.fit(Vs, beta=1, max_iter=12) <= 12 and bool(torch.all(m.W >= 0))


@pytest.mark.parametrize('beta', [1, 2, 0.5])
def test_rank_above_256_runs_on_the_gemm_engine(dev, beta):
    """Ranks beyond the fused kernel's 256 (the reference has no limit): NMF as the T = 1 member of the NMFD family."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(300)
    N, C, R = 350, 420, 300
    V = torch.rand(N, C, generator=g) + 1e-3
    W0, H0 = torch.randn(C, R, generator=g).abs(), torch.randn(N, R, generator=g).abs()
    m = NMF(W=W0, H=H0).to(dev)
    n = m.fit(V.to(dev), beta, 1e-4, 30, alpha=0.05, l1_ratio=0.5)
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, beta, 1e-4, 30, 0.05, 0.5)
    assert n == nr and rel_err(m.W.data.cpu(), Wr) < TOL and rel_err(m.H.data.cpu(), Hr) < TOL
    assert rel_err(m().cpu(), Hr @ Wr.t()) < TOL


# ----------------------------------------------------------------------------------------------------------
# PLCA: plca.py:193-373
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,ctor,fitkw', [
    ('plain', {}, {}), ('prior', {}, dict(W_alpha=1.02, H_alpha=0.99, Z_alpha=1.01)), ('frozenZ', dict(trainable_Z=False), {}),
    ('frozenW', dict(trainable_W=False), {}), ('stop', {}, dict(tol=1e-3, max_iter=200))])
def test_plca_fit_g10_golden(dev, name, ctor, fitkw):
    from torchnmf_amd.plca import PLCA
    g = load_golden('g10_plca')
    m = PLCA(W=t(g['W0']), H=t(g['H0']), Z=t(g['Z0']), **ctor).to(dev)
    if name == 'plain':
        assert rel_err(m().cpu(), g['recon_init']) < 1e-5 and rel_err(m(norm=3.0).cpu(), 3.0 * g['recon_init']) < 1e-5
    fitkw = dict(fitkw)
    fitkw.setdefault('tol', NO_STOP)
    fitkw.setdefault('max_iter', 30)
    n, norm = m.fit(t(g['V']).to(dev), **fitkw)
    assert n == int(g[f'{name}_n']) and float(norm) == pytest.approx(float(g[f'{name}_norm']), rel=1e-5)
    for p, k in ((m.W, 'W'), (m.H, 'H'), (m.Z, 'Z')):
        assert rel_err(p.data.cpu(), g[f'{name}_{k}']) < TOL, (k, rel_err(p.data.cpu(), g[f'{name}_{k}']))


def test_plca_fit_g13_tensor_alphas_golden(dev):
    """One-element tensor Dirichlet hyper-parameters (plca.py:197-199) against the reference's own run."""
    from torchnmf_amd.plca import PLCA
    g = load_golden('g13_plca_tensor_alpha')
    m = PLCA(W=t(g['W0']), H=t(g['H0']), Z=t(g['Z0'])).to(dev)
    wa, ha, za = (float(x) for x in g['alphas'])
    n, norm = m.fit(t(g['V']).to(dev), tol=NO_STOP, max_iter=30, W_alpha=torch.tensor(wa), H_alpha=torch.tensor([ha], device=dev),
                    Z_alpha=torch.tensor([za]))
    assert n == int(g['n']) and float(norm) == pytest.approx(float(g['norm']), rel=1e-5)
    for p, k in ((m.W, 'W'), (m.H, 'H'), (m.Z, 'Z')):
        assert rel_err(p.data.cpu(), g[k]) < TOL, (k, rel_err(p.data.cpu(), g[k]))
    with pytest.raises(NotImplementedError, match="f16x"):
        m.fit(t(g['V']).to(dev), max_iter=1, precision='f16x')
    with pytest.raises(NotImplementedError, match="below fp16's range"):      # ('f16' as well)
        m.fit(t(g['V']).to(dev), max_iter=1, precision='f16')
    #  the hyper-parameter of a FROZEN factor is never evaluated by the reference: a multi-element tensor passes there
    m2 = PLCA(W=t(g['W0']), H=t(g['H0']), Z=t(g['Z0']), trainable_W=False).to(dev)
    m2.fit(t(g['V']).to(dev), max_iter=2, W_alpha=torch.ones(3))


@pytest.mark.parametrize('rank,prec', [(5, 'bf16x3'), (100, 'bf16x3'), (100, 'bf16'), (200, None)])
def test_plca_medium_against_oracle(dev, rank, prec):
    from oracle import mu_oracle as O
    from torchnmf_amd.plca import PLCA
    g = torch.Generator().manual_seed(rank)
    N, C = 330, 520
    V = torch.rand(N, C, generator=g)
    W0, H0, Z0 = torch.rand(C, rank, generator=g), torch.rand(N, rank, generator=g), torch.rand(rank, generator=g)
    m = PLCA(W=W0, H=H0, Z=Z0).to(dev)
    n, norm = m.fit(V.to(dev), NO_STOP, 6, W_alpha=1.001, precision=prec)
    Wr, Hr, Zr, nr, _, _ = O.plca_fit(V, W0, H0, Z0, NO_STOP, 6, W_alpha=1.001)
    tol = TOL if prec == 'bf16x3' else 2e-2
    assert n == nr
    for p, ref in ((m.W, Wr), (m.H, Hr), (m.Z, Zr)):
        assert rel_err(p.data.cpu(), ref) < tol, rel_err(p.data.cpu(), ref)
    assert m.W.data.sum(0).cpu() == pytest.approx(torch.ones(rank).numpy(), rel=1e-4)


def test_fit_accepts_strided_and_non_fp32_targets(dev):
    """A transposed view, a row-strided slice and a float64 target give the same factors as their contiguous fp32 copy."""
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(4)
    base = torch.rand(300, 210, generator=g)
    W0, H0 = torch.rand(300, 12, generator=g), torch.rand(210, 12, generator=g)
    Vt = base.t()                                    # (210, 300) view with stride(1) != 1
    big = torch.rand(210, 640, generator=g)
    views = {'transposed': (Vt.to(dev).t().t() if False else base.to(dev).t(), Vt.contiguous()),
             'row_strided': (big.to(dev)[:, :300], big[:, :300].contiguous()),
             'float64': (Vt.contiguous().double().to(dev), Vt.contiguous())}
    for name, (v_dev, v_ref) in views.items():
        m1, m2 = NMF(W=W0, H=H0).to(dev), NMF(W=W0, H=H0).to(dev)
        m1.fit(v_dev, 1, NO_STOP, 5)
        m2.fit(v_ref.to(dev), 1, NO_STOP, 5)
        assert torch.equal(m1.W.data, m2.W.data) and torch.equal(m1.H.data, m2.H.data), name


@pytest.mark.parametrize('n', [1, 2, 1000, 70001])
def test_metrics_sparseness(dev, n):
    """metrics.sparseness (metrics.py:99-115): 0 for a constant vector, 1 for a one-hot one, the closed form between."""
    from torchnmf_amd.metrics import sparseness
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    if n > 1:
        want = (n ** 0.5 - x.norm(1) / x.norm(2)) / (n ** 0.5 - 1)
        assert float(sparseness(x.to(dev))) == pytest.approx(float(want), rel=1e-5, abs=1e-6)
        assert float(sparseness(torch.ones(n, device=dev))) == pytest.approx(0.0, abs=1e-5)
        hot = torch.zeros(n, device=dev)
        hot[n // 2] = 3.0
        assert float(sparseness(hot)) == pytest.approx(1.0, abs=1e-6)
        assert float(sparseness(x.reshape(-1, 1).repeat(1, 2).to(dev))) == pytest.approx(
            float((( 2 * n) ** 0.5 - (x.norm(1) * 2) / (x.norm(2) * 2 ** 0.5)) / ((2 * n) ** 0.5 - 1)), rel=1e-5, abs=1e-6)


@pytest.mark.parametrize('name,cls', [('1d', 'SIPLCA'), ('2d', 'SIPLCA2'), ('3d', 'SIPLCA3')])
@pytest.mark.parametrize('case,ctor,fitkw', [('plain', {}, {}), ('prior', {}, dict(W_alpha=1.02, H_alpha=0.99, Z_alpha=1.01)),
                                             ('frozenZ', dict(trainable_Z=False), {})])
def test_siplca_fit_g11_golden(dev, name, cls, case, ctor, fitkw):
    """Shift-invariant PLCA (plca.py:376-606) against the reference's outputs."""
    from torchnmf_amd import plca
    g = load_golden('g11_siplca')
    m = getattr(plca, cls)(W=t(g[f'{name}_W0']), H=t(g[f'{name}_H0']), Z=t(g[f'{name}_Z0']), **ctor).to(dev)
    if case == 'plain':
        assert rel_err(m().cpu(), g[f'{name}_recon_init']) < 1e-5
    n, norm = m.fit(t(g[f'{name}_V']).to(dev), tol=NO_STOP, max_iter=20, **fitkw)
    assert n == int(g[f'{name}_{case}_n']) and float(norm) == pytest.approx(float(g[f'{name}_{case}_norm']), rel=1e-5)
    for p, k in ((m.W, 'W'), (m.H, 'H'), (m.Z, 'Z')):
        assert rel_err(p.data.cpu(), g[f'{name}_{case}_{k}']) < TOL, (k, rel_err(p.data.cpu(), g[f'{name}_{case}_{k}']))


def test_siplca_implicit_operands_against_oracle(dev):
    """Taps and frames that are multiples of 8: the reconstruction operand comes from the window tables of H."""
    from oracle import mu_oracle as O
    from torchnmf_amd.plca import SIPLCA
    g = torch.Generator().manual_seed(88)
    V = torch.rand(2, 70, 304, generator=g)
    W0, H0, Z0 = torch.rand(70, 5, 16, generator=g), torch.rand(2, 5, 289, generator=g), torch.rand(5, generator=g)
    m = SIPLCA(W=W0, H=H0, Z=Z0).to(dev)
    n, _ = m.fit(V.to(dev), tol=NO_STOP, max_iter=4, H_alpha=1.001)
    Wr, Hr, Zr, nr, _, _ = O.plca_fit(V, W0, H0, Z0, tol=NO_STOP, max_iter=4, H_alpha=1.001)
    assert n == nr
    for p, ref in ((m.W, Wr), (m.H, Hr), (m.Z, Zr)):
        assert rel_err(p.data.cpu(), ref) < TOL


@pytest.mark.parametrize('cls,shape', [('SIPLCA2', (2, 6, (12, 24), 3, (3, 8))), ('SIPLCA3', (1, 70, (5, 6, 16), 2, (2, 2, 8))),
                                       ('SIPLCA2', (1, 5, (9, 11), 4, (2, 3)))])
def test_siplca_several_shift_axes_on_window_tables_against_oracle(dev, cls, shape, monkeypatch):
    """SIPLCA2 / SIPLCA3 (plca.py:452-606): with taps and frames of the last axis multiples of 8 the EM step's GEMMs take
    their H operands from the window tables of nmfmu_convnd_tables; (G W) comes from the window-operand GEMM over shifted
    rows of the ratio planes (no Y, no fold); the (G^T H) GEMM is contraction-split.  Against the oracle, and the third
    shape (unaligned) on explicit operands against the store-then-fold path."""
    from oracle import mu_oracle as O
    from torchnmf_amd import plca
    B, Cc, ls, R, ks = shape
    g = torch.Generator().manual_seed(sum(ls) + R)
    V = torch.rand(B, Cc, *ls, generator=g)
    W0, H0 = torch.rand(Cc, R, *ks, generator=g), torch.rand(B, R, *[l - k + 1 for l, k in zip(ls, ks)], generator=g)
    Z0 = torch.rand(R, generator=g)
    Wr, Hr, Zr, nr, _, _ = O.plca_fit(V, W0, H0, Z0, tol=NO_STOP, max_iter=4, W_alpha=1.001)
    res = {}
    for rows in ('1', '0'):
        monkeypatch.setenv('TORCHNMF_AMD_NMFD_H_ROWS', rows)
        m = getattr(plca, cls)(W=W0, H=H0, Z=Z0).to(dev)
        n, _ = m.fit(V.to(dev), tol=NO_STOP, max_iter=4, W_alpha=1.001)
        assert n == nr
        for p, ref in ((m.W, Wr), (m.H, Hr), (m.Z, Zr)):
            assert rel_err(p.data.cpu(), ref) < TOL
        res[rows] = m.H.data.cpu()
    assert rel_err(res['1'], res['0']) < 2e-5


@pytest.mark.parametrize('rank', [129, 200])
def test_auto_precision_meets_the_parity_bar_above_rank_128(dev, rank):
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(rank)
    V = torch.rand(260, 330, generator=g) + 1e-3
    W0, H0 = torch.randn(330, rank, generator=g).abs(), torch.randn(260, rank, generator=g).abs()
    m = NMF(W=W0, H=H0).to(dev)
    n = m.fit(V.to(dev), 1, NO_STOP, 5)
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 5)
    assert n == nr and rel_err(m.W.data.cpu(), Wr) < TOL and rel_err(m.H.data.cpu(), Hr) < TOL


@pytest.mark.parametrize('beta', [0.5, 1.5, 3])
@pytest.mark.parametrize('tag,args', [('run', (NO_STOP, 20, 0.0, 0.0)), ('reg', (NO_STOP, 10, 0.1, 0.5))])
def test_sparse_fit_generic_beta_g9_golden(dev, beta, tag, args):
    """Generic beta on a sparse target: gather kernel for the numerator, target-less fused pass for the dense positive
    term (nmfmu_den_partial), target-less loss."""
    from torchnmf_amd.nmf import NMF
    g = load_golden('g9_sparse')
    V = torch.sparse_coo_tensor(t(g['indices']), t(g['values']), tuple(g['shape'])).to(dev)
    m = NMF(W=t(g['W0']), H=t(g['H0'])).to(dev)
    tol, it, alpha, l1r = args
    n = m.fit(V, beta, tol, it, alpha=alpha, l1_ratio=l1r)
    assert n == int(g[f'b{beta}_{tag}_n'])
    assert rel_err(m.W.data.cpu(), g[f'b{beta}_{tag}_W']) < TOL and rel_err(m.H.data.cpu(), g[f'b{beta}_{tag}_H']) < TOL


@pytest.mark.parametrize('beta', [0.5, 1.5, 3])
def test_fit_sparse_equals_dense_generic_beta(dev, beta):
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(int(beta * 10))
    Vd = torch.rand(400, 900, generator=g)
    Vd = torch.where(Vd > 0.9, Vd, torch.zeros(()))
    W0, H0 = torch.randn(900, 40, generator=g).abs(), torch.randn(400, 40, generator=g).abs()
    ms, md = NMF(W=W0, H=H0).to(dev), NMF(W=W0, H=H0).to(dev)
    ms.fit(Vd.to_sparse().to(dev), beta, 0, 5)
    md.fit(Vd.to(dev), beta, 0, 5, precision='bf16x3')
    assert rel_err(ms.W.data.cpu(), md.W.data.cpu()) < TOL and rel_err(ms.H.data.cpu(), md.H.data.cpu()) < TOL


# ----------------------------------------------------------------------------------------------------------
# trainer.BetaMu over a chain of layers: the reference's own trainer test scenario
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('beta', [0.5, 1, 2])
@pytest.mark.parametrize('pen', ['plain', 'pen'])
def test_betamu_chain_g12_golden(dev, beta, pen):
    from torch import nn
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    g = load_golden('g12_betamu_chain')
    l1, l2, ortho = {'plain': (0, 0, 0), 'pen': (1e-3, 1e-3, 1e-2)}[pen]
    m = nn.Sequential(NMF(W=t(g['W1']), H=t(g['H1'])), NMF(W=t(g['W2'])), NMF(W=t(g['W3']))).to(dev)
    trainer = BetaMu(m.parameters(), beta, l1, l2, ortho)
    V = t(g['V']).to(dev)

    def closure():
        trainer.zero_grad()
        return V, m(None)
    for it in range(1, 6):
        trainer.step(closure)
        if it in (1, 5):
            for pn, p in (('W1', m[0].W), ('H1', m[0].H), ('W2', m[1].W), ('W3', m[2].W)):
                assert rel_err(p.data.cpu(), g[f'b{beta}_{pen}_{pn}_{it}']) < 2e-4, (pn, it)
    assert rel_err(m[2].W.grad.cpu(), g[f'b{beta}_{pen}_gradW3']) < 2e-3


@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
@pytest.mark.parametrize('l1_reg,l2_reg,orthogonal', [(0, 0, 0), (1e-3, 1e-3, 1e-2)])
def test_beta_trainer_like_reference(dev, beta, l1_reg, l2_reg, orthogonal):
    """tests/test_trainer.py:10-32 of the reference, on the device: three stacked layers stay non-negative."""
    from torch import nn
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    m = nn.Sequential(NMF((100, 16), rank=8), NMF(W=(32, 16)), NMF(W=(50, 32))).to(dev)
    target = torch.rand(100, 50, device=dev)
    trainer = BetaMu(m.parameters(), beta, l1_reg, l2_reg, orthogonal)

    def closure():
        trainer.zero_grad()
        return target, m(None)
    for _ in range(10):
        trainer.step(closure)
        for p in m.parameters():
            assert bool(torch.all(p >= 0.)) and bool(torch.isfinite(p).all())


# ----------------------------------------------------------------------------------------------------------
# round 2: the configuration holes VERDICT r1 named, the auto-precision policy, ADVICE r1 scenarios
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('beta', [2, 0, 0.5])
@pytest.mark.parametrize('prec,tol', [('bf16', 5e-3), ('f16', TOL)])
def test_rank128_single_plane_beta_sweep_half_steps(dev, beta, prec, tol):
    """BASELINE configs[2] instantiations: <rank pad 128, beta in {2, 0, 0.5}> with a real contraction length
    (N >= 256), bf16 operands and fp16 operands (the latter at the parity bar)."""
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(70 + int(2 * beta))
    N, C, R = 520, 1300, 128
    V = (torch.rand(N, C, generator=g) + (2.0 ** -7 if beta == 0 else 0)).bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    gam = O.gamma_of(beta)
    Wr = O.nmf_w_step(V, W0, H0, beta, gam)
    Hr = O.nmf_h_step(V, Wr, H0, beta, gam)
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, prec, 1)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('rank128_beta_sweep', beta=beta, prec=prec, relW=ew, relH=eh)
    assert ew < tol and eh < tol, (ew, eh)
    assert l0 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(H0, W0), V, beta)), rel=20 * tol)


@pytest.mark.parametrize('prec,tol', [('bf16', 5e-3), ('f16', TOL), ('bf16x3', TOL)])
def test_cfg1_full_size_one_iteration(dev, prec, tol):
    """One MU iteration at BASELINE configs[1]'s full size (4096 x 65536, rank 128, beta = 1) against the reference's
    op sequence (oracle.aten_port, ~1.5 s of CPU per iteration), every precision mode."""
    from oracle import aten_port
    g = torch.Generator().manual_seed(1)
    N, C, R = 4096, 65536, 128
    V = torch.rand(N, C, generator=g).bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    Wr, Hr = aten_port.mu_iterations(V, W0, H0, 1, 1)
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, 1, prec, 1)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('cfg1_full_size_one_iteration', prec=prec, relW=ew, relH=eh)
    assert ew < tol and eh < tol, (ew, eh)


@pytest.mark.parametrize('prec', ['f16', 'f16x'])
def test_cfg2_full_size_beta2_without_reconstruction(dev, prec):
    """BASELINE configs[2], beta = 2, through NMF.fit() -- which takes the path without reconstruction (X @ panel + Gram
    matrix) -- three iterations at full size against the reference's op sequence; 'f16x' on a target fp16 does not hold."""
    from oracle import aten_port
    from torchnmf_amd import engine
    from torchnmf_amd.nmf import NMF
    g = torch.Generator().manual_seed(2)
    N, C, R = 4096, 65536, 128
    V = torch.rand(N, C, generator=g)
    if prec == 'f16':
        V = V.bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    key = ('cfg2_gram', prec)
    if key not in _ORACLE_CACHE:
        torch.set_num_threads(min(16, torch.get_num_threads()))
        _ORACLE_CACHE[key] = aten_port.mu_iterations(V, W0, H0, 2, 3)
    Wr, Hr = _ORACLE_CACHE[key]
    seen = []
    orig = engine.DenseMU.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        seen.append((self.precision_name, self.gram_path))
    engine.DenseMU.__init__ = init
    try:
        m = NMF(W=W0, H=H0).to(dev)
        n = m.fit(V.to(dev), 2, NO_STOP, 3)
    finally:
        engine.DenseMU.__init__ = orig
    assert n == 3 and seen == [(prec, True)], seen
    ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
    record('cfg2_full_size_beta2_gram', prec=prec, relW=ew, relH=eh)
    assert ew < TOL and eh < TOL, (ew, eh)


@pytest.mark.parametrize('beta', [2, 0.5, 0])
def test_cfg2_full_size_one_iteration_f16(dev, beta):
    """BASELINE configs[2] (the beta sweep at the configs[1] shape) in the parity-grade single-plane mode: one full-size
    iteration against the reference's op sequence."""
    from oracle import aten_port
    g = torch.Generator().manual_seed(2)
    N, C, R = 4096, 65536, 128
    V = torch.rand(N, C, generator=g).bfloat16().float()
    if beta <= 0:
        V.clamp_(min=2.0 ** -7)                          # strictly positive for beta <= 0 (nmf.py:332-336)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    Wr, Hr = aten_port.mu_iterations(V, W0, H0, beta, 1)
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, 'f16', 1)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('cfg2_full_size_one_iteration_f16', beta=beta, relW=ew, relH=eh)
    assert ew < TOL and eh < TOL, (ew, eh)


def test_cfg1_full_size_20_iterations_f16_unrounded_target(dev):
    """20 iterations at configs[1]'s full size with a target that fp16 does NOT hold exactly
    (plain U[0,1) floats), against the reference's op sequence -- in the f16 mode (the rounding of V is its dominant
    error on such data and grows with the iteration count, DESIGN.md section 4, which is why 'auto' demands an fp16-exact
    target for it; at 20 iterations it is still inside the bar) and in 'f16x', which keeps V in fp32 and is what 'auto'
    takes here."""
    from oracle import aten_port
    from torchnmf_amd.engine import DenseMU
    g = torch.Generator().manual_seed(3)
    N, C, R = 4096, 65536, 128
    V = torch.rand(N, C, generator=g)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    Wr, Hr = aten_port.mu_iterations(V, W0, H0, 1, 20)
    Vd = V.to(dev)
    # not fp16-exact: 'auto' keeps the target in fp32 (round 4: at fp16 operands -- 'f16x'; before, split bf16)
    assert DenseMU(Vd[:4096, :4096].contiguous(), W0[:4096].clone().to(dev), H0.clone().to(dev), 1.0, precision='auto',
                   allow_f16=True).precision_name == 'f16r'
    for prec in ('f16', 'f16x', 'f16r'):
        W, H = W0.clone().to(dev), H0.clone().to(dev)
        eng = DenseMU(Vd, W, H, 1.0, precision=prec)
        for _ in range(20):
            eng.w_step()
            eng.h_step()
        torch.cuda.synchronize()
        ew, eh = rel_err(W.cpu(), Wr), rel_err(H.cpu(), Hr)
        record('cfg1_full_size_20_iterations_unrounded', prec=prec, relW=ew, relH=eh)
        assert ew < TOL and eh < TOL, (prec, ew, eh)
        del eng


@pytest.mark.parametrize('prec,tol', [('bf16', 5e-3), ('f16', TOL), ('f16x', TOL)])
def test_rank256_w_half_step_with_the_apply_in_its_epilogue(dev, prec, tol):
    """Padded rank 256, beta = 1, a W tall enough (>= 512 row blocks) that its contraction is not split: the W half-step
    applies nmf.py:78-92 in the kernel's epilogue and re-emits W's images through a 128 KiB LDS staging tile -- the launch
    configs[4]'s shard runs.  (Round 4: the single-image instances allocated only their 64 KiB ring; the epilogue ran off
    the end of it, no test came here, and the bench looked 20 % faster on the garbage.)  Two iterations, so that the images
    the epilogue wrote are what the next half-steps read."""
    from oracle import mu_oracle as O
    from torchnmf_amd.engine import DenseMU
    g = torch.Generator().manual_seed(256)
    N, C, R = 300, 65600, 200
    V = torch.rand(N, C, generator=g)
    if prec != 'f16x':
        V = V.bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = DenseMU(V.to(dev), W, H, 1.0, precision=prec)
    assert eng.r_pad == 256 and eng.step_w.nsplit == 1 and eng.step_h.nsplit > 1
    Wr, Hr = W0, H0
    for _ in range(2):
        eng.w_step()
        eng.h_step()
        Wr = O.nmf_w_step(V, Wr, Hr, 1, 1.0)
        Hr = O.nmf_h_step(V, Wr, Hr, 1, 1.0)
    torch.cuda.synchronize()
    ew, eh = rel_err(W.cpu(), Wr), rel_err(H.cpu(), Hr)
    record('rank256_fused_apply', prec=prec, relW=ew, relH=eh)
    assert ew < tol and eh < tol, (ew, eh)
    assert eng.divergence() == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, 1)), rel=20 * tol)


@pytest.mark.parametrize('prec,tol', [('bf16', 5e-3), ('f16', TOL)])
def test_cfg5_shard_slice_rank256(dev, prec, tol):
    """The rank-256 kernel of BASELINE configs[4]'s per-GPU shard on an 8192 x 16384 slice (the full 262144-column shard
    differs only in the number of row blocks), one iteration against the oracle; fp16 operands at the parity bar."""
    from oracle import mu_oracle as O
    g = torch.Generator().manual_seed(5)
    N, C, R = 8192, 16384, 256
    V = torch.rand(N, C, generator=g).bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, 1, prec, 1)
    Wr = O.nmf_w_step(V, W0, H0, 1, 1.0)
    Hr = O.nmf_h_step(V, Wr, H0, 1, 1.0)
    ew, eh = rel_err(W1, Wr), rel_err(H1, Hr)
    record('cfg5_shard_slice_rank256', prec=prec, relW=ew, relH=eh)
    assert ew < tol and eh < tol, (ew, eh)


@pytest.mark.parametrize('N,C,nsplit,regs', [(300, 200, None, (0.0, 0.0)),      # 4 tiles: the last group alone; ragged rows
                                              (130, 500, None, (0.0, 0.0)),      # 8 tiles: one pass of the loop + the last group
                                              (600, 1000, 1, (0.0, 0.0)),        # 16 tiles unsplit: H half-step with the fused apply
                                              (600, 1000, 2, (0.1, 0.5)),        # 8 + 8 tiles, regularised (general epilogue)
                                              (400, 1300, 3, (0.0, 0.0)),        # 24 tiles in splits of 8
                                              (260, 2100, 5, (0.0, 0.0)),        # 36 tiles in splits of 8: the last split holds 4
                                              (1100, 300, 8, (0.3, 0.0))])       # more splits than groups: empty workgroups
def test_rank256_software_pipelined_kernel(dev, monkeypatch, N, C, nsplit, regs):
    """nmfmu::sp_kernel (round 6: padded rank 256, beta = 1, fp16 -- the kernel of configs[4]'s shard) over its control
    flow: one / several four-tile groups, the last group's shorter final iteration, contraction splits that leave short and
    empty workgroups (tiles_per_split is rounded to a multiple of four), ragged rows and ranks, both epilogues (fused apply,
    plain and regularised; slab stores).  Two iterations against the oracle, so that the images the fused apply wrote are
    what the next half-steps read.  The tolerance is what fp16 operands give on contractions this short (64 .. 2100 terms;
    DESIGN.md section 4) -- the parity bar itself is held at configs[4]'s own lengths by test_cfg5_shard_slice_rank256."""
    from oracle import mu_oracle as O
    from torchnmf_amd import _capi
    from torchnmf_amd.engine import DenseMU
    R = 200 if N != 600 else 256
    g = torch.Generator().manual_seed(N + C)
    V = torch.rand(N, C, generator=g).bfloat16().float()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    if nsplit is not None:
        monkeypatch.setenv('TORCHNMF_AMD_NSPLIT', str(nsplit))
    alpha, l1r = regs
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = DenseMU(V.to(dev), W, H, 1.0, alpha * l1r, alpha * (1 - l1r), precision='f16')
    assert eng.r_pad == 256 and eng.be.kernel_family(256, _capi.PREC_F16, 1.0) == _capi.KERNEL_SP
    Wr, Hr = W0, H0
    for _ in range(2):
        eng.w_step()
        eng.h_step()
        Wr = O.nmf_w_step(V, Wr, Hr, 1, 1.0, alpha * l1r, alpha * (1 - l1r))
        Hr = O.nmf_h_step(V, Wr, Hr, 1, 1.0, alpha * l1r, alpha * (1 - l1r))
    torch.cuda.synchronize()
    ew, eh = rel_err(W.cpu(), Wr), rel_err(H.cpu(), Hr)
    record('rank256_sp_kernel', N=N, C=C, nsplit=(eng.step_w.nsplit, eng.step_h.nsplit), regs=regs, relW=ew, relH=eh)
    assert ew < 4e-4 and eh < 4e-4, (ew, eh)
    assert eng.divergence() == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, 1)), rel=1e-3)


def test_cfg5_full_shard_two_iterations(dev):
    """ONE full per-GPU shard of BASELINE configs[4] -- 8192 x 262144, rank 256, beta = 1, fp16 operands (4 GiB of packed V per
    orientation; 2 048 workgroups of 128 tiles in the W half-step, 256 x 1 024 tiles in the H half-step) -- through the
    engine: the loss decreases, every factor entry is finite and positive, and 64 sampled rows of W and of H after the second
    iteration agree with the oracle's half-steps recomputed for exactly those rows from the factors the GPU held before them
    (the W half-step of a row needs its column of V and all of H; the H half-step of a row its row of V and all of W).
    A size-independent check of the launch the slice test (one sixteenth of this shard) cannot see: every row block of the
    2 048, all eight rounds of workgroups, the fused apply over 256 MiB of master rows."""
    from oracle import mu_oracle as O
    from torchnmf_amd.engine import DenseMU
    N, C, R = 8192, 262144, 256
    g = torch.Generator(device=dev).manual_seed(54)
    V = torch.rand(N, C, device=dev, generator=g).bfloat16().float()
    W = torch.randn(C, R, device=dev, generator=g).abs_()
    H = torch.randn(N, R, device=dev, generator=g).abs_()
    eng = DenseMU(V, W, H, 1.0, precision='f16')
    assert eng.r_pad == 256 and eng.step_w.nsplit == 1 and eng.step_h.owner.rows_pad == N
    loss0 = eng.divergence()
    eng.w_step()
    eng.h_step()
    loss1 = eng.divergence()
    Wp, Hp = W.clone(), H.clone()
    gs = torch.Generator().manual_seed(7)
    cols = torch.randperm(C, generator=gs)[:64].sort().values
    rows = torch.randperm(N, generator=gs)[:64].sort().values
    eng.w_step()
    torch.cuda.synchronize()
    Hp_c = Hp.cpu()
    Wr = O.nmf_w_step(V[:, cols.to(dev)].cpu(), Wp[cols.to(dev)].cpu(), Hp_c, 1, 1.0)
    ew = rel_err(W[cols.to(dev)].cpu(), Wr)
    W2 = W.cpu()
    eng.h_step()
    torch.cuda.synchronize()
    Hr = O.nmf_h_step(V[rows.to(dev)].cpu(), W2, Hp_c[rows], 1, 1.0)
    eh = rel_err(H[rows.to(dev)].cpu(), Hr)
    loss2 = eng.divergence()
    wmin, hmin = float(W.min()), float(H.min())
    record('cfg5_full_shard', relW_rows=ew, relH_rows=eh, loss=(loss0, loss1, loss2), w_min=wmin, h_min=hmin)
    assert loss0 > loss1 > loss2 > 0
    assert bool(torch.isfinite(W).all()) and bool(torch.isfinite(H).all())
    assert wmin >= 0 and hmin >= 0        # (an entry may underflow to exactly 0, as it does in the reference; never below)
    assert ew < TOL and eh < TOL, (ew, eh)
    # the column sums the next half-step's closed-form denominators read (nmf.py:122-131) cover every row block
    assert rel_err(eng.fW.colsum[:R].cpu().double(), W.double().sum(0).cpu()) < 1e-5


@pytest.mark.parametrize('beta', [0.0, 0.5, 1.5, 0.3, 3.0, -1.0])
@pytest.mark.parametrize('N,C,nsplit,regs', [(300, 200, None, (0.0, 0.0)),      # 4 tiles: the last group alone; ragged rows
                                              (130, 500, None, (0.0, 0.0)),      # 8 tiles: one pass of the loop + the last group
                                              (600, 1000, 1, (0.1, 0.5)),        # 16 tiles unsplit: fused apply, regularised
                                              (260, 2100, 5, (0.0, 0.0)),        # 36 tiles in splits of 8: the last split holds 4
                                              (1100, 300, 8, (0.0, 0.0))])       # more splits than groups: empty workgroups
def test_rank128_two_accumulator_software_pipelined_kernel(dev, monkeypatch, beta, N, C, nsplit, regs):
    """nmfmu::sp2_kernel (round 6: padded rank 128, beta not in {1, 2}, fp16 -- the kernel of configs[2]'s beta < 1 legs) over
    its control flow -- one / several four-tile groups, the last group's shorter final iteration, contraction splits with
    short and empty workgroups, both epilogues -- for every elementwise branch: beta = 0 (reciprocal), 0.5 and 1.5 (rsqrt),
    and the generic log / exp branch on both sides of the scaled range (0.3, 3, -1).  Two iterations against the oracle.  The
    tolerance is what fp16 operands give on contractions this short; the parity bar itself is held at
    configs[2]'s own lengths by test_cfg2_full_size_*."""
    from oracle import mu_oracle as O
    from torchnmf_amd import _capi
    from torchnmf_amd.engine import DenseMU
    R = 100 if N != 600 else 128
    g = torch.Generator().manual_seed(N + C)
    V = torch.rand(N, C, generator=g).bfloat16().float() + 2.0 ** -7
    W0 = torch.randn(C, R, generator=g).abs() + 0.05
    H0 = torch.randn(N, R, generator=g).abs() + 0.05
    if nsplit is not None:
        monkeypatch.setenv('TORCHNMF_AMD_NSPLIT', str(nsplit))
    alpha, l1r = regs
    gam = O.gamma_of(beta)
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = DenseMU(V.to(dev), W, H, beta, alpha * l1r, alpha * (1 - l1r), precision='f16')
    assert eng.r_pad == 128 and eng.be.kernel_family(128, _capi.PREC_F16, float(beta)) == _capi.KERNEL_SP
    Wr, Hr = W0, H0
    for _ in range(2):
        eng.w_step()
        eng.h_step()
        Wr = O.nmf_w_step(V, Wr, Hr, beta, gam, alpha * l1r, alpha * (1 - l1r))
        Hr = O.nmf_h_step(V, Wr, Hr, beta, gam, alpha * l1r, alpha * (1 - l1r))
    torch.cuda.synchronize()
    ew, eh = rel_err(W.cpu(), Wr), rel_err(H.cpu(), Hr)
    record('rank128_sp2_kernel', beta=beta, N=N, C=C, nsplit=(eng.step_w.nsplit, eng.step_h.nsplit), regs=regs, relW=ew, relH=eh)
    assert ew < 6e-4 and eh < 6e-4, (ew, eh)


@pytest.mark.parametrize('N,C,R,beta', [(4096, 65536, 128, 0.5), (4096, 65536, 128, 0.0), (8192, 65536, 256, 1.0)])
def test_software_pipelined_kernels_are_deterministic_under_load(dev, N, C, R, beta):
    """The hand-placed kernels of round 6 order their memory traffic by counted waits and ONE barrier per tile; a wrong count
    or a misplaced barrier shows only under load, as run-to-run differences (the first sp2_kernel build left its LDS-DMA
    pieces "in flight behind" older register loads -- they complete first -- and prefetched operands ahead of the barrier
    that publishes them: every small test passed, configs[2]'s in-run parity at k = 10 did not).  Full-chip problems, two
    iterations, three runs from identical inputs: bitwise equal factors."""
    from torchnmf_amd.engine import DenseMU
    g = torch.Generator(device=dev).manual_seed(N + R)
    V = torch.rand(N, C, device=dev, generator=g).bfloat16().float() + (2.0 ** -7 if beta <= 0 else 0.0)
    W0 = torch.randn(C, R, device=dev, generator=g).abs_()
    H0 = torch.randn(N, R, device=dev, generator=g).abs_()
    outs = []
    for _ in range(3):
        W, H = W0.clone(), H0.clone()
        eng = DenseMU(V, W, H, beta, precision='f16')
        for _ in range(2):
            eng.w_step()
            eng.h_step()
        torch.cuda.synchronize()
        outs.append((W.clone(), H.clone()))
        del eng
    for W, H in outs[1:]:
        assert torch.equal(W, outs[0][0]) and torch.equal(H, outs[0][1])


@pytest.mark.parametrize('beta', [1, 0, 0.5, 1.5, 3])
@pytest.mark.parametrize('shape', [(600, 2000, 100), (300, 700, 200), (200, 330, 24), (520, 1100, 48)])
def test_half_steps_f16r_every_beta(dev, beta, shape):
    """precision='f16r' (round 6): fp16 operands, the target at THREE bytes per element -- the fp32 rounded to its top 24 bits,
    16 significant bits.  One iteration on plain fp32 floats (which fp16 would round) against the oracle ON THE UNROUNDED
    TARGET; as close as 'f16x' (whose target is exact) within the operands' own rounding.  beta == 1 at rank <= 128 runs the
    ping-pong kernel's 3-byte instance, everything else the four-wave kernel's."""
    from oracle import mu_oracle as O
    N, C, R = shape
    g = torch.Generator().manual_seed(N + C + R)
    V = torch.rand(N, C, generator=g) + (2.0 ** -7 if beta <= 0 else 0.0)
    W0 = torch.randn(C, R, generator=g).abs() + 0.05
    H0 = torch.randn(N, R, generator=g).abs() + 0.05
    gam = O.gamma_of(beta)
    Wr = O.nmf_w_step(V, W0, H0, beta, gam)
    Hr = O.nmf_h_step(V, Wr, H0, beta, gam)
    res = {}
    for prec in ('f16r', 'f16x'):
        W1, H1, l0, l1 = _one_iter(dev, V, W0, H0, beta, prec, 1)
        res[prec] = (rel_err(W1, Wr), rel_err(H1, Hr))
        assert l1 == pytest.approx(float(O.beta_div(O.nmf_reconstruct(Hr, Wr), V, beta)), rel=5e-3)
    record('f16r_half_steps', beta=beta, shape=shape, f16r=res['f16r'], f16x=res['f16x'])
    assert max(res['f16r']) < 6e-4                                   # short contractions: what fp16 operands give here
    assert max(res['f16r']) < 1.25 * max(res['f16x']) + 2e-6         # the 3-byte target costs nothing the operands do not


def test_f16r_target_packing_is_the_top_24_bits(dev):
    """nmfmu_pack_x for NMFMU_PREC_F16R: decode the packed bytes on the host (bits 31..16 = the 16-bit word of the f16 fragment
    order, bits 15..8 = the element's byte in the two chunks behind the four word chunks of every lane, csrc/nmfmu_layout.h)
    and compare with the fp32 rounded to nearest-even at bit 8: bit-exact, relative error <= 2^-16 everywhere in fp32's range
    (subnormals: absolute 2^-142), zero / one / huge / tiny values, a value that would round up to infinity truncated."""
    from torchnmf_amd import _capi
    lib = _capi.load()
    N, C = 256, 256
    g = torch.Generator().manual_seed(19)
    V = torch.rand(N, C, generator=g) * torch.tensor(10.0) ** torch.randint(-30, 30, (N, C), generator=g).float()
    V[0, :8] = torch.tensor([0.0, 1.0, 65504.0, 7e4, 6e-8, 1e-40, 0.333333343, 2049.0])
    V[0, 8] = torch.tensor(0x7f7fffff, dtype=torch.int32).view(torch.float32)       # FLT_MAX: truncated, not rounded up to inf
    Vd = V.to(dev)
    xp = torch.zeros(N * C * 3, dtype=torch.uint8, device=dev)
    flags = torch.tensor([0, 0x7f800000], dtype=torch.int32, device=dev)
    _capi.check(lib.nmfmu_pack_x(Vd.data_ptr(), C, N, C, 0, _capi.PREC_F16R, 128, xp.data_ptr(), N, C, flags.data_ptr(), _stream()), 'pack_x')
    torch.cuda.synchronize()
    raw = xp.cpu().numpy().astype(np.uint32)
    m, k = np.meshgrid(np.arange(N), np.arange(C), indexing='ij')
    mb, ml = m // 128, m % 128
    w, j = ml // 32, ml % 32
    kt, kl = k // 64, k % 64
    hl, i = kl // 32, kl % 32                        # element i of this lane's 32 columns
    lane = hl * 32 + j
    base = ((mb * (C // 64) + kt) * 4 + w) * 6       # first of this wave-tile's six 1-KiB chunk rows
    hoff = ((base + i // 8) * 64 + lane) * 16 + 2 * (i % 8)
    roff = ((base + 4 + i // 16) * 64 + lane) * 16 + (i % 16)
    got = (raw[hoff + 1] << 24) | (raw[hoff] << 16) | (raw[roff] << 8)
    bits = V.numpy().view(np.uint32).astype(np.uint64)
    want = (bits + 0x7f + ((bits >> 8) & 1)) & 0xffffff00
    want[0, 8] = 0x7f7fff00
    assert np.array_equal(got.astype(np.uint64), want)
    dec = got.astype(np.uint32).view(np.float32).astype(np.float64)
    ref = V.numpy().astype(np.float64)
    assert np.all(np.abs(dec - ref) <= np.maximum(ref * 2.0 ** -16, 2.0 ** -142))
    assert dec[0, 0] == 0.0 and dec[0, 1] == 1.0 and dec[0, 2] == 65504.0 and int(flags[0]) == 0


def test_auto_precision_policy(dev, monkeypatch):
    """'auto' = the fastest mode that meets the 1e-4 bar, never plain bf16: fp16 operands where both dimensions are
    >= 4096, the target is exactly representable in fp16 and the data sit inside fp16's range; split bf16 otherwise
    (small problems, targets fp16 would round, huge values, normalised tiny values); above rank 128 the GEMM engine
    when unsharded and an error when sharded."""
    from torchnmf_amd.engine import DenseMU
    g = torch.Generator().manual_seed(9)

    def pick(N, C, R, scale=1.0, beta=1.0, allow=True, exact=True):
        V = torch.rand(N, C, generator=g)
        V = ((V.half().float() if exact else V) * scale).to(dev)
        W = torch.randn(C, R, generator=g).abs().to(dev)
        H = torch.randn(N, R, generator=g).abs().to(dev)
        return DenseMU(V, W, H, beta, precision='auto', allow_f16=allow).precision_name
    assert pick(4096, 4352, 64) == 'f16'
    assert pick(4096, 4352, 64, beta=2.0) == 'f16' and pick(4096, 4352, 64, beta=0.5) == 'f16'
    assert pick(4096, 4352, 200) == 'f16'                      # padded rank 256: the four-wave fp16 kernel
    assert pick(4096, 4352, 64, exact=False) == 'f16r'         # fp16 would round the target: 3-byte form (round 6; 'f16x' in round 4)
    assert pick(4096, 4352, 200, exact=False, beta=0.5) == 'f16r'
    assert pick(4096, 4352, 200, exact=False, beta=2.0) == 'f16x'   # beta == 2: the target is an MFMA operand, it stays fp32
    monkeypatch.setenv('TORCHNMF_AMD_AUTO_F16X', '0')
    assert pick(4096, 4352, 64, exact=False) == 'bf16x3'
    monkeypatch.delenv('TORCHNMF_AMD_AUTO_F16X')
    assert pick(4096, 4352, 64, allow=False) == 'bf16x3'       # trainer / PLCA engines keep the fp32-grade default
    assert pick(2048, 8192, 64) == 'bf16x3'                    # short contraction: rounding errors do not average down
    assert pick(4096, 4352, 64, scale=2.0 ** 20) == 'bf16x3'   # outside fp16's range
    assert pick(4096, 4352, 64, scale=2.0 ** -20) == 'bf16x3'  # mostly fp16-subnormal targets
    with pytest.raises(NotImplementedError):                   # rank 129..256 and no parity-grade fused mode: no silent bf16
        pick(512, 600, 200)
    # through fit(): rank 200, small -> the GEMM engine (fp32-grade); large fp16-exact -> fused fp16 kernel
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd import nmfd_engine, engine
    seen = []

    def spy_on(cls):
        orig = cls.__init__

        def init(self, *a, **k):
            orig(self, *a, **k)
            seen.append((cls.__name__, self.precision_name))
        monkeypatch.setattr(cls, '__init__', init)
    spy_on(engine.DenseMU)
    spy_on(nmfd_engine.WideRankMU)
    V = torch.rand(300, 400, generator=g)
    NMF(V.shape, 200).to(dev).fit(V.to(dev), max_iter=2)
    V = torch.rand(4096, 4096, generator=g).half().float()
    NMF(V.shape, 200).to(dev).fit(V.to(dev), max_iter=2)
    assert seen[0][0] == 'WideRankMU' and seen[1] == ('DenseMU', 'f16'), seen


def test_betamu_converted_target_is_repacked(dev):
    """a float64 (converted) target edited in place between steps must not hit a stale packed copy."""
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.trainer import BetaMu
    g = torch.Generator().manual_seed(31)
    N, C, R = 96, 130, 8
    V64 = torch.rand(N, C, generator=g, dtype=torch.float64).to(dev)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    m = NMF(W=W0, H=H0).to(dev)
    tr = BetaMu(m.parameters(), beta=1, precision='bf16x3')
    tr.step(lambda: (V64, m))
    eng0 = next(iter(tr._engines.values()))[0]
    V64.mul_(3.0)                                   # same storage, same object, new values
    Wb, Hb = m.W.data.cpu().clone(), m.H.data.cpu().clone()
    tr.step(lambda: (V64, m))
    # ... and without rebuilding the engine: same buffers, target packed afresh into them
    assert len(tr._engines) == 1 and next(iter(tr._engines.values()))[0] is eng0
    # the update must have used the NEW target: compare with a fresh optimizer on a fresh fp32 copy
    m2 = NMF(W=Wb, H=Hb).to(dev)
    tr2 = BetaMu(m2.parameters(), beta=1, precision='bf16x3')
    Vf = V64.float()
    tr2.step(lambda: (Vf, m2))
    assert rel_err(m.W.data.cpu(), m2.W.data.cpu()) < 1e-6 and rel_err(m.H.data.cpu(), m2.H.data.cpu()) < 1e-6


def test_reconstruct_leading_batch_dims_and_star_import(dev):
    """NMF.reconstruct accepts leading batch dimensions like F.linear; `from ...nmf import *` exports the
    same names as the reference (torchnmf/nmf.py:16-18)."""
    import torchnmf_amd.nmf as mod
    assert set(mod.__all__) == {'BaseComponent', 'NMF', 'NMFD', 'NMF2D', 'NMF3D'}
    g = torch.Generator().manual_seed(2)
    H = torch.rand(3, 5, 7, generator=g)
    W = torch.rand(11, 7, generator=g)
    out = mod.NMF.reconstruct(H.to(dev), W.to(dev)).cpu()
    assert out.shape == (3, 5, 11)
    assert rel_err(out, H @ W.t()) < 1e-6


def test_nmfd_cfg4_full_size(dev):
    """BASELINE configs[3] at full size: NMFD 1 x 1025 x 8192, rank 8, T = 400, beta = 1 -- two iterations against the
    reference's op sequence (conv1d + two backward passes on the host, ~2 s per iteration)."""
    from torchnmf_amd.nmf import NMFD
    from oracle import aten_port
    g = torch.Generator().manual_seed(4)
    Cc, L, R, T = 1025, 8192, 8, 400
    V = torch.rand(1, Cc, L, generator=g)
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(1, R, L - T + 1, generator=g).abs()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    Wr, Hr = aten_port.mu_iterations_nmfd(V, W0, H0, 1, 2)
    for prec, tol in (('bf16x3', 1e-4), ('bf16', 2e-2), ('f16', 1e-4), ('auto', 1e-4)):
        m = NMFD(W=W0, H=H0).to(dev)
        n = m.fit(V.to(dev), 1, NO_STOP, 2, precision=prec)
        assert n == 2
        ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
        print(f'configs[3] full size, {prec}: relW={ew:.2e} relH={eh:.2e}')
        assert ew < tol and eh < tol, (prec, ew, eh)


# ----------------------------------------------------------------------------------------------------------
# Round 5: window staging of the implicit Toeplitz operand (csrc/nmfmu_gemm.h, WS) -- the NMFD GEMMs move 19 instead of
# 32 LDS-DMA pieces per k-tile.  The staged launches feed the MFMAs the same operands in the same order, so the whole
# iteration must agree with the chunk-major path BIT FOR BIT; the oracle comparison rides on the existing NMFD tests
# (configs[3] at full size takes the staged path: test_nmfd_cfg4_full_size).
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,Cc,L,R,T,staged', [
    (2, 129, 384, 4, 160, dict(recon_w=1, num_w=1, recon_h=1)),     # rows (b,l) and rows (r,t) staged; T8 = 20: r boundaries mid k-tile
    (1, 140, 256, 8, 64, dict(recon_w=1, num_w=0, recon_h=1)),      # T = 64: boundaries on k-tile edges; rows (r,t) need T >= 128
    (1, 70, 256, 8, 136, dict(recon_w=1, num_w=0, recon_h=1)),      # T8 = 17 (odd): a boundary between the two halves of a k-step
    (3, 33, 128, 16, 72, dict(recon_w=1, num_w=0, recon_h=1)),      # T8 = 9, three batch entries of one tile each
    (1, 64, 320, 2, 192, dict(recon_w=0, num_w=1, recon_h=0)),      # L % 128 != 0: only the rows-(r,t) operand is staged; tiles spanning two r
    (1, 1025, 1024, 2, 128, dict(recon_w=1, num_w=1, recon_h=1)),   # ragged 1025th channel inside the staged reconstruction GEMMs
])
@pytest.mark.parametrize('prec,beta', [('bf16x3', 1.0), ('f16', 1.0), ('bf16', 1.0), ('bf16x3', 0.5), ('bf16', 2.0)])
def test_nmfd_window_staging_is_bit_identical(dev, B, Cc, L, R, T, staged, prec, beta):
    """nmfmu_gemm_desc.stage_mode 0 (window staging where the shape admits it) against 1 (the chunk-major tiles of rounds 1-4):
    the MFMAs see the same operands in the same order, so factors and loss agree BIT FOR BIT; `staged` says which launches
    of the iteration must have taken the window (nmfmu_gemm_window_staged)."""
    from torchnmf_amd.nmfd_engine import ConvMU
    g = torch.Generator().manual_seed(B * 1000 + T)
    V = (torch.rand(B, Cc, L, generator=g) + 1e-3).to(dev)
    W0 = torch.randn(Cc, R, T, generator=g).abs()
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
    res = {}
    for mode in ('1', '0'):
        os.environ['TORCHNMF_AMD_NMFD_WINSTAGE'] = mode
        try:
            W, H = W0.clone().to(dev), H0.clone().to(dev)
            try:
                eng = ConvMU(V, W, H, beta, precision=prec)
            except ValueError:
                pytest.skip('fp16 planes are not built for this shape')
            for _ in range(2):
                eng.w_step()
                eng.h_step()
            loss = eng.divergence()
            torch.cuda.synchronize()
            res[mode] = (W.cpu(), H.cpu(), loss, dict(eng.staged))
        finally:
            os.environ.pop('TORCHNMF_AMD_NMFD_WINSTAGE', None)
    got = {k: v for k, v in res['1'][3].items() if k in staged}
    assert got == staged, got
    assert all(v == 0 for v in res['0'][3].values())
    assert torch.isfinite(res['1'][0]).all() and torch.isfinite(res['1'][1]).all()
    assert torch.equal(res['1'][0], res['0'][0]) and torch.equal(res['1'][1], res['0'][1]) and res['1'][2] == res['0'][2]


def test_nmfd_window_staging_against_oracle(dev):
    """One staged shape end to end against the oracle (split bf16, three iterations): both operand forms, r boundaries
    inside k-tiles and inside tile rows."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMFD
    g = torch.Generator().manual_seed(55)
    B, Cc, L, R, T = 2, 129, 384, 4, 160
    V = torch.rand(B, Cc, L, generator=g) + 1e-3
    W0, H0 = torch.randn(Cc, R, T, generator=g).abs(), torch.randn(B, R, L - T + 1, generator=g).abs()
    m = NMFD(W=W0, H=H0).to(dev)
    assert m.fit(V.to(dev), 1, NO_STOP, 3, precision='bf16x3') == 3
    Wr, Hr = W0, H0
    for _ in range(3):
        Wr = O.nmfd_w_step(V, Wr, Hr, 1, 1.0)
        Hr = O.nmfd_h_step(V, Wr, Hr, 1, 1.0)
    assert rel_err(m.W.data.cpu(), Wr) < TOL and rel_err(m.H.data.cpu(), Hr) < TOL


# ----------------------------------------------------------------------------------------------------------
# Round 5: launch diet of the window-operand path (NMF2D / NMF3D / NMFD below 128 taps, beta == 1): the three rank-sum
# launches and the Wk packing ride in the two apply kernels (nmfmu_conv_apply_pack_w_wk, nmfmu_conv_apply_h_rows_sums).
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cls,shape,rank,ks', [
    ('NMF2D', (1, 20, 40, 72), 8, (8, 8)),        # 64 taps: every 64-wide k tile is one rank
    ('NMF2D', (1, 70, 24, 80), 5, (4, 32)),       # 128 taps, rank 5: 51 positions per block step, idle lanes; two channel tiles
    ('NMFD', (2, 33, 300), 16, (96,)),            # one shift axis below 128 taps (the short-kernel NMFD path), two batch entries
    ('NMF3D', (2, 6, 9, 10, 24), 3, (2, 4, 8)),   # three shift axes, 64 taps
])
@pytest.mark.parametrize('prec', ['bf16x3', 'f16'])
def test_rows_path_launch_diet(dev, cls, shape, rank, ks, prec):
    from oracle import mu_oracle as O
    from torchnmf_amd.nmfd_engine import ConvMU
    g = torch.Generator().manual_seed(sum(shape) + rank)
    V = torch.rand(*shape, generator=g) + 1e-3
    W0 = torch.randn(shape[1], rank, *ks, generator=g).abs()
    H0 = torch.randn(shape[0], rank, *[l - k + 1 for l, k in zip(shape[2:], ks)], generator=g).abs()
    res = {}
    for mode in ('1', '0'):
        os.environ['TORCHNMF_AMD_NMFD_ROWS_FUSED'] = mode
        try:
            W, H = W0.clone().to(dev), H0.clone().to(dev)
            try:
                eng = ConvMU(V.to(dev), W, H, 1.0, precision=prec)
            except ValueError:
                pytest.skip('fp16 planes are not built for this shape')
            assert eng.h_rows and eng.rows_fused == (mode == '1')
            for _ in range(3):
                eng.w_step()
                eng.h_step()
            eng.refresh_images()                 # (an outside change of the factors: both forms must take it)
            eng.w_step()
            eng.h_step()
            loss = eng.divergence()
            torch.cuda.synchronize()
            res[mode] = (W.cpu(), H.cpu(), loss)
        finally:
            os.environ.pop('TORCHNMF_AMD_NMFD_ROWS_FUSED', None)
    # the rank sums are added in another order (tile / block partials instead of 128 chunks): fp32 rounding of the
    # denominators; a single-plane mode then flips the odd operand rounding
    tol = 1e-5 if prec == 'bf16x3' else 3e-4
    assert rel_err(res['1'][0], res['0'][0]) < tol and rel_err(res['1'][1], res['0'][1]) < tol
    assert abs(res['1'][2] - res['0'][2]) <= 1e-4 * abs(res['0'][2])
    if prec == 'bf16x3':
        Wr, Hr = W0, H0
        w_step, h_step = (O.nmfd_w_step, O.nmfd_h_step) if len(ks) == 1 else (O.convnd_w_step, O.convnd_h_step)
        for _ in range(4):
            Wr = w_step(V, Wr, Hr, 1, 1.0)
            Hr = h_step(V, Wr, Hr, 1, 1.0)
        assert rel_err(res['1'][0], Wr) < TOL and rel_err(res['1'][1], Hr) < TOL
