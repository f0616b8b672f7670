"""Seeded random sweep of dense NMF shapes / ranks / betas through the HIP path against the CPU oracle.

Complements the hand-picked cases of test_gpu_parity.py: ragged sizes hit every padding combination, ranks cover all
four rank pads and both tile heights, and the engine's own contraction-split heuristic decides the tile counts."""
import random

import pytest
import torch

from conftest import record, rel_err

pytestmark = pytest.mark.gpu


def _cases():
    rng = random.Random(20260926)
    out = []
    for i in range(36):
        N = rng.choice([1, 7, 64, 129, 255, 256, 257, 600, 1100])
        C = rng.choice([1, 33, 128, 500, 1025, 2600, 5000])
        R = rng.choice([1, 3, 16, 32, 33, 64, 100, 128, 129, 200, 256])
        beta = rng.choice([1, 1, 1, 2, 0.5, 0, 1.5, 3])
        reg = rng.choice([(0.0, 0.0), (0.1, 0.5)])
        out.append((i, N, C, R, beta, reg))
    return out


@pytest.mark.parametrize('i,N,C,R,beta,reg', _cases())
def test_random_dense_fit_matches_oracle(i, N, C, R, beta, reg):
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMF
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1000 + i)
    V = torch.rand(N, C, generator=g) + (1e-3 if beta <= 0 else 0.0)
    W0, H0 = torch.randn(C, R, generator=g).abs() + 1e-3, torch.randn(N, R, generator=g).abs() + 1e-3
    # bf16x3 above rank 128 runs on the GEMM engine.  The single-plane modes see a target their storage type holds
    # exactly; at these (short) contraction lengths fp16 operands are held to what 11 significant bits give without
    # averaging -- the 1e-4 bar of that mode is for the BASELINE-sized contractions (test_gpu_parity.py)
    for prec, tol in (('bf16x3', 1e-4), ('f16', 1e-3), ('bf16', 3e-2)):
        Vp = V.bfloat16().float() if prec != 'bf16x3' else V
        m = NMF(W=W0, H=H0).to(dev)
        n = m.fit(Vp.to(dev), beta, -1e9, 3, alpha=reg[0], l1_ratio=reg[1], precision=prec)
        Wr, Hr, nr, _, _ = O.fit(Vp, W0, H0, beta, -1e9, 3, reg[0], reg[1])
        ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
        record('fuzz_dense', i=i, shape=(N, C, R), beta=beta, prec=prec, relW=ew, relH=eh)
        assert n == nr == 3 and ew < tol and eh < tol, (prec, N, C, R, beta, ew, eh)
        assert bool(torch.isfinite(m.W.data).all()) and bool(torch.isfinite(m.H.data).all())


def _conv_cases():
    rng = random.Random(7)
    out = []
    for i in range(14):
        B = rng.choice([1, 1, 2, 3])
        C = rng.choice([1, 17, 64, 130, 257])
        T = rng.choice([1, 3, 8, 16, 24, 40])
        L = rng.choice([48, 96, 200, 304, 520]) + rng.choice([0, 0, 3])
        L = max(L, T + 4)
        R = rng.choice([1, 2, 5, 8])
        out.append((i, B, C, L, R, T, rng.choice([1, 1, 2, 0.5])))
    return out


@pytest.mark.parametrize('i,B,C,L,R,T,beta', _conv_cases())
def test_random_nmfd_fit_matches_oracle(i, B, C, L, R, T, beta):
    """Taps / frames that are multiples of 8 run on the implicit window tables, the others on explicit planes."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMFD
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(500 + i)
    V = torch.rand(B, C, L, generator=g) + 1e-3
    W0 = torch.randn(C, R, T, generator=g).abs() + 1e-3
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs() + 1e-3
    m = NMFD(W=W0, H=H0).to(dev)
    n = m.fit(V.to(dev), beta, -1e9, 3, precision='bf16x3')
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, beta, -1e9, 3, kind='nmfd')
    ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
    assert n == nr == 3 and ew < 1e-4 and eh < 1e-4, (B, C, L, R, T, beta, ew, eh)


def _conv_cases_long_taps():
    rng = random.Random(11)
    out = []
    for i in range(7):     # (the CPU oracle dominates the run time of these)
        B = rng.choice([1, 1, 2])
        C = rng.choice([40, 128, 129, 130, 136, 200, 257, 385])      # 128 k + 1..8 take the ragged-channel path
        T = rng.choice([128, 130, 136, 200, 400])                    # >= 128 taps: fold from tile diagonal sums, fused sums
        L = T + rng.choice([40, 200, 333, 600, 2100])               # B L >= 2048 (and even in 64s): split-K numerator
        R = rng.choice([1, 2, 3, 9])
        out.append((i, B, C, L, R, T, rng.choice([1, 1, 1, 2, 0.5]), rng.choice(['bf16x3', 'bf16x3', 'bf16'])))
    return out


@pytest.mark.parametrize('i,B,C,L,R,T,beta,prec', _conv_cases_long_taps())
def test_random_nmfd_long_taps_matches_oracle(i, B, C, L, R, T, beta, prec):
    """The round-2 NMFD paths through the public ``NMFD.fit``: per-tile diagonal sums instead of the Y matrix, ragged
    channels off the GEMM tile grid, rank sums fused into their producers, split-K W numerator -- whichever combination
    the shape selects -- over 12 iterations (one loss evaluation, same stopping decision as the oracle)."""
    from oracle import mu_oracle as O
    from torchnmf_amd.nmf import NMFD
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(900 + i)
    V = torch.rand(B, C, L, generator=g) + 1e-3
    W0 = torch.randn(C, R, T, generator=g).abs() + 1e-3
    H0 = torch.randn(B, R, L - T + 1, generator=g).abs() + 1e-3
    Vp = V.bfloat16().float() if prec == 'bf16' else V
    m = NMFD(W=W0, H=H0).to(dev)
    n = m.fit(Vp.to(dev), beta, -1e9, 12, precision=prec)
    Wr, Hr, nr, losses, _ = O.fit(Vp, W0, H0, beta, -1e9, 12, kind='nmfd')
    tol = 1e-4 if prec == 'bf16x3' else 3e-2
    ew, eh = rel_err(m.W.data.cpu(), Wr), rel_err(m.H.data.cpu(), Hr)
    assert n == nr == 12 and ew < tol and eh < tol, (B, C, L, R, T, beta, prec, ew, eh)
    assert bool(torch.isfinite(m.W.data).all()) and bool(torch.isfinite(m.H.data).all())
