#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (torchnmf 0.3.5).

Runs only in the build container, where the reference is mounted read-only at
/root/reference; the reference cannot travel to the GPU box, the vectors do.
Nothing from the reference is copied: this script imports it, feeds it seeded
inputs through its public constructor (``NMF(W=W0, H=H0)``), and stores inputs
and outputs as plain arrays.

    python tools/make_golden.py            # rewrites tests/golden/

Sets:
  g1_nmf_small   NMF 64x96 r8, every beta branch x regularisation, tol disabled
  g2_cfg1        NMF 256x512 r16 beta=1, 50 iterations (BASELINE configs[0])
  g3_early_stop  tol=1e-4 / max_iter=200: returned n_iter and final factors
  g4_frozen      trainable_W=False and trainable_H=False
  g5_nmfd        NMFD (1,33,50) r4 T=3 ; (1,65,300) r4 T=12 ; (2,20,64) r3 T=5
  g6_beta_div    metrics.beta_div known answers incl. zeros
  g8_convnd      NMF2D (1,4,20,18) r3 k=(3,4), (2,3,12,10) r2 k=(2,2); NMF3D (1,3,8,9,10) r2 k=(2,3,2): 20 iterations
  g9_sparse      NMF.fit on a sparse-COO target (nmf.py:351-398, 602-638), beta in {1, 2}: factors, losses, n_iter
  g11_siplca     plca.SIPLCA / SIPLCA2 / SIPLCA3 (shift-invariant PLCA, plca.py:376-606): plain / priors / frozen Z
  g13_plca_tensor_alpha  PLCA.fit with one-element tensor Dirichlet hyper-parameters + the multi-element error
  g10_plca       plca.PLCA.fit (EM, plca.py:244-304): plain / Dirichlet priors / frozen Z / frozen W
  g12_betamu_chain  trainer.BetaMu.step on a three-layer nn.Sequential of NMF layers (tests/test_trainer.py:10-32)
  g7_betamu      trainer.BetaMu.step on one NMF layer: every beta x penalties, factors after 1 and 5 steps, p.grad
  g14_betamu_conv  trainer.BetaMu.step on ONE convolutive layer (NMFD / NMF2D / NMF3D): beta in {0.5, 1, 2} x penalties,
                 factors after 1 and 3 steps, p.grad of the first step
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get('NMF_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
import torchnmf  # noqa: E402
from torchnmf import nmf as ref_nmf  # noqa: E402
from torchnmf.metrics import beta_div as ref_beta_div  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
NO_STOP = -1e9  # (prev - loss) / loss_init < tol is never true -> all iterations run


class _LossTap:
    """Stand-in for tqdm that records what fit() reports (nmf.py:365, 403-404)."""
    log = []

    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def set_postfix(self, loss=None, **k):
        _LossTap.log.append(float(loss))

    def update(self, n):
        pass


ref_nmf.tqdm = _LossTap


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def run_ref(cls, V, W0, H0, beta, tol, max_iter, alpha=0.0, l1_ratio=0.0, tW=True, tH=True):
    m = cls(W=W0, H=H0, trainable_W=tW, trainable_H=tH)
    _LossTap.log = []
    n = m.fit(V, beta, tol, max_iter, False, alpha, l1_ratio)
    return m.W.data.clone(), m.H.data.clone(), n, list(_LossTap.log)


def loss_of(cls, V, W, H, beta):
    with torch.no_grad():
        return float((ref_beta_div(cls.reconstruct(H, W), V, beta) * 2).sqrt())


def g1():
    g = torch.Generator().manual_seed(1001)
    V = torch.rand(64, 96, generator=g)
    W0 = torch.randn(96, 8, generator=g).abs()
    H0 = torch.randn(64, 8, generator=g).abs()
    out = {'V': V.numpy(), 'W0': W0.numpy(), 'H0': H0.numpy(), 'v_shift_nonpos_beta': np.float32(1e-3)}
    cases = []
    for beta in [-1, 0, 0.5, 1, 1.5, 2, 3]:
        Vb = V + 1e-3 if beta <= 0 else V
        for (alpha, l1r) in [(0, 0), (0.1, 0), (0.1, 0.5), (0.1, 1.0)]:
            ks = [1, 10, 50] if alpha == 0 else [50]
            tag = f'b{beta}_a{alpha}_l{l1r}'
            for k in ks:
                W, H, n, losses = run_ref(ref_nmf.NMF, Vb, W0, H0, beta, NO_STOP, k, alpha, l1r)
                assert n == k
                out[f'{tag}_W{k}'] = W.numpy()
                out[f'{tag}_H{k}'] = H.numpy()
                if k == 50:
                    out[f'{tag}_losses'] = np.array(losses, dtype=np.float64)
            out[f'{tag}_loss_init'] = np.float64(loss_of(ref_nmf.NMF, Vb, W0, H0, beta))
            cases.append(tag)
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, 'g1_nmf_small.npz'), **out)


def g2():
    g = torch.Generator().manual_seed(1002)
    V = bf16_round(torch.rand(256, 512, generator=g))
    W0 = torch.randn(512, 16, generator=g).abs()
    H0 = torch.randn(256, 16, generator=g).abs()
    out = {'V_bf16_bits': V.to(torch.bfloat16).view(torch.int16).numpy(), 'W0': W0.numpy(), 'H0': H0.numpy()}
    for k in (10, 50):
        W, H, n, losses = run_ref(ref_nmf.NMF, V, W0, H0, 1, NO_STOP, k)
        out[f'W{k}'] = W.numpy()
        out[f'H{k}'] = H.numpy()
        out[f'losses{k}'] = np.array(losses, dtype=np.float64)
    out['loss_init'] = np.float64(loss_of(ref_nmf.NMF, V, W0, H0, 1))
    np.savez_compressed(os.path.join(OUT, 'g2_cfg1.npz'), **out)


def g3():
    g = torch.Generator().manual_seed(1003)
    V = torch.rand(64, 96, generator=g)
    W0 = torch.randn(96, 8, generator=g).abs()
    H0 = torch.randn(64, 8, generator=g).abs()
    out = {'V': V.numpy(), 'W0': W0.numpy(), 'H0': H0.numpy()}
    for beta in (0.5, 1, 2):
        W, H, n, losses = run_ref(ref_nmf.NMF, V, W0, H0, beta, 1e-4, 200)
        out[f'b{beta}_n_iter'] = np.int64(n)
        out[f'b{beta}_W'] = W.numpy()
        out[f'b{beta}_H'] = H.numpy()
        out[f'b{beta}_losses'] = np.array(losses, dtype=np.float64)
    # tol = 0 still stops when the loss goes up or stalls (nmf.py:405)
    W, H, n, losses = run_ref(ref_nmf.NMF, V, W0, H0, 1, 0.0, 400)
    out['tol0_n_iter'] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, 'g3_early_stop.npz'), **out)


def g4():
    g = torch.Generator().manual_seed(1004)
    V = torch.rand(48, 80, generator=g)
    W0 = torch.randn(80, 6, generator=g).abs()
    H0 = torch.randn(48, 6, generator=g).abs()
    out = {'V': V.numpy(), 'W0': W0.numpy(), 'H0': H0.numpy()}
    for beta in (1, 2):
        for name, tW, tH in (('frozenW', False, True), ('frozenH', True, False)):
            W, H, n, _ = run_ref(ref_nmf.NMF, V, W0, H0, beta, NO_STOP, 20, tW=tW, tH=tH)
            out[f'b{beta}_{name}_W'] = W.numpy()
            out[f'b{beta}_{name}_H'] = H.numpy()
    np.savez_compressed(os.path.join(OUT, 'g4_frozen.npz'), **out)


def g5():
    out = {}
    shapes = {'doc': ((1, 33, 50), 4, 3), 'mid': ((1, 65, 300), 4, 12), 'batch': ((2, 20, 64), 3, 5)}
    for name, ((B, C, L), R, T) in shapes.items():
        g = torch.Generator().manual_seed(1005 + len(name))
        V = torch.rand(B, C, L, generator=g)
        W0 = torch.randn(C, R, T, generator=g).abs()
        H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
        out[f'{name}_V'], out[f'{name}_W0'], out[f'{name}_H0'] = V.numpy(), W0.numpy(), H0.numpy()
        for beta in (0.5, 1, 2):
            W, H, n, losses = run_ref(ref_nmf.NMFD, V, W0, H0, beta, NO_STOP, 30)
            out[f'{name}_b{beta}_W30'] = W.numpy()
            out[f'{name}_b{beta}_H30'] = H.numpy()
            out[f'{name}_b{beta}_losses'] = np.array(losses, dtype=np.float64)
            out[f'{name}_b{beta}_loss_init'] = np.float64(loss_of(ref_nmf.NMFD, V, W0, H0, beta))
        W, H, n, _ = run_ref(ref_nmf.NMFD, V, W0, H0, 1, NO_STOP, 10, alpha=0.1, l1_ratio=0.5)
        out[f'{name}_reg_W10'], out[f'{name}_reg_H10'] = W.numpy(), H.numpy()
    np.savez_compressed(os.path.join(OUT, 'g5_nmfd.npz'), **out)


def g6():
    g = torch.Generator().manual_seed(1006)
    xs = {'rand': torch.rand(100, generator=g), 'zero': torch.zeros(100)}
    ys = {'rand': torch.rand(100, generator=g), 'zero': torch.zeros(100)}
    out = {'x_rand': xs['rand'].numpy(), 'y_rand': ys['rand'].numpy()}
    for beta in [-1, 0, 0.5, 1, 1.5, 2, 3]:
        for xn, x in xs.items():
            for yn, y in ys.items():
                out[f'b{beta}_x{xn}_y{yn}'] = np.float64(float(ref_beta_div(x, y, beta)))
    np.savez_compressed(os.path.join(OUT, 'g6_beta_div.npz'), **out)


def g7():
    """trainer.BetaMu (trainer.py:35-121) driving a single NMF layer, the closure of tests/test_trainer.py:54-73."""
    from torchnmf.trainer import BetaMu
    g = torch.Generator().manual_seed(1007)
    N, C, R = 48, 40, 6
    V = bf16_round(torch.rand(N, C, generator=g)) + 2.0 ** -7   # strictly positive (beta <= 0 cases)
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    out = {'V': V.numpy(), 'W0': W0.numpy(), 'H0': H0.numpy()}
    cases = []
    for beta in [-1, 0, 0.5, 1, 1.5, 2, 3]:
        for name, (l1, l2, ortho) in {'plain': (0, 0, 0), 'pen': (1e-3, 1e-3, 1e-2)}.items():
            for which in ('both', 'W', 'H'):
                if which != 'both' and name == 'pen':
                    continue
                m = torchnmf.nmf.NMF(W=W0.clone(), H=H0.clone())
                params = list(m.parameters()) if which == 'both' else [getattr(m, which)]
                trainer = BetaMu(params, beta, l1, l2, ortho)

                def closure():
                    trainer.zero_grad()
                    return V, m()
                key = f'b{beta}_{name}_{which}'
                for it in range(1, 6):
                    trainer.step(closure)
                    if it in (1, 5):
                        out[f'{key}_W{it}'] = m.W.detach().numpy().copy()
                        out[f'{key}_H{it}'] = m.H.detach().numpy().copy()
                    if it == 1:
                        for pn in ('W', 'H'):
                            gr = getattr(m, pn).grad
                            if gr is not None and (which in ('both', pn)):
                                out[f'{key}_grad{pn}1'] = gr.detach().numpy().copy()
                cases.append(key)
    out['cases'] = np.array(cases)
    out['param_order'] = np.array([n for n, _ in torchnmf.nmf.NMF((4, 3), rank=2).named_parameters()])
    np.savez_compressed(os.path.join(OUT, 'g7_betamu.npz'), **out)


def g8():
    """NMF2D / NMF3D (nmf.py:782-942): the conv2d / conv3d members of the NMFD family."""
    out = {}
    cases = {'2d_a': (ref_nmf.NMF2D, (1, 4, 20, 18), 3, (3, 4)), '2d_b': (ref_nmf.NMF2D, (2, 3, 12, 10), 2, (2, 2)),
             '3d_a': (ref_nmf.NMF3D, (1, 3, 8, 9, 10), 2, (2, 3, 2))}
    for name, (cls, vshape, R, ks) in cases.items():
        g = torch.Generator().manual_seed(1008 + len(name) + vshape[-1])
        V = torch.rand(*vshape, generator=g)
        B, C = vshape[:2]
        hshape = (B, R) + tuple(l - k + 1 for l, k in zip(vshape[2:], ks))
        W0 = torch.randn(C, R, *ks, generator=g).abs()
        H0 = torch.randn(*hshape, generator=g).abs()
        out[f'{name}_V'], out[f'{name}_W0'], out[f'{name}_H0'] = V.numpy(), W0.numpy(), H0.numpy()
        for beta in (0.5, 1, 2):
            W, H, n, losses = run_ref(cls, V, W0, H0, beta, NO_STOP, 20)
            out[f'{name}_b{beta}_W20'] = W.numpy()
            out[f'{name}_b{beta}_H20'] = H.numpy()
            out[f'{name}_b{beta}_losses'] = np.array(losses, dtype=np.float64)
            out[f'{name}_b{beta}_loss_init'] = np.float64(loss_of(cls, V, W0, H0, beta))
        W, H, n, _ = run_ref(cls, V, W0, H0, 1, NO_STOP, 10, alpha=0.1, l1_ratio=0.5)
        out[f'{name}_reg_W10'], out[f'{name}_reg_H10'] = W.numpy(), H.numpy()
        m = cls(W=W0.clone(), H=H0.clone())
        out[f'{name}_recon'] = m().detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'g8_convnd.npz'), **out)


def g9():
    """Sparse-COO target through the reference's sparse branches (its dense-equals-sparse property is
    tests/test_nmf_sparse.py:8-37; here the sparse run's own outputs are stored, including its loss formula)."""
    g = torch.Generator().manual_seed(1009)
    N, C, R = 120, 90, 5
    D = torch.rand(N, C, generator=g)
    mask = torch.rand(N, C, generator=g) < 0.12
    mask[7, :] = False          # an empty row and an empty column
    mask[:, 11] = False
    idx = torch.nonzero(mask).T
    vals = D[idx[0], idx[1]]
    Vs = torch.sparse_coo_tensor(idx, vals, (N, C)).coalesce()
    W0 = torch.randn(C, R, generator=g).abs()
    H0 = torch.randn(N, R, generator=g).abs()
    out = {'indices': Vs.indices().numpy(), 'values': Vs.values().numpy(), 'shape': np.array([N, C]),
           'W0': W0.numpy(), 'H0': H0.numpy()}
    for beta in (1, 2):
        for tag, (tol, it, alpha, l1r) in {'run': (NO_STOP, 25, 0.0, 0.0), 'reg': (NO_STOP, 10, 0.1, 0.5),
                                            'stop': (1e-3, 200, 0.0, 0.0)}.items():
            W, H, n, losses = run_ref(ref_nmf.NMF, Vs, W0, H0, beta, tol, it, alpha, l1r)
            out[f'b{beta}_{tag}_W'], out[f'b{beta}_{tag}_H'] = W.numpy(), H.numpy()
            out[f'b{beta}_{tag}_n'] = np.int64(n)
            out[f'b{beta}_{tag}_losses'] = np.array(losses, dtype=np.float64)
        m = ref_nmf.NMF(W=W0.clone(), H=H0.clone())
        with torch.no_grad():
            pos, neg = m._sp_recon_beta_pos_neg(Vs, m.H, m.W, beta)
            out[f'b{beta}_loss_init'] = np.float64(float((ref_nmf._get_V_norm(Vs, beta) + pos - neg).mul(2).sqrt()))
    # the generic-beta branch (nmf.py:628-636): the positive term is a dense pass over all of W H^T
    for beta in (0.5, 1.5, 3):
        for tag, (it, alpha, l1r) in {'run': (20, 0.0, 0.0), 'reg': (10, 0.1, 0.5)}.items():
            W, H, n, losses = run_ref(ref_nmf.NMF, Vs, W0, H0, beta, NO_STOP, it, alpha, l1r)
            out[f'b{beta}_{tag}_W'], out[f'b{beta}_{tag}_H'] = W.numpy(), H.numpy()
            out[f'b{beta}_{tag}_n'] = np.int64(n)
            out[f'b{beta}_{tag}_losses'] = np.array(losses, dtype=np.float64)
        m = ref_nmf.NMF(W=W0.clone(), H=H0.clone())
        with torch.no_grad():
            pos, neg = m._sp_recon_beta_pos_neg(Vs, m.H, m.W, beta)
            out[f'b{beta}_loss_init'] = np.float64(float((ref_nmf._get_V_norm(Vs, beta) + pos - neg).mul(2).sqrt()))
    np.savez_compressed(os.path.join(OUT, 'g9_sparse.npz'), **out)


def g10():
    """PLCA (plca.py:311-373) fitted by the reference's EM loop (plca.py:193-304)."""
    from torchnmf import plca as ref_plca
    ref_plca.tqdm = _LossTap
    g = torch.Generator().manual_seed(1010)
    N, C, R = 50, 40, 4
    V = torch.rand(N, C, generator=g)
    W0 = torch.rand(C, R, generator=g)
    H0 = torch.rand(N, R, generator=g)
    Z0 = torch.rand(R, generator=g)
    out = {'V': V.numpy(), 'W0': W0.numpy(), 'H0': H0.numpy(), 'Z0': Z0.numpy()}
    cases = {'plain': dict(), 'prior': dict(W_alpha=1.02, H_alpha=0.99, Z_alpha=1.01),
             'frozenZ': dict(trainable_Z=False), 'frozenW': dict(trainable_W=False), 'stop': dict(tol=1e-3, max_iter=200)}
    for name, kw in cases.items():
        ctor = {k: v for k, v in kw.items() if k.startswith('trainable')}
        fitkw = {k: v for k, v in kw.items() if not k.startswith('trainable')}
        m = ref_plca.PLCA(W=W0.clone(), H=H0.clone(), Z=Z0.clone(), **ctor)
        if name == 'plain':   # the constructor normalises its arguments (plca.py:91, 105, 121)
            out['W_init'], out['H_init'], out['Z_init'] = m.W.data.numpy().copy(), m.H.data.numpy().copy(), m.Z.data.numpy().copy()
            out['recon_init'] = m().detach().numpy().copy()
        _LossTap.log = []
        fitkw.setdefault('tol', NO_STOP)
        fitkw.setdefault('max_iter', 30)
        n, norm = m.fit(V, **fitkw)
        out[f'{name}_W'], out[f'{name}_H'], out[f'{name}_Z'] = m.W.data.numpy().copy(), m.H.data.numpy().copy(), m.Z.data.numpy().copy()
        out[f'{name}_n'], out[f'{name}_norm'] = np.int64(n), np.float64(float(norm))
        out[f'{name}_losses'] = np.array(_LossTap.log, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'g10_plca.npz'), **out)


def g11():
    """SIPLCA, SIPLCA2, SIPLCA3 fitted by the same EM loop (plca.py:193-304) with convNd reconstructions."""
    from torchnmf import plca as ref_plca
    ref_plca.tqdm = _LossTap
    out = {}
    shapes = {'1d': (ref_plca.SIPLCA, (2, 20, 60), 3, (4,)), '2d': (ref_plca.SIPLCA2, (1, 3, 14, 12), 2, (2, 3)),
              '3d': (ref_plca.SIPLCA3, (1, 2, 6, 7, 8), 2, (2, 2, 3))}
    cases = {'plain': ({}, {}), 'prior': ({}, dict(W_alpha=1.02, H_alpha=0.99, Z_alpha=1.01)),
             'frozenZ': (dict(trainable_Z=False), {})}
    for name, (cls, vshape, R, ks) in shapes.items():
        g = torch.Generator().manual_seed(1011 + len(vshape))
        V = torch.rand(*vshape, generator=g)
        W0 = torch.rand(vshape[1], R, *ks, generator=g)
        H0 = torch.rand(vshape[0], R, *[l - k + 1 for l, k in zip(vshape[2:], ks)], generator=g)
        Z0 = torch.rand(R, generator=g)
        out[f'{name}_V'], out[f'{name}_W0'], out[f'{name}_H0'], out[f'{name}_Z0'] = V.numpy(), W0.numpy(), H0.numpy(), Z0.numpy()
        for cname, (ctor, fitkw) in cases.items():
            m = cls(W=W0.clone(), H=H0.clone(), Z=Z0.clone(), **ctor)
            if cname == 'plain':
                out[f'{name}_recon_init'] = m().detach().numpy().copy()
            _LossTap.log = []
            n, norm = m.fit(V, tol=NO_STOP, max_iter=20, **fitkw)
            key = f'{name}_{cname}'
            out[f'{key}_W'], out[f'{key}_H'], out[f'{key}_Z'] = m.W.data.numpy().copy(), m.H.data.numpy().copy(), m.Z.data.numpy().copy()
            out[f'{key}_n'], out[f'{key}_norm'] = np.int64(n), np.float64(float(norm))
            out[f'{key}_losses'] = np.array(_LossTap.log, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'g11_siplca.npz'), **out)


def g12():
    """The reference's own trainer test scenario (tests/test_trainer.py:10-32): nn.Sequential(NMF((100,16),rank=8),
    NMF(W=(32,16)), NMF(W=(50,32))), m(None) chains the layers; BetaMu updates H1, W1, W2, W3 in turn."""
    from torchnmf.trainer import BetaMu
    g = torch.Generator().manual_seed(1012)
    H1, W1 = torch.randn(100, 8, generator=g).abs(), torch.randn(16, 8, generator=g).abs()
    W2, W3 = torch.randn(32, 16, generator=g).abs(), torch.randn(50, 32, generator=g).abs()
    V = torch.rand(100, 50, generator=g) + 2.0 ** -7
    out = {'V': V.numpy(), 'H1': H1.numpy(), 'W1': W1.numpy(), 'W2': W2.numpy(), 'W3': W3.numpy()}
    for beta in (0.5, 1, 2):
        for name, (l1, l2, ortho) in {'plain': (0, 0, 0), 'pen': (1e-3, 1e-3, 1e-2)}.items():
            m = torch.nn.Sequential(torchnmf.nmf.NMF(W=W1.clone(), H=H1.clone()), torchnmf.nmf.NMF(W=W2.clone()),
                                    torchnmf.nmf.NMF(W=W3.clone()))
            trainer = BetaMu(m.parameters(), beta, l1, l2, ortho)

            def closure():
                trainer.zero_grad()
                return V, m(None)
            for it in range(1, 6):
                trainer.step(closure)
                if it in (1, 5):
                    for pn, p in (('W1', m[0].W), ('H1', m[0].H), ('W2', m[1].W), ('W3', m[2].W)):
                        out[f'b{beta}_{name}_{pn}_{it}'] = p.detach().numpy().copy()
            out[f'b{beta}_{name}_gradW3'] = m[2].W.grad.detach().numpy().copy()
    out['param_order'] = np.array([n for n, _ in torch.nn.Sequential(torchnmf.nmf.NMF((4, 3), rank=2), torchnmf.nmf.NMF(W=(5, 3))).named_parameters()])
    np.savez_compressed(os.path.join(OUT, 'g12_betamu_chain.npz'), **out)


def g13():
    """PLCA.fit with tensor-valued Dirichlet hyper-parameters (plca.py:197-199: Union[float, Tensor]).  The reference's EM
    loop tests them with ``if alpha != 1`` (plca.py:257, 271, 285), so tensors work when they hold ONE element and raise a
    RuntimeError otherwise; both behaviours are recorded."""
    from torchnmf import plca as ref_plca
    ref_plca.tqdm = _LossTap
    g = torch.Generator().manual_seed(1313)
    N, C, R = 44, 36, 5
    V = torch.rand(N, C, generator=g)
    W0, H0, Z0 = torch.rand(C, R, generator=g), torch.rand(N, R, generator=g), torch.rand(R, generator=g)
    out = {'V': V.numpy(), 'W0': W0.numpy(), 'H0': H0.numpy(), 'Z0': Z0.numpy()}
    alphas = dict(W_alpha=torch.tensor(1.03), H_alpha=torch.tensor([0.98]), Z_alpha=torch.tensor([1.02]))
    m = ref_plca.PLCA(W=W0.clone(), H=H0.clone(), Z=Z0.clone())
    _LossTap.log = []
    n, norm = m.fit(V, tol=NO_STOP, max_iter=30, **alphas)
    out['W'], out['H'], out['Z'] = m.W.data.numpy().copy(), m.H.data.numpy().copy(), m.Z.data.numpy().copy()
    out['n'], out['norm'] = np.int64(n), np.float64(float(norm))
    out['losses'] = np.array(_LossTap.log, dtype=np.float64)
    out['alphas'] = np.array([float(a.reshape(())) for a in alphas.values()], dtype=np.float64)   # W, H, Z
    m = ref_plca.PLCA(W=W0.clone(), H=H0.clone(), Z=Z0.clone())
    try:
        m.fit(V, tol=NO_STOP, max_iter=3, W_alpha=torch.full((C, R), 1.03))
        out['multi_error'] = np.array('')
    except RuntimeError as e:
        out['multi_error'] = np.array(f'{type(e).__name__}: {e}')
    m = ref_plca.PLCA(W=W0.clone(), H=H0.clone(), Z=Z0.clone())
    try:     # one element, but more dimensions than Z: the in-place add of plca.py:258 cannot broadcast
        m.fit(V, tol=NO_STOP, max_iter=3, Z_alpha=torch.tensor([[1.02]]))
        out['ndim_error'] = np.array('')
    except RuntimeError as e:
        out['ndim_error'] = np.array(f'{type(e).__name__}: {e}')
    np.savez_compressed(os.path.join(OUT, 'g13_plca_tensor_alpha.npz'), **out)


def g14():
    """trainer.BetaMu (trainer.py:35-121) driving one convolutive layer: the backward passes go through conv1d / conv2d / conv3d
    (nmf.py:776-779, 857-860, 937-940)."""
    from torchnmf.trainer import BetaMu
    out = {}
    cases = {'1d': (ref_nmf.NMFD, (2, 9, 30), 3, (4,)), '2d': (ref_nmf.NMF2D, (1, 4, 14, 12), 3, (3, 2)),
             '3d': (ref_nmf.NMF3D, (1, 3, 6, 7, 8), 2, (2, 3, 2))}
    names = []
    for name, (cls, vshape, R, ks) in cases.items():
        g = torch.Generator().manual_seed(1014 + len(name) + vshape[-1])
        V = bf16_round(torch.rand(*vshape, generator=g)) + 2.0 ** -7
        B, C = vshape[:2]
        hshape = (B, R) + tuple(l - k + 1 for l, k in zip(vshape[2:], ks))
        W0 = torch.randn(C, R, *ks, generator=g).abs()
        H0 = torch.randn(*hshape, generator=g).abs()
        out[f'{name}_V'], out[f'{name}_W0'], out[f'{name}_H0'] = V.numpy(), W0.numpy(), H0.numpy()
        for beta in (0.5, 1, 2):
            for pen, (l1, l2, ortho) in {'plain': (0, 0, 0), 'pen': (1e-3, 1e-3, 1e-2)}.items():
                m = cls(W=W0.clone(), H=H0.clone())
                trainer = BetaMu(m.parameters(), beta, l1, l2, ortho)

                def closure():
                    trainer.zero_grad()
                    return V, m()
                key = f'{name}_b{beta}_{pen}'
                for it in range(1, 4):
                    trainer.step(closure)
                    if it in (1, 3):
                        out[f'{key}_W{it}'] = m.W.detach().numpy().copy()
                        out[f'{key}_H{it}'] = m.H.detach().numpy().copy()
                    if it == 1:      # (zero_grad() in the closure drops the earlier parameter's grad: the last one survives)
                        for pn in ('W', 'H'):
                            gr = getattr(m, pn).grad
                            if gr is not None:
                                out[f'{key}_grad{pn}1'] = gr.detach().numpy().copy()
                names.append(key)
        out[f'{name}_param_order'] = np.array([n for n, _ in cls(W=W0.clone(), H=H0.clone()).named_parameters()])
    out['cases'] = np.array(names)
    np.savez_compressed(os.path.join(OUT, 'g14_betamu_conv.npz'), **out)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)  # reproducible summation order
    assert torchnmf.__version__ == '0.3.5', torchnmf.__version__
    # `python tools/make_golden.py g13` regenerates only the named sets (round 5 added g13 without touching the others)
    todo = [fn for fn in (g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14) if len(sys.argv) < 2 or fn.__name__ in sys.argv[1:]]
    for fn in todo:
        fn()
        print('wrote', fn.__name__)
    with open(os.path.join(OUT, 'PROVENANCE.txt'), 'a' if len(sys.argv) > 1 else 'w') as f:
        if len(sys.argv) > 1:
            f.write(f'{" ".join(sys.argv[1:])}: ')
        f.write(f'generated by tools/make_golden.py from torchnmf {torchnmf.__version__} '
                f'(reference mounted at {REF}), torch {torch.__version__}, CPU fp32, 1 thread\n')
