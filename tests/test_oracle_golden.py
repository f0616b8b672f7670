"""Pin the oracle (oracle/mu_oracle.py) against vectors produced by the reference
itself (tools/make_golden.py ran torchnmf 0.3.5 in the build container).

The reference's own tests hold no golden vectors for this path;
these fixtures are the pin.  Unregularised cases agree bit-for-bit on the build
container; the bar written here is 2e-6 relative so the suite is robust to a
different BLAS summation order on another host.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import mu_oracle as O

TOL = 2e-6
NO_STOP = -1e9


def t(x):
    return torch.from_numpy(np.asarray(x))


@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
@pytest.mark.parametrize('reg', [(0, 0), (0.1, 0), (0.1, 0.5), (0.1, 1.0)])
def test_g1_every_beta_branch(beta, reg):
    g = load_golden('g1_nmf_small')
    alpha, l1r = reg
    V = t(g['V']) + (float(g['v_shift_nonpos_beta']) if beta <= 0 else 0.0)
    tag = f'b{beta}_a{alpha}_l{l1r}'
    ks = [1, 10, 50] if alpha == 0 else [50]
    W, H, n, losses, snaps = O.fit(V, t(g['W0']), t(g['H0']), beta, NO_STOP, 50, alpha, l1r, snapshots=ks)
    assert n == 50
    for k in ks:
        assert rel_err(snaps[k][0], g[f'{tag}_W{k}']) < TOL
        assert rel_err(snaps[k][1], g[f'{tag}_H{k}']) < TOL
    assert abs(losses[0] - float(g[f'{tag}_loss_init'])) <= 1e-5 * abs(losses[0])
    np.testing.assert_allclose(losses[1:], g[f'{tag}_losses'], rtol=1e-5)


def test_g2_cfg1():
    g = load_golden('g2_cfg1')
    V = t(g['V_bf16_bits']).view(torch.bfloat16).float()
    W, H, n, losses, snaps = O.fit(V, t(g['W0']), t(g['H0']), 1, NO_STOP, 50, snapshots=[10, 50])
    for k in (10, 50):
        assert rel_err(snaps[k][0], g[f'W{k}']) < TOL
        assert rel_err(snaps[k][1], g[f'H{k}']) < TOL
    np.testing.assert_allclose(losses[1:], g['losses50'], rtol=1e-5)


@pytest.mark.parametrize('beta', [0.5, 1, 2])
def test_g3_early_stop(beta):
    g = load_golden('g3_early_stop')
    W, H, n, losses, _ = O.fit(t(g['V']), t(g['W0']), t(g['H0']), beta, 1e-4, 200)
    assert n == int(g[f'b{beta}_n_iter'])
    assert rel_err(W, g[f'b{beta}_W']) < TOL and rel_err(H, g[f'b{beta}_H']) < TOL


def test_g3_tol0_still_stops():
    g = load_golden('g3_early_stop')
    _, _, n, _, _ = O.fit(t(g['V']), t(g['W0']), t(g['H0']), 1, 0.0, 400)
    assert n == int(g['tol0_n_iter'])


@pytest.mark.parametrize('beta', [1, 2])
@pytest.mark.parametrize('name,tW,tH', [('frozenW', False, True), ('frozenH', True, False)])
def test_g4_frozen(beta, name, tW, tH):
    g = load_golden('g4_frozen')
    W, H, n, _, _ = O.fit(t(g['V']), t(g['W0']), t(g['H0']), beta, NO_STOP, 20, trainable_W=tW, trainable_H=tH)
    assert rel_err(W, g[f'b{beta}_{name}_W']) < TOL and rel_err(H, g[f'b{beta}_{name}_H']) < TOL
    if not tW:
        assert torch.equal(W, t(g['W0']))
    if not tH:
        assert torch.equal(H, t(g['H0']))


@pytest.mark.parametrize('name', ['doc', 'mid', 'batch'])
@pytest.mark.parametrize('beta', [0.5, 1, 2])
def test_g5_nmfd(name, beta):
    g = load_golden('g5_nmfd')
    V, W0, H0 = t(g[f'{name}_V']), t(g[f'{name}_W0']), t(g[f'{name}_H0'])
    W, H, n, losses, _ = O.fit(V, W0, H0, beta, NO_STOP, 30, kind='nmfd')
    assert rel_err(W, g[f'{name}_b{beta}_W30']) < 5e-6
    assert rel_err(H, g[f'{name}_b{beta}_H30']) < 5e-6
    np.testing.assert_allclose(losses[1:], g[f'{name}_b{beta}_losses'], rtol=2e-5)


@pytest.mark.parametrize('name', ['doc', 'mid', 'batch'])
def test_g5_nmfd_regularised(name):
    g = load_golden('g5_nmfd')
    V, W0, H0 = t(g[f'{name}_V']), t(g[f'{name}_W0']), t(g[f'{name}_H0'])
    W, H, _, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 10, alpha=0.1, l1_ratio=0.5, kind='nmfd')
    assert rel_err(W, g[f'{name}_reg_W10']) < 5e-6 and rel_err(H, g[f'{name}_reg_H10']) < 5e-6


@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
def test_g6_beta_div_known_answers(beta):
    g = load_golden('g6_beta_div')
    xs = {'rand': t(g['x_rand']), 'zero': torch.zeros(100)}
    ys = {'rand': t(g['y_rand']), 'zero': torch.zeros(100)}
    for xn, x in xs.items():
        for yn, y in ys.items():
            want = float(g[f'b{beta}_x{xn}_y{yn}'])
            got = float(O.beta_div(x, y, beta))
            assert got == pytest.approx(want, rel=1e-5, abs=1e-5), (beta, xn, yn)
            assert not np.isnan(got) and got >= 0  # tests/test_metrics.py:6-14 of the reference


@pytest.mark.parametrize('world', [2, 8])
@pytest.mark.parametrize('beta', [0.5, 1, 2])
def test_sharded_simulation_matches_unsharded(world, beta):
    """column shards + summed H partials == the single-device iteration."""
    g = load_golden('g1_nmf_small')
    V, W0, H0 = t(g['V']), t(g['W0']), t(g['H0'])
    Ws, Hs = O.nmf_fit_sharded(V, W0, H0, world, beta=beta, n_iter=20)
    W, H, _, _, _ = O.fit(V, W0, H0, beta, NO_STOP, 20)
    assert rel_err(Ws, W) < 5e-6 and rel_err(Hs, H) < 5e-6


@pytest.mark.parametrize('beta', [0, 0.5, 1, 2])
def test_aten_port_equals_closed_form(beta):
    """bench.py times oracle/aten_port.py as the CPU baseline; it must be the same maths as the oracle."""
    from oracle import aten_port
    g = load_golden('g1_nmf_small')
    V = t(g['V']) + (1e-3 if beta <= 0 else 0.0)
    Wp, Hp = aten_port.mu_iterations(V, t(g['W0']), t(g['H0']), beta, 10, alpha=0.1, l1_ratio=0.5)
    W, H, _, _, _ = O.fit(V, t(g['W0']), t(g['H0']), beta, NO_STOP, 10, 0.1, 0.5)
    assert rel_err(Wp, W) < 2e-6 and rel_err(Hp, H) < 2e-6


def _c_oracle():
    import ctypes, os, subprocess
    from conftest import ROOT
    so = os.path.join(ROOT, 'oracle', 'libmu_oracle_c.so')
    if not os.path.exists(so):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True)
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.mu_oracle_c_iterate.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.mu_oracle_c_beta_div.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    lib.mu_oracle_c_beta_div.restype = ctypes.c_double
    return lib, fp


@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
def test_c_oracle_against_golden(beta):
    """oracle/mu_oracle_c.c (scalar loops, its own summation order) reproduces the reference's vectors too."""
    import ctypes
    lib, fp = _c_oracle()
    g = load_golden('g1_nmf_small')
    V = (t(g['V']) + (float(g['v_shift_nonpos_beta']) if beta <= 0 else 0.0)).contiguous()
    W, H = t(g['W0']).clone().contiguous(), t(g['H0']).clone().contiguous()
    N, C = V.shape
    R = W.shape[1]
    p = lambda x: ctypes.cast(x.data_ptr(), fp)
    loss0 = lib.mu_oracle_c_beta_div(p(V), p(W), p(H), N, C, R, beta)
    assert (2 * loss0) ** 0.5 == pytest.approx(float(g[f'b{beta}_a0.1_l0.5_loss_init']), rel=2e-5)
    lib.mu_oracle_c_iterate(p(V), p(W), p(H), N, C, R, beta, 0.05, 0.05, 50, 1, 1)
    assert rel_err(W, g[f'b{beta}_a0.1_l0.5_W50']) < 2e-5 and rel_err(H, g[f'b{beta}_a0.1_l0.5_H50']) < 2e-5


# ---- trainer.BetaMu on one layer --------------------------------------------------
G7_PEN = {'plain': (0.0, 0.0, 0.0), 'pen': (1e-3, 1e-3, 1e-2)}


def g7_cases():
    return [str(c) for c in load_golden('g7_betamu')['cases']]


@pytest.mark.parametrize('case', g7_cases())
def test_g7_betamu_oracle(case):
    """oracle.betamu_step restates trainer.py:35-121 for a single NMF layer: pinned on the reference's outputs."""
    g = load_golden('g7_betamu')
    b, pen, which = case.split('_')
    beta = float(b[1:])
    l1, l2, ortho = G7_PEN[pen]
    params = ('W', 'H') if which == 'both' else (which,)
    V, W, H = (torch.from_numpy(g[k]) for k in ('V', 'W0', 'H0'))
    assert list(g['param_order']) == ['W', 'H']
    for it in range(1, 6):
        W, H, grads = O.betamu_step(V, W, H, beta, l1, l2, ortho, params)
        if it in (1, 5):
            assert rel_err(W, g[f'{case}_W{it}']) < 2e-6 and rel_err(H, g[f'{case}_H{it}']) < 2e-6
        if it == 1:
            for n in params:
                if f'{case}_grad{n}1' in g:
                    assert rel_err(grads[n], g[f'{case}_grad{n}1']) < 2e-6


@pytest.mark.parametrize('beta', [-1, 0, 0.5, 1, 1.5, 2, 3])
@pytest.mark.parametrize('pen', ['plain', 'pen'])
def test_g7_aten_port_is_bit_identical(beta, pen):
    """The CPU-baseline port of BetaMu (bench.py --workload betamu) reproduces the reference bit for bit."""
    from oracle import aten_port
    torch.set_num_threads(1)
    g = load_golden('g7_betamu')
    V, W0, H0 = (torch.from_numpy(g[k]) for k in ('V', 'W0', 'H0'))
    W, H = aten_port.betamu_iterations(V, W0, H0, beta, 5, *G7_PEN[pen])
    assert np.array_equal(W.numpy(), g[f'b{beta}_{pen}_both_W5']) and np.array_equal(H.numpy(), g[f'b{beta}_{pen}_both_H5'])


# ---- NMF2D / NMF3D ------------------------------------------------------------------
@pytest.mark.parametrize('name', ['2d_a', '2d_b', '3d_a'])
@pytest.mark.parametrize('beta', [0.5, 1, 2])
def test_g8_convnd_oracle(name, beta):
    g = load_golden('g8_convnd')
    V, W0, H0 = (torch.from_numpy(g[f'{name}_{k}']) for k in ('V', 'W0', 'H0'))
    assert rel_err(O.convnd_reconstruct(H0, W0), g[f'{name}_recon']) < 2e-6
    W, H, n, losses, _ = O.fit(V, W0, H0, beta, NO_STOP, 20, kind='convnd')
    assert n == 20
    assert rel_err(W, g[f'{name}_b{beta}_W20']) < 1e-5 and rel_err(H, g[f'{name}_b{beta}_H20']) < 1e-5
    assert np.allclose(losses[1:], g[f'{name}_b{beta}_losses'], rtol=1e-5)
    if beta == 1:
        W, H, _, _, _ = O.fit(V, W0, H0, 1, NO_STOP, 10, 0.1, 0.5, kind='convnd')
        assert rel_err(W, g[f'{name}_reg_W10']) < 1e-5 and rel_err(H, g[f'{name}_reg_H10']) < 1e-5


# ---- sparse-COO target ---------------------------------------------------------------
@pytest.mark.parametrize('beta', [1, 2])
@pytest.mark.parametrize('tag,args', [('run', (NO_STOP, 25, 0.0, 0.0)), ('reg', (NO_STOP, 10, 0.1, 0.5)),
                                      ('stop', (1e-3, 200, 0.0, 0.0))])
def test_g9_sparse_oracle(beta, tag, args):
    g = load_golden('g9_sparse')
    idx, vals = torch.from_numpy(g['indices']), torch.from_numpy(g['values'])
    W0, H0 = torch.from_numpy(g['W0']), torch.from_numpy(g['H0'])
    assert O.sp_fit_loss(idx, vals, W0, H0, beta) == pytest.approx(float(g[f'b{beta}_loss_init']), rel=1e-5)
    W, H, n, losses = O.sp_fit(idx, vals, tuple(g['shape']), W0, H0, beta, *args)
    assert n == int(g[f'b{beta}_{tag}_n'])
    assert rel_err(W, g[f'b{beta}_{tag}_W']) < 5e-6 and rel_err(H, g[f'b{beta}_{tag}_H']) < 5e-6
    assert np.allclose(losses[1:], g[f'b{beta}_{tag}_losses'], rtol=1e-5)
    # the property the reference tests (tests/test_nmf_sparse.py:8-37): the sparse updates equal the dense ones
    V = torch.sparse_coo_tensor(idx, vals, tuple(g['shape'])).to_dense()
    Wd, Hd, _, _, _ = O.fit(V, W0, H0, beta, NO_STOP, 5)
    Ws, Hs, _, _ = O.sp_fit(idx, vals, tuple(g['shape']), W0, H0, beta, NO_STOP, 5)
    assert rel_err(Ws, Wd) < 5e-6 and rel_err(Hs, Hd) < 5e-6


# ---- PLCA --------------------------------------------------------------------------
G10_CASES = {'plain': {}, 'prior': dict(W_alpha=1.02, H_alpha=0.99, Z_alpha=1.01), 'frozenZ': dict(train=(True, True, False)),
             'frozenW': dict(train=(False, True, True)), 'stop': dict(tol=1e-3, max_iter=200)}


@pytest.mark.parametrize('name', list(G10_CASES))
def test_g10_plca_oracle(name):
    g = load_golden('g10_plca')
    V, W0, H0, Z0 = (torch.from_numpy(g[k]) for k in ('V', 'W0', 'H0', 'Z0'))
    kw = dict(G10_CASES[name])
    kw.setdefault('tol', NO_STOP)
    kw.setdefault('max_iter', 30)
    W, H, Z, n, norm, losses = O.plca_fit(V, W0, H0, Z0, **kw)
    assert n == int(g[f'{name}_n']) and norm == pytest.approx(float(g[f'{name}_norm']), rel=1e-6)
    for t_, k in ((W, 'W'), (H, 'H'), (Z, 'Z')):
        assert rel_err(t_, g[f'{name}_{k}']) < 5e-6
    assert np.allclose(losses[1:], g[f'{name}_losses'], rtol=1e-5)


def test_g13_plca_tensor_alpha_oracle():
    """PLCA.fit with one-element TENSOR Dirichlet hyper-parameters (plca.py:197-199), generated from the reference: the
    oracle, fed the tensors' values, lands on the same factors (the reference's fp32 ``alpha - 1`` against the oracle's
    differs in the last bit of the added constant only)."""
    g = load_golden('g13_plca_tensor_alpha')
    V, W0, H0, Z0 = (torch.from_numpy(g[k]) for k in ('V', 'W0', 'H0', 'Z0'))
    wa, ha, za = (float(x) for x in g['alphas'])
    W, H, Z, n, norm, losses = O.plca_fit(V, W0, H0, Z0, tol=NO_STOP, max_iter=30, W_alpha=wa, H_alpha=ha, Z_alpha=za)
    assert n == int(g['n']) and norm == pytest.approx(float(g['norm']), rel=1e-6)
    for t_, k in ((W, 'W'), (H, 'H'), (Z, 'Z')):
        assert rel_err(t_, g[k]) < 5e-6
    assert np.allclose(losses[1:], g['losses'], rtol=1e-5)
    assert str(g['multi_error']).startswith('RuntimeError: Boolean value of Tensor with more than one value is ambiguous')
    assert 'broadcast shape' in str(g['ndim_error'])


@pytest.mark.parametrize('name', ['1d', '2d', '3d'])
@pytest.mark.parametrize('case', ['plain', 'prior', 'frozenZ'])
def test_g11_siplca_oracle(name, case):
    """SIPLCA / SIPLCA2 / SIPLCA3 (plca.py:376-606): the PLCA EM step on the convNd reconstruction."""
    g = load_golden('g11_siplca')
    V, W0, H0, Z0 = (torch.from_numpy(g[f'{name}_{k}']) for k in ('V', 'W0', 'H0', 'Z0'))
    kw = {'plain': {}, 'prior': dict(W_alpha=1.02, H_alpha=0.99, Z_alpha=1.01), 'frozenZ': dict(train=(True, True, False))}[case]
    W, H, Z, n, norm, losses = O.plca_fit(V, W0, H0, Z0, tol=NO_STOP, max_iter=20, **kw)
    assert n == int(g[f'{name}_{case}_n'])
    for t_, k in ((W, 'W'), (H, 'H'), (Z, 'Z')):
        assert rel_err(t_, g[f'{name}_{case}_{k}']) < 1e-5
    assert np.allclose(losses[1:], g[f'{name}_{case}_losses'], rtol=1e-5)


@pytest.mark.parametrize('beta', [0.5, 1.5, 3])
@pytest.mark.parametrize('tag,args', [('run', (NO_STOP, 20, 0.0, 0.0)), ('reg', (NO_STOP, 10, 0.1, 0.5))])
def test_g9_sparse_oracle_generic_beta(beta, tag, args):
    """The generic-beta branch (nmf.py:628-636): numerator over the stored entries, positive term over every entry."""
    g = load_golden('g9_sparse')
    idx, vals = torch.from_numpy(g['indices']), torch.from_numpy(g['values'])
    W0, H0 = torch.from_numpy(g['W0']), torch.from_numpy(g['H0'])
    assert O.sp_fit_loss(idx, vals, W0, H0, beta) == pytest.approx(float(g[f'b{beta}_loss_init']), rel=1e-5)
    W, H, n, losses = O.sp_fit(idx, vals, tuple(g['shape']), W0, H0, beta, *args)
    assert n == int(g[f'b{beta}_{tag}_n'])
    assert rel_err(W, g[f'b{beta}_{tag}_W']) < 1e-5 and rel_err(H, g[f'b{beta}_{tag}_H']) < 1e-5
    assert np.allclose(losses[1:], g[f'b{beta}_{tag}_losses'], rtol=1e-5)


@pytest.mark.parametrize('beta', [0.5, 1, 2])
@pytest.mark.parametrize('pen', ['plain', 'pen'])
def test_g12_betamu_chain_oracle(beta, pen):
    """BetaMu over nn.Sequential of three NMF layers (tests/test_trainer.py:10-32 of the reference)."""
    g = load_golden('g12_betamu_chain')
    assert list(g['param_order']) == ['0.W', '0.H', '1.W']          # torch's order: W before H
    V, X0 = torch.from_numpy(g['V']), torch.from_numpy(g['H1'])
    Ws = [torch.from_numpy(g[k]) for k in ('W1', 'W2', 'W3')]
    for it in range(1, 6):
        X0, Ws, grads = O.betamu_chain_step(V, X0, Ws, beta, *G7_PEN[pen])
        if it in (1, 5):
            for pn, t_ in (('W1', Ws[0]), ('H1', X0), ('W2', Ws[1]), ('W3', Ws[2])):
                assert rel_err(t_, g[f'b{beta}_{pen}_{pn}_{it}']) < 1e-4
    assert rel_err(grads['W3'], g[f'b{beta}_{pen}_gradW3']) < 1e-3


def g14_cases():
    return [str(c) for c in load_golden('g14_betamu_conv')['cases']]


@pytest.mark.parametrize('case', g14_cases())
def test_g14_betamu_conv_oracle(case):
    """oracle.mu_oracle.betamu_conv_step against the reference's trainer.BetaMu on ONE convolutive layer (NMFD / NMF2D /
    NMF3D; golden g14, tools/make_golden.py g14): factors after 1 and 3 steps, p.grad of the first step's last parameter."""
    from oracle import mu_oracle as O
    g = load_golden('g14_betamu_conv')
    name, bs, pen = case.split('_')
    beta = float(bs[1:])
    l1, l2, ortho = (1e-3, 1e-3, 1e-2) if pen == 'pen' else (0, 0, 0)
    V, W, H = t(g[f'{name}_V']), t(g[f'{name}_W0']), t(g[f'{name}_H0'])
    order = tuple(str(x) for x in g[f'{name}_param_order'])
    for it in range(1, 4):
        W, H, grads = O.betamu_conv_step(V, W, H, beta, l1, l2, ortho, params=order)
        if it in (1, 3):
            assert rel_err(W, t(g[f'{case}_W{it}'])) < 2e-6 and rel_err(H, t(g[f'{case}_H{it}'])) < 2e-6
        if it == 1:
            for pn in ('W', 'H'):
                if f'{case}_grad{pn}1' in g.files:
                    want = t(g[f'{case}_grad{pn}1'])
                    assert float((grads[pn] - want).abs().max()) < 2e-5 * float(want.abs().max())
