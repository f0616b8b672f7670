"""CPU tests of the host side: module surface (mirrors the reference's tests/test_nmf.py constructor tests),
error behaviour, the fit() driver semantics (with the oracle-backed stand-in backend), and the C ABI surface."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, rel_err
from torchnmf_amd import _capi, engine
from torchnmf_amd import nmf as anmf
from torchnmf_amd.nmf import NMF, NMFD, BaseComponent


def t(x):
    return torch.from_numpy(np.asarray(x))


# ---- constructors: reference tests/test_nmf.py:8-66 -----------------------------------------------------
@pytest.mark.parametrize('W', [(50, 8), torch.rand(50, 8), None])
@pytest.mark.parametrize('H', [(100, 8), torch.rand(100, 8), None])
def test_base_valid_construct(W, H):
    m = BaseComponent(8, W, H)
    assert (m.H is None) == (H is None) and (m.W is None) == (W is None)
    assert m.rank == 8


@pytest.mark.parametrize('rank, W, H', [
    (None, None, None),
    (None, (50, 8), (100, 10)),
    (None, torch.rand(50, 8), (100, 10)),
    (None, torch.randn(50, 8), (100, 8)),
    (None, (50, 8), torch.rand(100, 10)),
    (None, (50, 8), torch.randn(100, 8)),
    (None, torch.rand(50, 8), torch.rand(100, 10)),
    (None, torch.randn(50, 8), torch.rand(100, 8)),
    (None, torch.rand(50, 8), torch.randn(100, 8)),
])
def test_base_invalid_construct(rank, W, H):
    with pytest.raises(AssertionError):
        BaseComponent(rank, W, H)


def test_nmf_shapes_and_attributes():
    m = NMF((100, 50))
    assert m.W.shape == (50, 50) and m.H.shape == (100, 50) and m.rank == 50 and m.out_channels == 50
    m = NMF((20, 30), 5)
    assert m.W.shape == (30, 5) and m.H.shape == (20, 5)
    assert bool(torch.all(m.W >= 0)) and bool(torch.all(m.H >= 0))
    assert 'out_channels=30' in repr(m)
    for bad in [(100, 50, 50), (100,)]:
        with pytest.raises(Exception):
            NMF(bad)


def test_nmfd_shapes_and_attributes():
    m = NMFD((1, 33, 50), 16, 3)
    assert m.W.shape == (33, 16, 3) and m.H.shape == (1, 16, 48) and m.kernel_size == (3,)
    assert 'kernel_size=(3,)' in repr(m)
    for bad in [(100, 50), (100,), (100, 50) * 2]:
        with pytest.raises(Exception):
            NMFD(bad)


def test_trainable_flags_and_state_dict_roundtrip():
    W0, H0 = torch.rand(30, 4), torch.rand(20, 4)
    m = NMF(W=W0, H=H0, trainable_W=False)
    assert not m.W.requires_grad and m.H.requires_grad
    assert torch.equal(m.W.data, W0) and m.W.data.data_ptr() != W0.data_ptr()
    m2 = NMF((20, 30), 4)
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m2.W.data, W0) and torch.equal(m2.H.data, H0)
    assert [n for n, _ in m.named_parameters()] == ['W', 'H']


def test_forward_needs_both_factors():
    with pytest.raises(AssertionError):
        BaseComponent(4, (10, 4), None)()


def test_no_cpu_fallback():
    """Compute entry points refuse CPU tensors instead of silently computing on the host."""
    m = NMF((20, 30), 4)
    with pytest.raises(RuntimeError):
        m()
    with pytest.raises(RuntimeError):
        m.fit(torch.rand(20, 30))
    from torchnmf_amd.metrics import beta_div
    with pytest.raises(RuntimeError):
        beta_div(torch.rand(5), torch.rand(5), 1)
    with pytest.raises(RuntimeError):
        engine.HipBackend()          # no ROCm device in this container
    with pytest.raises(RuntimeError):                      # sparse targets go to the device path too
        m.fit(torch.rand(20, 30).to_sparse())
    with pytest.raises(NotImplementedError):
        m.sparse_fit(torch.rand(20, 30))


def test_eps_is_the_same_on_both_sides_of_the_abi():
    """constants.eps (reference constants.py:3) and nmfmu::kEps in the HIP sources are the same fp32 value."""
    from torchnmf_amd.constants import eps
    src = open(os.path.join(ROOT, 'pytorch-nmf_amd', 'csrc', 'nmfmu_fused.h')).read()
    k = float(re.search(r'constexpr float kEps = ([0-9.e+-]+)f;', src).group(1))
    assert np.float32(k) == np.float32(eps) == np.float32(torch.finfo(torch.float32).eps)


# ---- the C ABI surface ------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'nmfmu.h')).read()
    declared = set(re.findall(r'\b(nmfmu_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_static_queries():
    lib = _capi.load()
    assert lib.nmfmu_abi_version() == _capi.ABI_VERSION == 9
    assert lib.nmfmu_abi_check(9) == 0 and lib.nmfmu_abi_check(8) == _capi.ERR_ARG     # load-time guard of foreign bindings
    assert [lib.nmfmu_pad_rows(r) for r in (1, 256, 257, 4096)] == [256, 256, 512, 4096]
    assert [lib.nmfmu_pad_rank(r) for r in (1, 32, 33, 88, 128, 129, 256)] == [32, 32, 64, 128, 128, 256, 256]
    assert lib.nmfmu_pad_rank(257) == _capi.ERR_UNSUPPORTED
    assert [lib.nmfmu_beta_kind(b) for b in (1.0, 2.0, 0.0, 0.5, -1.0)] == [0, 1, 2, 3, 3]
    assert lib.nmfmu_supported(128, _capi.PREC_BF16X3) == 1 and lib.nmfmu_supported(256, _capi.PREC_BF16X3) == 0
    assert lib.nmfmu_supported(256, _capi.PREC_F16) == 1 and lib.nmfmu_supported(256, _capi.PREC_BF16) == 1
    # tile heights: the eight-wave ping-pong kernel (256 rows) serves beta == 1 with one operand plane up to rank pad 128
    assert lib.nmfmu_block_rows(128, _capi.PREC_F16, 1.0) == 256 and lib.nmfmu_block_rows(128, _capi.PREC_BF16, 1.0) == 256
    assert lib.nmfmu_block_rows(128, _capi.PREC_F16, 2.0) == 128 and lib.nmfmu_block_rows(256, _capi.PREC_F16, 1.0) == 128
    assert lib.nmfmu_block_rows(128, _capi.PREC_BF16X3, 1.0) == 128
    # fp16 operands with an fp32 target (round 4): four-wave kernel at every rank pad, X stored at 4 bytes per element
    assert lib.nmfmu_supported(256, _capi.PREC_F16X) == 1 and lib.nmfmu_block_rows(128, _capi.PREC_F16X, 1.0) == 128
    assert lib.nmfmu_xp_bytes(256, 512, _capi.PREC_F16X) == 4 * 256 * 512 == 2 * lib.nmfmu_xp_bytes(256, 512, _capi.PREC_F16)
    assert lib.nmfmu_debug_set_buffer(None) == _capi.ERR_UNSUPPORTED   # diagnostic hook: NMFMU_DEBUG_HOOKS builds only
    assert lib.nmfmu_choose_nsplit(4096, 65536, 128, 256) == 16      # 32 owner blocks x 16 chunks = 512 workgroups
    assert lib.nmfmu_choose_nsplit(65536, 4096, 128, 256) == 1
    assert lib.nmfmu_choose_nsplit(256, 256, 128, 256) == 1          # tiny problems are not split below 4 tiles
    assert lib.nmfmu_xp_bytes(4096, 65536, _capi.PREC_BF16) == 4096 * 65536 * 2
    assert lib.nmfmu_slab_bytes(4096, 128, 16) == 16 * 4096 * 128 * 4
    # struct layouts agree with the header (sizeof via a known-good packing: 7 pointers + 2 int32)
    assert ctypes.sizeof(_capi.Factor) == 64 and ctypes.sizeof(_capi.Step) == 8 + 64 * 2 + 16 + 4 * 6 + 4 * 4 + 8 + 8   # (+ stamps, ABI 9)


# ---- fit() driver semantics on the stand-in backend -------------------------------------------------------
@pytest.fixture
def cpu_engine(monkeypatch):
    from cpu_backend import OracleBackend
    monkeypatch.setattr(engine, 'DEFAULT_BACKEND_FACTORY', OracleBackend)
    monkeypatch.setattr(anmf, '_require_device', lambda t_, what: None)


def test_static_queries_of_the_gemm_engine():
    """Round-2 host-side queries of the NMFD path (no device work): which shapes take which kernel, buffer sizes."""
    lib = _capi.load()
    # per-tile diagonal sums instead of the Y matrix: >= 128 taps (a 128 x 128 tile then holds at most two ranks / batches)
    assert lib.nmfmu_fold_parts_supported(1, 8, 7793, 400) == 1 and lib.nmfmu_fold_parts_supported(1, 8, 7793, 127) == 0
    assert lib.nmfmu_fold_parts_supported(0, 8, 100, 400) == 0
    assert lib.nmfmu_fold_part_bytes(3200, 8192) == 25 * 64 * 4 * 256 * 4          # 4 segments x 256 diagonals per tile
    assert lib.nmfmu_fold_hsum_parts(2, 7793) == 2 * 31
    # ragged channels: LDS for eight (W row + H window) pairs must fit 64 KiB
    assert lib.nmfmu_conv_ragged_supported(8, 400) == 1 and lib.nmfmu_conv_ragged_supported(8, 1100) == 0
    assert lib.nmfmu_conv_ragged_blocks(1, 7793, 400) == 128 and lib.nmfmu_conv_ragged_blocks(3, 201, 400) == 3 * 10
    # fp16 operands: the beta == 1 iteration on implicit operands
    E, O = _capi, _capi
    assert lib.nmfmu_gemm_f16_supported(1.0, E.EPI_RATIO, O.OPS_B_HU) == 1
    assert lib.nmfmu_gemm_f16_supported(1.0, E.EPI_RATIO, O.OPS_PLANES) == 0
    assert lib.nmfmu_gemm_f16_supported(2.0, E.EPI_RATIO, O.OPS_B_HU) == 0
    assert lib.nmfmu_gemm_f16_supported(2.0, E.EPI_FOLD, O.OPS_PLANES) == 1         # beta-independent epilogues
    # descriptor layout: 12 pointers / 64-bit slots first, then int32 fields (header order)
    d = _capi.GemmDesc()
    assert [f[0] for f in d._fields_][-15:] == ['tile_rows', 'n_ld', 'k_len', 'k_split', 'tail_rows', 'rag_c0', 'rag_channels',
                                                'win_nd', 'win_lh', 'win_taps', 'win_channels', 'win_pitch', 'win_fold', 't_koff',
                                                'stage_mode']
    # window staging of an implicit operand (round 5, ABI 8): the library's shape test, host only.  configs[3] -- 1025 bins,
    # 8192 frames, rank 8, 400 taps: both reconstructions (the GEMM runs over the 1024 whole channels), the loss and the W
    # numerator qualify; stage_mode = 1 keeps the chunk-major tiles; padding inside the implicit operand's tiles disqualifies
    import ctypes as C
    one = C.c_char_p(b'x')       # any non-null pointers: nothing is dereferenced by the query

    def desc(ops, m_pad, n_pad, k_pad, B, R, T, Lh, precision=_capi.PREC_BF16, **kw):
        p = C.cast(one, C.c_void_p)
        dd = _capi.GemmDesc(p, p, p, p, m_pad, n_pad, k_pad, precision, 1.0, p, p, p, p, p, p, 0, 0, ops, B, R, T, Lh, 128, 0, 0, 0, 0, 0, 0)
        for k, v in kw.items():
            setattr(dd, k, v)
        return dd
    q = lambda dd, epi: lib.nmfmu_gemm_window_staged(C.byref(dd), epi)
    assert q(desc(O.OPS_B_HU, 1024, 8192, 3200, 1, 8, 400, 7793), E.EPI_RATIO) == 1
    assert q(desc(O.OPS_A_HU, 8192, 1024, 3200, 1, 8, 400, 7793), E.EPI_RATIO) == 1
    assert q(desc(O.OPS_B_HU, 1024, 8192, 3200, 1, 8, 400, 7793), E.EPI_LOSS) == 1
    assert q(desc(O.OPS_B_HUT, 1152, 3200, 8192, 1, 8, 400, 7793, k_split=2), E.EPI_F32) == 1
    assert q(desc(O.OPS_B_HU, 1024, 8192, 3200, 1, 8, 400, 7793, stage_mode=1), E.EPI_RATIO) == 0
    assert q(desc(O.OPS_B_HU, 1024, 8192, 3200, 1, 8, 400, 7793, stage_mode=2), E.EPI_RATIO) == _capi.ERR_ARG
    assert q(desc(O.OPS_B_HU, 128, 8320, 3200, 1, 8, 400, 7800 + 121), E.EPI_RATIO) == 1      # L = 8320 = 65 tiles
    assert q(desc(O.OPS_B_HU, 128, 8064, 3200, 1, 8, 400, 7601), E.EPI_RATIO) == 0            # L = 8000: padding rows in the last tile
    assert q(desc(O.OPS_B_HU, 128, 256, 512, 1, 8, 56, 201), E.EPI_RATIO) == 0                # fewer than 64 taps
    assert q(desc(O.OPS_B_HU, 128, 256, 640, 1, 5, 104, 153), E.EPI_RATIO) == 0               # R T = 520: a dead k-chunk in the last k-tile
    assert q(desc(O.OPS_B_HUT, 128, 640, 256, 1, 8, 72, 185), E.EPI_F32) == 0                 # rows (r,t) need >= 128 taps
    assert q(desc(O.OPS_B_HUT, 128, 1152, 384, 1, 8, 136, 249), E.EPI_F32) == 0               # R T = 1088: padding rows
    assert q(desc(O.OPS_PLANES, 128, 128, 128, 1, 8, 400, 7793), E.EPI_F32) == 0
    # ragged channels inside the GEMM grid: eight workgroups share out a tile's frames -> >= 8 tiles of the explicit operand
    assert lib.nmfmu_gemm_ragged_supported(O.OPS_B_HU, 1024, 8192, 1) == 1           # configs[3], W half-step
    assert lib.nmfmu_gemm_ragged_supported(O.OPS_A_HU, 8192, 1024, 1) == 1           # ... H half-step
    assert lib.nmfmu_gemm_ragged_supported(O.OPS_B_HU, 896, 8192, 1) == 0
    assert lib.nmfmu_gemm_ragged_supported(O.OPS_A_HU, 8192, 896, 1) == 0
    assert lib.nmfmu_gemm_ragged_supported(O.OPS_B_HU, 1024, 8192, 17) == 0 and lib.nmfmu_gemm_ragged_supported(O.OPS_B_HU, 1024, 8192, 0) == 0
    assert lib.nmfmu_gemm_ragged_supported(O.OPS_PLANES, 1024, 8192, 1) == 0 and lib.nmfmu_gemm_ragged_supported(O.OPS_B_HUT, 1024, 8192, 1) == 0
    # argument checking happens before any device work
    assert lib.nmfmu_gemm(None, 0, None) == _capi.ERR_ARG
    assert lib.nmfmu_conv_ragged_rows(None, 1, 1, 1, None, 1, 1, 0, 0, 1.0, 0, None, 0, None, None, None, None, None,
                                      None) == _capi.ERR_ARG


@pytest.mark.parametrize('beta', [0.5, 1, 2])
def test_fit_loop_early_stop_matches_reference(cpu_engine, beta):
    g = load_golden('g3_early_stop')
    m = NMF(W=t(g['W0']), H=t(g['H0']))
    n = m.fit(t(g['V']), beta, 1e-4, 200)
    assert n == int(g[f'b{beta}_n_iter'])
    assert rel_err(m.W.data, g[f'b{beta}_W']) < 5e-6 and rel_err(m.H.data, g[f'b{beta}_H']) < 5e-6


@pytest.mark.parametrize('beta', [1, 2])     # (G3's beta = 0.5 run never stops within its 200 iterations)
def test_fit_loop_async_checkpoints_equal_the_synchronous_loop(cpu_engine, beta, monkeypatch):
    """Round 4: the loss checkpoints of fit() are judged one checkpoint late (no host sync in the loop) and a stop is
    rolled back to the snapshot of its checkpoint.  Count and factors must equal the synchronous loop's bit for bit --
    when the stop rule fires (golden G3), when max_iter ends the loop with a checkpoint still unjudged (max_iter just past
    the stopping checkpoint: the pending one must still stop the fit there), and when it never fires."""
    from torchnmf_amd import engine
    g = load_golden('g3_early_stop')
    n_stop = int(g[f'b{beta}_n_iter'])
    assert n_stop % 10 == 0 and n_stop >= 20
    calls = {'begin': 0, 'rollback': 0}
    for name in ('checkpoint_begin', 'rollback'):
        orig = getattr(engine.AsyncLossMixin, name)
        monkeypatch.setattr(engine.AsyncLossMixin, name,
                            (lambda o, k: lambda self: (calls.__setitem__(k, calls[k] + 1), o(self))[1])(orig, name.split('_')[-1]))
    for max_iter, tol in [(200, 1e-4), (n_stop + 3, 1e-4), (n_stop + 10, 1e-4), (n_stop - 5, 1e-4), (40, -1e9)]:
        out = {}
        for mode in ('1', '0'):
            monkeypatch.setenv('TORCHNMF_AMD_ASYNC_LOSS', mode)
            calls.update(begin=0, rollback=0)
            m = NMF(W=t(g['W0']), H=t(g['H0']))
            n = m.fit(t(g['V']), beta, tol, max_iter)
            out[mode] = (n, m.W.data.clone(), m.H.data.clone(), dict(calls))
        (na, Wa, Ha, ca), (ns, Ws, Hs, cs) = out['1'], out['0']
        assert na == ns == (min(n_stop, max_iter) if tol > 0 else max_iter), (max_iter, na, ns)
        assert torch.equal(Wa, Ws) and torch.equal(Ha, Hs), max_iter
        assert cs == {'begin': 0, 'rollback': 0}
        assert ca['rollback'] == (1 if (tol > 0 and max_iter >= n_stop + 1) else 0), (max_iter, ca)
        assert ca['begin'] >= 1


@pytest.mark.parametrize('name,tW,tH', [('frozenW', False, True), ('frozenH', True, False)])
def test_fit_loop_respects_frozen_factors(cpu_engine, name, tW, tH):
    g = load_golden('g4_frozen')
    m = NMF(W=t(g['W0']), H=t(g['H0']), trainable_W=tW, trainable_H=tH)
    m.fit(t(g['V']), 1, -1e9, 20)
    assert rel_err(m.W.data, g[f'b1_{name}_W']) < 5e-6 and rel_err(m.H.data, g[f'b1_{name}_H']) < 5e-6


def test_fit_loop_validation_and_verbose(cpu_engine, capsys):
    m = NMF((20, 30), 4)
    V = torch.rand(20, 30)
    assert m.fit(V, 1, 0, 30, verbose=True) <= 30
    V[0, 0] = 0
    with pytest.raises(ValueError):
        m.fit(V, beta=0)
    V[0, 0] = -1
    with pytest.raises(AssertionError):
        m.fit(V)
    with pytest.raises(ValueError):
        m.fit(torch.rand(20, 30), precision='fp64')


def test_nmf2d_nmf3d_constructors():
    """nmf.py:842-855, 922-935: shape inference, rank default (the first shift extent), kernel_size attribute."""
    from torchnmf_amd.nmf import NMF2D, NMF3D
    m = NMF2D((1, 1, 33, 50), 16, 3)
    assert tuple(m.W.shape) == (1, 16, 3, 3) and tuple(m.H.shape) == (1, 16, 31, 48) and m.kernel_size == (3, 3)
    assert NMF2D((2, 5, 7, 9), kernel_size=(2, 3)).rank == 7
    m = NMF3D((1, 3, 64, 64, 100), 8, (5, 5, 20))
    assert tuple(m.W.shape) == (3, 8, 5, 5, 20) and tuple(m.H.shape) == (1, 8, 60, 60, 81) and m.out_channels == 3
    assert NMF3D((1, 2, 5, 7, 9), kernel_size=2).rank == 7
    m = NMF2D(W=torch.rand(4, 3, 2, 2), H=torch.rand(1, 3, 5, 6), trainable_W=False)
    assert m.rank == 3 and not m.W.requires_grad and m.H.requires_grad
    with pytest.raises(_capi.NmfmuError):          # no CPU fallback
        m()


def test_plca_constructor_rules():
    """plca.py:60-140, 352-361: arguments are normalised, Z defaults to uniform, rank inference and its asserts."""
    from torchnmf_amd.plca import PLCA
    m = PLCA((20, 30), 5)
    assert tuple(m.W.shape) == (30, 5) and tuple(m.H.shape) == (20, 5) and tuple(m.Z.shape) == (5,)
    assert torch.allclose(m.W.sum(0), torch.ones(5)) and torch.allclose(m.H.sum(0), torch.ones(5))
    assert torch.allclose(m.Z.data, torch.full((5,), 0.2)) and m.rank == 5 and m.out_channels == 30
    g = load_golden('g10_plca')
    m = PLCA(W=t(g['W0']), H=t(g['H0']), Z=t(g['Z0']), trainable_Z=False)
    assert rel_err(m.W.data, g['W_init']) < 1e-6 and rel_err(m.H.data, g['H_init']) < 1e-6 and rel_err(m.Z.data, g['Z_init']) < 1e-6
    assert not m.Z.requires_grad and m.W.requires_grad
    assert PLCA((7, 9)).rank == 9
    with pytest.raises(AssertionError):
        PLCA(W=torch.rand(9, 3), H=torch.rand(7, 4))
    with pytest.raises(AssertionError):
        PLCA(W=-torch.rand(9, 3), H=torch.rand(7, 3))
    with pytest.raises(AssertionError):
        PLCA(W=torch.rand(9, 3), H=torch.rand(7, 3), Z=torch.rand(3, 1))
    with pytest.raises(_capi.NmfmuError):          # no CPU fallback
        m()


def test_siplca_constructors():
    """plca.py:430-445, 505-520, 585-600: the docstring shapes and the rank default."""
    from torchnmf_amd.plca import SIPLCA, SIPLCA2, SIPLCA3
    m = SIPLCA((1, 33, 50), 16, 3)
    assert tuple(m.W.shape) == (33, 16, 3) and tuple(m.H.shape) == (1, 16, 48) and tuple(m.Z.shape) == (16,)
    assert torch.allclose(m.W.sum((0, 2)), torch.ones(16)) and torch.allclose(m.H.sum((0, 2)), torch.ones(16))
    m = SIPLCA2((1, 3, 14, 12), 2, (2, 3))
    assert tuple(m.W.shape) == (3, 2, 2, 3) and tuple(m.H.shape) == (1, 2, 13, 10) and m.kernel_size == (2, 3)
    m = SIPLCA3((1, 2, 6, 7, 8), kernel_size=2)
    assert tuple(m.W.shape) == (2, 7, 2, 2, 2) and m.rank == 7 and tuple(m.H.shape) == (1, 7, 5, 6, 7)


# ---- trainer.BetaMu host logic on the stand-in backend --------------------------------------------------------
@pytest.mark.parametrize('case', ['b1_plain_both', 'b0.5_pen_both', 'b2_plain_W', 'b3_plain_H', 'b-1_pen_both'])
def test_betamu_step_matches_reference(cpu_engine, case):
    from torchnmf_amd.trainer import BetaMu
    g = load_golden('g7_betamu')
    b, pen, which = case.split('_')
    l1, l2, ortho = {'plain': (0, 0, 0), 'pen': (1e-3, 1e-3, 1e-2)}[pen]
    m = NMF(W=t(g['W0']), H=t(g['H0']))
    params = list(m.parameters()) if which == 'both' else [getattr(m, which)]
    trainer = BetaMu(params, float(b[1:]), l1, l2, ortho)
    V = t(g['V'])
    calls = []

    def closure():
        trainer.zero_grad()
        calls.append(1)
        return V, m            # deferred form: the layer itself (no reconstruction on the host in this test)
    for it in range(1, 6):
        trainer.step(closure)
        if it in (1, 5):
            assert rel_err(m.W.data, g[f'{case}_W{it}']) < 5e-6 and rel_err(m.H.data, g[f'{case}_H{it}']) < 5e-6
        if it == 1:
            last = 'H' if which in ('both', 'H') else 'W'
            assert rel_err(getattr(m, last).grad, g[f'{case}_grad{last}1']) < 5e-6
    assert len(calls) == 5 * len(params)                       # closure re-evaluated per parameter (trainer.py:72)
    assert m.W.requires_grad and m.H.requires_grad             # flags restored (trainer.py:117-119)


def test_betamu_argument_checks_and_unsupported_graphs(cpu_engine):
    from torchnmf_amd.trainer import BetaMu
    m = NMF((12, 9), 3)
    for kw in ({'l1_reg': -1}, {'l2_reg': -1}, {'orthogonal': -1}):
        with pytest.raises(ValueError):
            BetaMu(m.parameters(), **kw)
    trainer = BetaMu(m.parameters())
    V = torch.rand(12, 9)
    with pytest.raises(NotImplementedError):                   # not the direct output of one layer
        trainer.step(lambda: (V, torch.rand(12, 9)))
    from torchnmf_amd.nmf import NMFD
    d = NMFD((1, 9, 12), 3, 2)
    with pytest.raises(Exception, match='no CPU fallback|MI355X'):      # (round 6) one conv layer is a supported graph: it gets
        BetaMu(d.parameters()).step(lambda: (torch.rand(1, 9, 12), d))   # as far as the device check of its engine
    # a parameter that does not feed the prediction is skipped, frozen ones are left alone
    other = torch.nn.Parameter(torch.rand(3, 3))
    m2 = NMF(W=torch.rand(9, 3), H=torch.rand(12, 3), trainable_H=False)   # (a shape spec ignores trainable_*)
    H_before = m2.H.data.clone()
    tr = BetaMu(list(m2.parameters()) + [other])
    tr.step(lambda: (V, m2))
    assert torch.equal(m2.H.data, H_before) and other.grad is None and not m2.H.requires_grad
    # external in-place edits of a factor between steps are picked up (images refreshed)
    with torch.no_grad():
        m2.W.mul_(2.0)
    W_edit = m2.W.data.clone()
    tr.step(lambda: (V, m2))
    from oracle import mu_oracle as O
    W_ref, _, _ = O.betamu_step(V, W_edit, H_before, 1.0, params=('W',))
    assert rel_err(m2.W.data, W_ref) < 5e-6


def test_f16_admission_test_of_auto_precision():
    """'auto' takes the fp16 mode only for targets fp16 holds EXACTLY and data inside its range:
    the host-side test, on CPU tensors (it is plain torch code)."""
    from torchnmf_amd.engine import DenseMU
    g = torch.Generator().manual_seed(0)
    V = torch.rand(300, 500, generator=g)
    W, H = torch.rand(500, 8, generator=g) + 0.1, torch.rand(300, 8, generator=g) + 0.1
    assert not DenseMU.f16_in_range(V, W, H)                          # plain floats: fp16 would round them
    assert DenseMU.f16_in_range(V.bfloat16().float(), W, H)           # bf16-sourced data is exact in fp16
    assert DenseMU.f16_in_range(V.half().float(), W, H)
    assert DenseMU.f16_in_range(torch.randint(0, 2048, (300, 500), generator=g).float(), W, H)   # counts below 2^11
    assert not DenseMU.f16_in_range(torch.randint(0, 4096, (300, 500), generator=g).float() + 2049, W, H)
    assert not DenseMU.f16_in_range(V.half().float() * 2.0 ** 17, W, H)   # exact, but beyond the range gate
    assert not DenseMU.f16_in_range(V.half().float() * 2.0 ** -20, W, H)  # mostly subnormal
    assert not DenseMU.f16_in_range(V.half().float(), W * 1e5, H)
    Vb = V.half().float()
    Vb[:, 250:] = V[:, 250:]                                              # inexact values in a later row chunk's columns
    assert not DenseMU.f16_in_range(Vb, W, H)


def test_auto_precision_policy_on_the_standin_backend(cpu_engine, monkeypatch):
    """The decision tree of 'auto' without a GPU (the stand-in backend supports every single-plane rank and split bf16 up
    to rank 128, like the library): fp16 needs both dimensions >= F16_MIN_DIM and an admissible target; otherwise split
    bf16; above rank 128 an error from the fused engine -- never plain bf16."""
    from torchnmf_amd import _capi
    from torchnmf_amd.engine import DenseMU
    g = torch.Generator().manual_seed(1)
    old = DenseMU.F16_MIN_DIM
    DenseMU.F16_MIN_DIM = 64
    try:
        def pick(N, C, R, exact=True, allow=True, beta=1.0):
            V = torch.rand(N, C, generator=g)
            V = V.half().float() if exact else V
            return DenseMU(V, torch.rand(C, R, generator=g) + 0.1, torch.rand(N, R, generator=g) + 0.1, beta,
                           precision='auto', allow_f16=allow).precision_name
        assert pick(64, 80, 8) == 'f16'
        assert pick(64, 80, 8, exact=False) == 'f16r'       # round 6: an inexact target keeps its top 24 bits (3 bytes per element)
        assert pick(64, 80, 8, exact=False, beta=2.0) == 'f16x'   # beta == 2: the target is an MFMA operand, it stays fp32
        assert pick(64, 80, 8, allow=False) == 'bf16x3'
        assert pick(63, 80, 8) == 'bf16x3'
        assert pick(64, 80, 200) == 'f16'
        assert pick(64, 80, 200, exact=False) == 'f16r'
        monkeypatch.setenv('TORCHNMF_AMD_AUTO_F16R', '0')
        assert pick(64, 80, 8, exact=False) == 'f16x' and pick(64, 80, 200, exact=False) == 'f16x'
        monkeypatch.delenv('TORCHNMF_AMD_AUTO_F16R')
        with pytest.raises(NotImplementedError):
            pick(63, 80, 200)
        monkeypatch.setenv('TORCHNMF_AMD_AUTO_F16X', '0')
        assert pick(64, 80, 8, exact=False) == 'bf16x3'
        with pytest.raises(NotImplementedError):
            pick(64, 80, 200, exact=False)
        monkeypatch.setenv('TORCHNMF_AMD_AUTO_F16', '0')
        assert pick(64, 80, 8) == 'bf16x3'
    finally:
        DenseMU.F16_MIN_DIM = old
    assert _capi.PRECISIONS['f16'] == _capi.PREC_F16 == 2 and _capi.PRECISIONS['f16x'] == _capi.PREC_F16X == 3


def test_bench_block_timing_rules(monkeypatch):
    """bench.py's timed_blocks: at least --repeats blocks of exactly K steps, more until two consecutive blocks agree within
    2 % (or --max-repeats); the median of the settled tail is reported.  Driven with a fake clock, no GPU."""
    import importlib
    import sys as _sys
    monkeypatch.setattr(torch.cuda, 'synchronize', lambda *a, **k: None)
    _sys.modules.pop('bench', None)
    bench = importlib.import_module('bench')
    calls = {'n': 0}
    durations = iter([1.30, 1.20, 1.10, 1.00, 0.99, 0.985, 0.98, 0.98, 0.98])   # seconds per block: still falling at first
    now = {'t': 0.0}
    pending = {'d': None}

    def perf_counter():
        return now['t']

    def run():
        calls['n'] += 1
        if calls['n'] % 10 == 1:                     # first step of a block: advance the clock by that block's duration
            pending['d'] = next(durations)
            now['t'] += pending['d']
    monkeypatch.setattr(bench.time, 'perf_counter', perf_counter)
    ms, blocks = bench.timed_blocks(run, 10, 3, 40)
    assert calls['n'] == 10 * len(blocks)            # exactly K steps per block
    assert len(blocks) == 5                          # 1.30, 1.20, 1.10 (3 = --repeats), 1.00, 0.99: the last two agree
    assert ms == pytest.approx(sorted(blocks[-3:])[1]) and blocks[0] == pytest.approx(130.0)
    calls['n'] = 0
    durations = iter([1.0] * 50)
    ms, blocks = bench.timed_blocks(run, 10, 5, 40)
    assert len(blocks) == 5 and ms == pytest.approx(100.0)
    calls['n'] = 0
    durations = iter([2.0 ** -i for i in range(50)])  # never settles: stops at --max-repeats
    ms, blocks = bench.timed_blocks(run, 10, 2, 6)
    assert len(blocks) == 6


@pytest.mark.parametrize('unit', ['nmfmu_inst_r128', 'nmfmu_inst_r256', 'nmfmu_inst_pp', 'nmfmu_inst_sp', 'nmfmu_inst_sp2a'])
def test_fused_kernels_do_not_spill_to_scratch(tmp_path, unit):
    """Guard: the fused kernels must keep their accumulators in registers.  A runtime-indexed register array silently
    moves to scratch memory AND is kept up to date from inside the main loop: the padded-rank-256 kernels ran 3x slower
    that way until their epilogue loops became compile-time (static_for).  Checked with the library's own flags for
    every instantiation: no scratch instruction inside any loop, and at most a handful of spilled words in the
    prologue / epilogue of the instances that fill the whole 512-register file (padded rank 256, two accumulator sets)."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, 'pytorch-nmf_amd', 'csrc', unit + '.hip')
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-slp-vectorize', '-Wno-inline-asm']
    r = subprocess.run([hipcc] + flags + ['-Rpass-analysis=kernel-resource-usage', '-S', '--cuda-device-only', src, '-o',
                                          str(tmp_path / 'x.s')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = r.stderr.split('Function Name: ')[1:]
    assert len(blocks) >= (12 if 'sp' not in unit else 1)
    spilled = set()
    for b in blocks:
        name = b.split('\n')[0].strip()
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', b)
        assert m, name
        if int(m.group(1)) > 0:
            assert unit == 'nmfmu_inst_r256' and int(m.group(1)) <= 64, (name, m.group(1))
            spilled.add(name)
    # every scratch access of the spilling instances sits outside the loops (LLVM tags loop blocks in the label comment)
    asm = open(tmp_path / 'x.s').read()
    for fn in re.split(r'\n(?=_ZN5nmfmu\w+:)', asm)[1:]:
        in_loop = False
        for line in fn.split('\n'):
            if re.match(r'\.LBB\d+_\d+:', line):
                in_loop = 'Loop' in line
            elif 'scratch_' in line:
                assert not in_loop, (fn.split(':')[0], line)
    if 'sp' in unit:
        # the software-pipelined kernels (round 6): every instruction of their tile loops is an asm statement; what hipcc adds by
        # itself there may be scalar address arithmetic and its own wait-state padding, NEVER a vector move / accumulator move /
        # wait / memory access (a copy of a register an asm load is still landing in is silent corruption), and never M0
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import asm_audit
        res = asm_audit.audit(str(tmp_path / 'x.s'), 'sp', verbose=False)
        assert res
        for name, (asm_ops, comp, comp_lines) in res.items():
            assert asm_ops['v_mfma_f32_32x32x16_f16'] >= 192, name
            assert set(comp) <= {'s_nop', 's_add_u32', 's_addc_u32', 's_add_i32', 's_lshl_b64', 's_min_i32', 's_ashr_i32', 's_mov_b64',
                                 's_mov_b32', 's_cmp_lt_i32', 's_cbranch_scc1', 's_mul_i32', 's_mul_hi_u32', 's_sub_i32', 's_lshl_b32',
                                 's_and_b32', 's_or_b32', 's_cselect_b32', 's_cmp_eq_u32', 's_branch'}, (name, sorted(comp))
            assert not any('m0' in l for l in comp_lines), name


def test_tail_round_split_selection():
    """Host rule of the NMFD H-numerator GEMM's tail-round split (nmfd_engine.tail_round_split): chosen only when the tiles
    make whole rounds of the chip plus at most a quarter round of whole tile rows; every contraction part non-empty."""
    from torchnmf_amd.nmfd_engine import tail_round_split
    # configs[3]: R T = 3200 -> 25 tile rows, B L = 8192 -> 64 tile columns, 512 slots, 1025 channels -> 17 k-tiles
    assert tail_round_split(25, 64, 512, 17) == (1, 6)          # min(512 / 64, 17 // 2, 8) = 8 -> parts of 3 k-tiles -> 6 parts
    assert tail_round_split(25, 64, 512, 16) == (1, 8)
    # exactly whole rounds, less than one round, a remainder that is not whole rows or above a quarter round: none
    assert tail_round_split(24, 64, 512, 17) == (0, 1)
    assert tail_round_split(7, 64, 512, 17) == (0, 1)
    assert tail_round_split(25, 60, 512, 17) == (0, 1)          # 1500 % 512 = 476 > 128
    assert tail_round_split(27, 64, 512, 17) == (0, 1)          # 3 rows = 192 > 128
    assert tail_round_split(26, 64, 512, 17) == (2, 4)          # 128 tiles left: 512 / 128 = 4 parts of 5, 5, 5, 2
    assert tail_round_split(25, 64, 512, 7) == (0, 1)           # short contractions are not worth splitting
    # forced (tests): normalised so that no part is empty -- 17 k-tiles in 8 parts of 3 would leave parts 6, 7 empty
    assert tail_round_split(25, 64, 512, 17, '4,8') == (4, 6)
    assert tail_round_split(5, 5, 512, 9, '1,3') == (1, 3)
    assert tail_round_split(5, 5, 512, 10, '2,2') == (2, 2)
    for kt in range(2, 40):
        for want_split in range(2, min(kt, 12) + 1):
            rows, split = tail_round_split(30, 10, 512, kt, f'3,{want_split}')
            per = -(-kt // split)
            assert rows == 3 and 2 <= split <= want_split and (split - 1) * per < kt <= split * per


@pytest.mark.parametrize('B,R,lhs,ts', [(2, 3, (5, 9), (3, 8)), (1, 2, (3, 4, 17), (2, 2, 8)), (2, 2, (17,), (8,)),
                                         (1, 2, (6, 25), (4, 16))])
def test_window_tables_with_several_shift_axes(B, R, lhs, ts):
    """nmfmu_convnd_koff (host code) + the chunk-index rule of include/nmfmu.h (nmfmu_convnd_tables): a numpy model of the
    tables, gathered as the GEMM lanes gather them, must reproduce the unfolded operands Hu[(b,l)][(r,t)] = H[b][r][l - t]
    and its transpose (nmf.py:857-860 / 937-940: conv2d / conv3d with the flipped kernel)."""
    import ctypes as C
    import itertools
    lib = _capi.load()
    nd = len(lhs)
    ls = tuple(lh + t - 1 for lh, t in zip(lhs, ts))
    jj = tuple(lh + 2 * t - 2 for lh, t in zip(lhs, ts))
    JJ = int(np.prod(jj))
    S = [int(np.prod(jj[d + 1:])) for d in range(nd)]
    rng = np.random.default_rng(5)
    H = rng.random((B, R) + lhs).astype(np.float32)
    arr = (C.c_int32 * nd)
    nb = lib.nmfmu_convnd_table_bytes(B, R, nd, arr(*lhs), arr(*ts))
    assert nb == 16 * (1 + B * R * JJ)
    # numpy model of the two tables (8 values per chunk)
    rev, fwd = np.zeros((nb // 16, 8), np.float32), np.zeros((nb // 16, 8), np.float32)
    for b, r in itertools.product(range(B), range(R)):
        for p in itertools.product(*[range(x) for x in jj]):
            j = [pd - (t - 1) for pd, t in zip(p, ts)]
            if any(not 0 <= jd < lh for jd, lh in zip(j[:-1], lhs[:-1])):
                continue
            line = H[(b, r) + tuple(j[:-1])]
            i = 1 + (b * R + r) * JJ + sum(pd * sd for pd, sd in zip(p, S))
            for e in range(8):
                if 0 <= j[-1] - e < lhs[-1]:
                    rev[i, e] = line[j[-1] - e]
                if 0 <= j[-1] + e < lhs[-1]:
                    fwd[i, e] = line[j[-1] + e]
    L, T = int(np.prod(ls)), int(np.prod(ts))
    rp_pad, bl_pad = -(-R * T // 128) * 128, -(-B * L // 128) * 128

    def h_at(b, r, l, t):
        j = tuple(ld - td for ld, td in zip(l, t))
        return H[(b, r) + j] if all(0 <= jd < lh for jd, lh in zip(j, lhs)) else 0.0

    # rows (b, l), k = (r, t)
    koff = np.zeros(rp_pad // 8 + 8, np.int32)
    assert lib.nmfmu_convnd_koff(_capi.OPS_B_HU, B, R, nd, arr(*lhs), arr(*ts), rp_pad, koff.ctypes.data) == 0
    assert all(koff[R * T // 8:] == np.iinfo(np.int32).min)
    for b in range(B):
        for l in itertools.product(*[range(x) for x in ls]):
            row_term = b * R * JJ + sum((ld + t - 1) * sd for ld, t, sd in zip(l, ts, S))
            for kc in range(R * T // 8):
                r, tf = divmod(8 * kc, T)
                t = np.unravel_index(tf, ts)
                want = [h_at(b, r, l, tuple(t[:-1]) + (t[-1] + e,)) for e in range(8)]
                assert np.array_equal(rev[row_term + koff[kc]], np.array(want, np.float32))
    # rows (r, t), k = (b, l)
    koff = np.zeros(bl_pad // 8 + 8, np.int32)
    assert lib.nmfmu_convnd_koff(_capi.OPS_B_HUT, B, R, nd, arr(*lhs), arr(*ts), bl_pad, koff.ctypes.data) == 0
    assert all(koff[B * L // 8:] == np.iinfo(np.int32).min)
    for r in range(R):
        for t in itertools.product(*[range(x) for x in ts]):
            row_term = r * JJ + sum((td_max - 1 - td) * sd for td_max, td, sd in zip(ts, t, S))
            for kc in range(B * L // 8):
                b, lf = divmod(8 * kc, L)
                l = np.unravel_index(lf, ls)
                want = [h_at(b, r, tuple(l[:-1]) + (l[-1] + e,), t) for e in range(8)]
                assert np.array_equal(fwd[row_term + koff[kc]], np.array(want, np.float32))
    # argument checks: last-axis alignment
    bad = np.zeros(32, np.int32)
    assert lib.nmfmu_convnd_koff(_capi.OPS_B_HU, 1, 1, 2, (C.c_int32 * 2)(4, 5), (C.c_int32 * 2)(2, 4), 128, bad.ctypes.data) == _capi.ERR_ARG
    assert lib.nmfmu_convnd_koff(_capi.OPS_PLANES, 1, 1, 1, (C.c_int32 * 1)(9), (C.c_int32 * 1)(8), 128, bad.ctypes.data) == _capi.ERR_ARG


def test_convolutive_host_rules_of_round_4():
    """nmfd_engine.w_contraction_split / h_tap_fold: the split of the W-numerator GEMM's contraction and the tap fold of the
    window-operand GEMM (pure host rules; the kernels take whatever they say)."""
    from torchnmf_amd.nmfd_engine import h_tap_fold, w_contraction_split
    # NMF2D bench shape: 8 tiles, 2 048 k-tiles, 512 slots -> 64 parts of 32 k-tiles
    assert w_contraction_split(8, 2048, 512) == 64
    assert w_contraction_split(225, 128, 512) == 2            # configs[3]-like: two parts fill the second slot
    assert w_contraction_split(600, 4096, 512) == 1           # more tiles than slots: no split
    assert w_contraction_split(1, 20, 512) == 2               # at least eight k-tiles per part
    assert w_contraction_split(4, 7 * 11, 512) in (1, 7)      # must divide the k-tiles
    for tiles, kt in ((1, 2), (3, 1000), (17, 4096), (500, 64)):
        s = w_contraction_split(tiles, kt, 512)
        assert 1 <= s <= 64 and kt % s == 0 and (s == 1 or (kt // s >= 8 and tiles * s <= 512))
    assert h_tap_fold(8, 16) == 4 and h_tap_fold(5, 8) == 4 and h_tap_fold(7, 30) == 2 and h_tap_fold(16, 16) == 2
    assert h_tap_fold(8, 3) == 1 and h_tap_fold(17, 16) == 1 and h_tap_fold(40, 8) == 1 and h_tap_fold(9, 6) == 2


@pytest.mark.parametrize('B,Cc,R,lhs,ts,F', [(2, 5, 3, (4, 7), (2, 4), 4), (1, 3, 2, (6,), (6,), 2), (2, 4, 5, (3, 2, 5), (2, 2, 3), 1),
                                              (1, 70, 8, (3, 9), (2, 8), 4)])
def test_window_operand_gemm_contract(B, Cc, R, lhs, ts, F):
    """The contract of NMFMU_OPS_A_WIN + win_fold as include/nmfmu.h states it, in numpy, against the oracle's conv
    backward pass wrt H (nmf.py:857-860 through autograd; oracle._convnd_grad_h): rows of A are shifted rows of the ratio
    plane P[(b,l)][c]; k = (t_outer, q, ck, c'); B = the nmfmu_conv_pack_wk layout; F taps of the last axis go to F
    columns and the consumer adds out[(.., j + d)][r F + d] over d."""
    from oracle import mu_oracle as O
    nd = len(lhs)
    ls = tuple(lh + t - 1 for lh, t in zip(lhs, ts))
    rng = np.random.default_rng(11)
    G = rng.random((B, Cc) + ls).astype(np.float64)                  # the ratio Gn[b][c][l]
    W = rng.random((Cc, R) + ts).astype(np.float64)
    H = np.zeros((B, R) + lhs)
    want = O._convnd_grad_h(torch.from_numpy(G), torch.from_numpy(W), torch.from_numpy(H)).numpy()
    L, T, CK = int(np.prod(ls)), int(np.prod(ts)), -(-Cc // 64)
    P = np.zeros((B * L, CK * 64))                                    # [(b,l)][c], channel padding zero
    P[:, :Cc] = np.moveaxis(G, 1, -1).reshape(B * L, Cc)
    TQ, T_outer = ts[-1] // F, T // ts[-1]
    # B operand: Wk[r F + d][((to TQ + q) CK + ck) 64 + c'] = W[c][r][to T_last + F q + d]
    Wf = W.reshape(Cc, R, T)
    Wk = np.zeros((R * F, T_outer * TQ * CK * 64))
    for r in range(R):
        for d in range(F):
            for to in range(T_outer):
                for q in range(TQ):
                    k0 = (to * TQ + q) * CK * 64
                    Wk[r * F + d, k0:k0 + Cc] = Wf[:, r, to * ts[-1] + F * q + d]
    # A operand rows (b, jo, j'), j' < lh_last + F - 1; k-tile (to, q): row (b, jo + to, j' + F q) of P
    lw = lhs[-1] + F - 1
    outer = lhs[:-1]
    rows = [(b,) + jo + (jp,) for b in range(B) for jo in np.ndindex(*outer) for jp in range(lw)]
    out = np.zeros((len(rows), R * F))
    for i, (b, *jo, jp) in enumerate(rows):
        a_row = []
        for to in np.ndindex(*ts[:-1]):
            for q in range(TQ):
                l = tuple(j + t for j, t in zip(jo, to)) + (jp + F * q,)
                a_row.append(P[b * L + int(np.ravel_multi_index(l, ls))])
        out[i] = Wk @ np.concatenate(a_row)
    # the consumer (nmfmu_conv_apply_h_rows / nmfmu_conv_rows_fold)
    out = out.reshape((B,) + outer + (lw, R * F))
    got = np.zeros((B, R) + lhs)
    for r in range(R):
        for d in range(F):
            got[:, r] += out[..., d:d + lhs[-1], r * F + d]
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)


def test_plca_tensor_alpha_contract_matches_the_reference():
    """plca.py:197-199 types the Dirichlet hyper-parameters Union[float, Tensor]; its EM loop evaluates ``if alpha != 1``
    (plca.py:257, 271, 285) and adds ``alpha - 1`` in place, so a tensor works when it has ONE element and no more
    dimensions than the factor, and raises RuntimeError otherwise (messages recorded from the reference in g13).  The check
    runs before anything touches a device."""
    from torchnmf_amd.plca import PLCA, _scalar_alpha
    g = load_golden('g13_plca_tensor_alpha')
    m = PLCA((6, 5), 3)
    V = torch.rand(6, 5)
    with pytest.raises(RuntimeError, match='Boolean value of Tensor with more than one value is ambiguous'):
        m.fit(V, W_alpha=torch.full((5, 3), 1.03))
    with pytest.raises(RuntimeError, match='broadcast shape'):
        m.fit(V, Z_alpha=torch.tensor([[1.02]]))
    assert str(g['multi_error']).split(': ', 1)[1] in 'Boolean value of Tensor with more than one value is ambiguous'
    assert _scalar_alpha(torch.tensor([0.98]), 'H_alpha', m.H) == pytest.approx(0.98, rel=1e-7)
    assert _scalar_alpha(torch.tensor([[1.5]]), 'W_alpha', m.W) == 1.5 and _scalar_alpha(2, 'Z_alpha', m.Z) == 2.0
    with pytest.raises(Exception, match='no CPU fallback|MI355X'):      # admitted tensors get as far as the device check
        m.fit(V, W_alpha=torch.tensor(1.03), H_alpha=torch.tensor([0.98]))


def test_plca_rejects_f16x_with_a_clear_message():
    """'f16x' passed PLCA's constructor checks and failed inside the first EM step with a misleading rank
    message.  The precision is validated where the engine is built -- visible here without a GPU through the message."""
    import inspect
    from torchnmf_amd import plca
    src = inspect.getsource(plca._PlcaEM.__init__)
    assert "precision in ('f16x', 'f16')" in src and 'NotImplementedError' in src
