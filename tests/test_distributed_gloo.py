"""world_size-2 CPU tests (gloo) of the column-sharded MU path.

Each rank owns V[:, Cg] and W[Cg]; H is replicated.  The W half-step is local, the H half-step all-reduces ONE
packed buffer [numerator | denominator]; validation flags and the loss are all-reduced too.  Compute is done by
the oracle-backed stand-in backend (tests/cpu_backend.py); what is under test is torchnmf_amd's host logic,
which is the same code that runs over RCCL on the GPUs.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, rel_err


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, beta, alpha, out_dir):
    for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cpu_backend import OracleBackend
        from oracle import mu_oracle as O
        from torchnmf_amd import engine
        from torchnmf_amd import nmf as anmf
        from torchnmf_amd.nmf import NMF
        engine.DEFAULT_BACKEND_FACTORY = OracleBackend
        anmf._require_device = lambda t_, what: None
        torch.set_num_threads(1)
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g1_nmf_small.npz'))
        V = torch.from_numpy(g['V']) + (1e-3 if beta <= 0 else 0.0)
        W0, H0 = torch.from_numpy(g['W0']), torch.from_numpy(g['H0'])
        s, e = O.shard_bounds(V.shape[1], world)[rank]
        m = NMF(W=W0[s:e].clone(), H=H0.clone())
        n = m.fit(V[:, s:e].contiguous(), beta, 1e-4, 60, alpha=alpha, l1_ratio=0.5, process_group=dist.group.WORLD)
        torch.save({'W': m.W.data, 'H': m.H.data, 'n': n, 's': s, 'e': e}, os.path.join(out_dir, f'r{rank}.pt'))
        # negative entries on ONE rank must fail the assertion on EVERY rank (flags are all-reduced)
        Vbad = V[:, s:e].clone()
        if rank == 1:
            Vbad[0, 0] = -1.0
        try:
            NMF(W=W0[s:e].clone(), H=H0.clone()).fit(Vbad, beta, 1e-4, 2, process_group=dist.group.WORLD)
            raised = False
        except AssertionError:
            raised = True
        torch.save(raised, os.path.join(out_dir, f'bad{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('beta,alpha', [(1, 0.0), (2, 0.1), (0.5, 0.0)])
def test_column_sharded_fit_world2(tmp_path, beta, alpha):
    from oracle import mu_oracle as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), beta, alpha, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f'r{r}.pt') for r in range(world)]
    g = load_golden('g1_nmf_small')
    V = torch.from_numpy(g['V'])
    W0, H0 = torch.from_numpy(g['W0']), torch.from_numpy(g['H0'])
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, beta, 1e-4, 60, alpha, 0.5)
    assert all(p['n'] == nr for p in parts)                      # same stop decision on every rank
    W = torch.cat([p['W'] for p in parts])
    assert rel_err(W, Wr) < 1e-5
    for p in parts:                                              # H identical (replicated) and right
        assert rel_err(p['H'], Hr) < 1e-5
    assert torch.equal(parts[0]['H'], parts[1]['H'])
    assert all(torch.load(tmp_path / f'bad{r}.pt') for r in range(world))


def _worker_rows(rank, world, port, beta, overlap, out_dir, split=None):
    for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if overlap == 'default':        # nothing chosen: the engine's own default must be north_star's single all-reduce
        os.environ.pop('TORCHNMF_AMD_AR_OVERLAP', None)
    else:
        os.environ['TORCHNMF_AMD_AR_OVERLAP'] = overlap
    if split is not None:
        os.environ['TORCHNMF_AMD_AR_SPLIT'] = split
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cpu_backend import OracleBackend
        from oracle import mu_oracle as O
        from torchnmf_amd import engine
        from torchnmf_amd import nmf as anmf
        from torchnmf_amd.nmf import NMF
        engine.DEFAULT_BACKEND_FACTORY = OracleBackend
        anmf._require_device = lambda t_, what: None
        torch.set_num_threads(1)
        V, W0, H0 = _tall_problem(beta)
        s, e = O.shard_bounds(V.shape[1], world)[rank]
        m = NMF(W=W0[s:e].clone(), H=H0.clone())
        seen = {'sum_per_h_step': set()}
        orig = engine.DenseMU.h_step
        orig_ar = dist.all_reduce
        count = {'n': 0}

        def counting_all_reduce(tensor, op=dist.ReduceOp.SUM, *a_, **k_):
            if op == dist.ReduceOp.SUM:
                count['n'] += 1
            return orig_ar(tensor, op, *a_, **k_)
        dist.all_reduce = counting_all_reduce

        def spy(self):
            seen['rows'] = None if self._h_rows is None else [(v.r0, v.owner.rows, v.owner.rows_pad) for v in self._h_rows]
            before = count['n']
            out = orig(self)
            seen['sum_per_h_step'].add(count['n'] - before)     # SUM collectives of this half-step
            return out
        engine.DenseMU.h_step = spy
        n = m.fit(V[:, s:e].contiguous(), beta, 1e-4, 25, alpha=0.05, l1_ratio=0.5, process_group=dist.group.WORLD)
        torch.save({'W': m.W.data, 'H': m.H.data, 'n': n, 'rows': seen.get('rows'), 'sums': sorted(seen['sum_per_h_step'])},
                   os.path.join(out_dir, f'r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def _tall_problem(beta):
    g = torch.Generator().manual_seed(77)
    N, Cc, R = 700, 90, 5          # 700 rows pad to 768: row halves [0, 256) and [256, 768) with 444 valid rows
    V = torch.rand(N, Cc, generator=g) + (1e-3 if beta <= 0 else 0.0)
    return V, torch.randn(Cc, R, generator=g).abs() + 1e-3, torch.randn(N, R, generator=g).abs() + 1e-3


@pytest.mark.parametrize('beta', [1, 2])
@pytest.mark.parametrize('overlap', ['1', '0', 'default'])
def test_sharded_h_step_in_row_halves_world2(tmp_path, beta, overlap):
    """The overlapped form of the sharded H half-step: two row halves, the first half's numerators all-reduced
    (async) while the second half is computed, the denominators with the second; one apply.  Must give what the
    single-launch form gives and what the unsharded oracle gives."""
    from oracle import mu_oracle as O
    world = 2
    mp.spawn(_worker_rows, args=(world, _free_port(), beta, overlap, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f'r{r}.pt') for r in range(world)]
    V, W0, H0 = _tall_problem(beta)
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, beta, 1e-4, 25, 0.05, 0.5)
    for p in parts:
        assert p['rows'] == ([(0, 256, 256), (256, 444, 512)] if overlap == '1' else None)
        # north_star's form (the default since round 5; TORCHNMF_AMD_AR_OVERLAP=0 / fit(..., allreduce='single')): exactly ONE SUM
        # all-reduce of the packed [numerator | denominator] buffer per iteration; the overlapped form sends the two row halves
        # separately
        assert p['sums'] == ([2] if overlap == '1' else [1])
        assert p['n'] == nr and rel_err(p['H'], Hr) < 1e-5
    assert rel_err(torch.cat([p['W'] for p in parts]), Wr) < 1e-5
    assert torch.equal(parts[0]['H'], parts[1]['H'])


@pytest.mark.parametrize('beta', [1, 2])
def test_sharded_fit_world4_uneven_shards_and_moved_split(tmp_path, beta):
    """Four ranks with uneven column shards (90 columns: 23 / 23 / 22 / 22), the row split of the overlapped H half-step
    moved with TORCHNMF_AMD_AR_SPLIT (two thirds of the row blocks in the first part): same factors on every rank, the
    unsharded oracle's result, the same stop decision."""
    from oracle import mu_oracle as O
    world = 4
    mp.spawn(_worker_rows, args=(world, _free_port(), beta, '1', str(tmp_path), '0.67'), nprocs=world, join=True)
    parts = [torch.load(tmp_path / f'r{r}.pt') for r in range(world)]
    V, W0, H0 = _tall_problem(beta)
    assert [e - s for s, e in O.shard_bounds(V.shape[1], world)] == [23, 23, 22, 22]
    Wr, Hr, nr, _, _ = O.fit(V, W0, H0, beta, 1e-4, 25, 0.05, 0.5)
    for p in parts:
        assert p['rows'] == [(0, 512, 512), (512, 188, 256)]
        assert p['n'] == nr and rel_err(p['H'], Hr) < 1e-5
        assert torch.equal(p['H'], parts[0]['H'])
    assert rel_err(torch.cat([p['W'] for p in parts]), Wr) < 1e-5


def _worker_gate(rank, world, port, cols, inexact_rank, out_dir):
    for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cpu_backend import OracleBackend
        from oracle import mu_oracle as O
        from torchnmf_amd import engine
        engine.DEFAULT_BACKEND_FACTORY = OracleBackend
        engine.DenseMU.F16_MIN_DIM = 64
        torch.set_num_threads(1)
        g = torch.Generator().manual_seed(5)
        N, R = 64, 4
        V = torch.rand(N, cols, generator=g).half().float()
        W0, H0 = torch.rand(cols, R, generator=g) + 0.1, torch.rand(N, R, generator=g) + 0.1
        s, e = O.shard_bounds(cols, world)[rank]
        Vs = V[:, s:e].contiguous()
        if rank == inexact_rank:
            Vs[0, 0] += 2.0 ** -15                    # one value fp16 does not hold, on one rank only
        eng = engine.DenseMU(Vs, W0[s:e].clone(), H0.clone(), 1.0, precision='auto', group=dist.group.WORLD, allow_f16=True)
        eng.w_step()
        eng.h_step()                                   # would hang / mis-pair if the ranks disagreed about the gate's collective
        torch.save((eng.precision_name, e - s), os.path.join(out_dir, f'g{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('cols,inexact_rank,want', [(128, -1, 'f16'), (128, 1, 'f16r'), (127, -1, 'bf16x3'), (127, 0, 'bf16x3')])
def test_auto_precision_gate_is_rank_invariant(tmp_path, cols, inexact_rank, want):
    """the admission test of the fp16 modes looks at the LOCAL shard (size, exactness), so with uneven shards
    straddling F16_MIN_DIM (64 | 63 columns here) or one rank holding an fp16-inexact value the ranks used to disagree about
    entering the flag all-reduce.  Now every rank enters it and all take the weakest rank's answer."""
    world = 2
    mp.spawn(_worker_gate, args=(world, _free_port(), cols, inexact_rank, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(tmp_path / f'g{r}.pt') for r in range(world)]
    assert [g_[0] for g_ in got] == [want] * world, got
    assert sorted(g_[1] for g_ in got) == ([64, 64] if cols == 128 else [63, 64])


def test_bench_script_control_flow_world2(tmp_path):
    """`python bench.py --gpus 2` end to end (its N > 1 control flow had never executed anywhere): the script
    spawns its own two ranks (torch.multiprocessing), they rendezvous on 127.0.0.1, run warm-up / pre-roll / barrier-bracketed
    blocks of exactly K steps with the MAX over ranks, the roofline leg with the all-reduce span, the same-shard 1-GPU
    denominator on rank 0 -- and rank 0 prints exactly ONE JSON line.  `--standin` swaps CUDA tensors / RCCL / the HIP library
    for CPU tensors / gloo / the oracle-backed stand-in backend of this suite: the line measures nothing (and says so); what is
    under test is every line of host code between `main()` and that print."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--standin', '--steps', '3', '--warmup', '1',
                        '--repeats', '2', '--cpu-iters', '0', '--rows', '48', '--cols', '40', '--rank', '5'],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['nranks'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert 'NOT a measurement' in d['data']
    assert len(d['blocks_ms_per_step']) == 2 and d['ms_per_step'] > 0 and d['value'] >= 0    # fixed block count when sharded (the stand-in's tiny shape may round to 0.0 GFLOP/s on a busy host)
    assert d['config']['parallelism'] == 'column-shard x2' and d['config']['cols_per_gpu'] == 40
    rf = d['roofline']
    assert rf['avg_allreduce_ms'] > 0 and rf['avg_launch_ms_h_step'] > 0 and rf['avg_launch_ms_w_step'] > 0
    s1 = d['same_shard_1gpu']
    assert s1['ms_per_step'] > 0 and d['efficiency'] == pytest.approx(s1['ms_per_step'] / d['ms_per_step'], rel=1e-3)
    assert 'another workload' in d['scaling_note']
