#!/usr/bin/env python3
"""Headline benchmark: fused beta-divergence MU iterations of dense NMF on MI355X.

    python bench.py                         # 1 GPU, BASELINE configs[1]: NMF 4096x65536 rank 128 beta=1
    python bench.py --gpus N                # N GPUs of this node: spawns one rank per GPU itself, or, under a launcher:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one MU iteration (W half-step then H half-step, nmf.py:366-391 of the reference) over a synthetic
V already resident in HBM in the engine's packed layout.  With N > 1 every rank owns its own column shard of V and of
W (weak scaling; preset configs[4]: 8192 x 262144 rank 256 per GPU); H is replicated and the H half-step all-reduces
its numerators (RCCL).  `value` is the whole-job algorithmic GFLOP/s (8 * rows * total_cols * rank flops per
iteration); iterations/s is reported next to it.

Timing: W warm-up steps, an untimed pre-roll (`preroll_steps`, ~0.4 s: the chip's clocks settle under load), then
blocks of exactly K steps, each bracketed by barrier + synchronize, until two consecutive blocks agree within 2 %;
the median of the settled blocks is reported and every block is listed.

The headline leg runs in the 'f16' operand mode (fp16 operands and target, fp32 accumulation: bf16's MFMA rate, and the
mode that meets the reference within 1e-4 -- checked in the same run, `parity`); the 'bf16' mode configs[1] names is
timed next to it (`bf16_mode`).  Printed JSON also carries
  roofline      the fused kernel's achieved TFLOP/s (algorithmic flops per launch / mean launch time measured
                live with hipEvents on the launching stream) against the dense bf16 MFMA peak, plus the same
                launch expressed as HBM GB/s of algorithmic bytes;
  cpu_baseline  the reference's ATen op sequence (oracle/aten_port.py, fp32) timed on this box's host cores
                on a bounded sample of the same workload (rank 0, N = 1 only);
  parity        the same k iterations on the GPU and in the CPU reference from identical V, W0, H0: relative errors;
  beta_sweep    BASELINE configs[2]: beta in {2, 0.5, 0} at the same shape (iterations/s, kernel fraction at
                12*N*C*R, parity at the headline's k);
  nmfd          BASELINE configs[3]: NMFD 1025 x 8192, rank 8, T = 400 (iterations/s, per-GEMM fractions, parity);
  nmf2d         SURVEY 8 row f2 (no reference headline): NMF2D 1 x 64 x 256 x 512, rank 8, 8 x 16 kernel, fit()'s own mode;
  ref_notebook  the workload the reference publishes numbers for (benchmark.ipynb: 5168 x 1025, rank 88, five betas, fit());
  roofline.ceiling_tflops / frac_of_ceiling   the zero-overhead MFMA + HBM-stream ceiling measured in this run (nmfmu_ubench_mfma_hbm).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # spec; ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default=None, choices=['cfg1', 'cfg5', 'cfg5full'],
                    help="dense-NMF preset: cfg1 = BASELINE configs[1] (4096x65536 r128; default at 1 GPU), cfg5 = configs[4]'s "
                         "per-GPU shard (8192x262144 r256; default when --gpus > 1), cfg5full = ALL of configs[4] "
                         "(8192x2097152 r256, 64 GiB of packed V) on ONE GPU: the strong-scaling denominator of the 8-GPU run")
    ap.add_argument('--rows', type=int, default=None)
    ap.add_argument('--cols', type=int, default=None, help='columns PER GPU')
    ap.add_argument('--rank', type=int, default=None)
    ap.add_argument('--repeats', type=int, default=5, help='minimum number of timed blocks of --steps steps each; blocks '
                    'are added (up to --max-repeats) until two consecutive ones agree within 2 %%; the median block is reported')
    ap.add_argument('--max-repeats', type=int, default=40)
    ap.add_argument('--preroll-s', type=float, default=0.4, help='untimed iterations before the first block, in seconds of '
                    'GPU work (the chip needs ~0.3 s under load to settle its clocks; reported as preroll_steps)')
    ap.add_argument('--no-parity-mode', action='store_true', help="skip the secondary timed leg in the other single-plane mode")
    ap.add_argument('--no-sweep', action='store_true', help='skip the beta_sweep (configs[2]) and nmfd (configs[3]) sub-objects '
                    'of the default run')
    ap.add_argument('--beta', type=float, default=1.0)
    ap.add_argument('--precision', default=None, choices=['bf16', 'bf16x3', 'f16', 'f16x', 'f16r', 'auto'],
                    help="operand type of the headline leg: 'f16' (default: fp16 operands, bf16's MFMA rate, meets the 1e-4 "
                         "parity bar; nmf and nmfd workloads), 'bf16' (the type configs[1] names; factors ~2e-4 after 3 "
                         "iterations; default of the other workloads), 'bf16x3'")
    ap.add_argument('--cpu-iters', type=int, default=10, help='timed CPU-baseline iterations = iterations of the in-run parity '
                    'check (0 disables both)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--block-rows', type=int, default=None, help='force the 128- or 256-row workgroup tile')
    ap.add_argument('--workload', default='nmf', choices=['nmf', 'nmfd', 'betamu', 'nmf2d', 'sparse', 'plca'],
                    help="'nmfd' = BASELINE configs[3]: NMFD 1x1025x8192 rank 8 T=400 (1 GPU only)")
    ap.add_argument('--taps', type=int, default=400)
    ap.add_argument('--kernel2d', type=int, nargs=2, default=[8, 16],
                    help='nmf2d: kernel size; the target is (1, 64, 256, 512), rank 8 (SURVEY.md 8 row f2: no reference headline)')
    ap.add_argument('--density', type=float, default=0.01, help='sparse: fraction of stored entries of the target')
    ap.add_argument('--materialise', action='store_true',
                    help="betamu: the closure returns m() (reconstruction written out, as in the reference's tests) "
                         "instead of the layer itself")
    ap.add_argument('--force-dist', action='store_true', help='run the sharded (all-reduce) path even at world size 1')
    ap.add_argument('--gram', action='store_true', help="--beta 2 only: time fit()'s path without reconstruction (X @ panel + Gram "
                    'matrix) instead of the 12*N*C*R kernel')
    ap.add_argument('--ref-notebook', action='store_true', help="add the ref_notebook sub-object (the reference's own published "
                    'workload, 5168x1025 rank 88, five betas) to a run that is not the default one')
    ap.add_argument('--telemetry-s', type=float, default=0.8, help='seconds of back-to-back iterations (untimed, after the '
                    'timed blocks) during which a side thread samples core clock and socket power through amdsmi; the means go '
                    'into roofline.clock_mhz / power_w (0 disables)')
    ap.add_argument('--standin', action='store_true',
                    help='TEST ONLY (tests/test_distributed_gloo.py): run the control flow of this script -- rank spawn, rendezvous, '
                         'barrier-bracketed blocks, MAX over ranks, the one JSON line of rank 0 -- on CPU tensors over gloo with the '
                         "oracle-backed stand-in backend of the CPU test-suite (tests/cpu_backend.py).  The numbers of such a run "
                         'measure nothing and the line says so (`data`).')
    a = ap.parse_args()
    if a.precision == 'auto' and a.workload not in ('nmfd', 'nmf2d', 'betamu'):
        ap.error("--precision auto: only for --workload nmfd / nmf2d / betamu (the dense workloads report 'auto' as real_data_mode)")
    return a


# HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE, separate
# passes, MI355X_MICROARCH.md), which cannot run inside this process.  profiles/pmc_traffic.json holds the per-launch bytes
# of the last passes TOGETHER WITH a hash of the kernel sources they measured; `traffic` is emitted only when that hash is
# the hash of the sources this run was built from -- a record of another build reads null (the field must not
# go stale silently).  tools/pmc_traffic_update.py rewrites a record from a fresh pmc_summary.txt.
KERNEL_SOURCES = {'pp': ('nmfmu_pp.h', 'nmfmu_fused.h', 'nmfmu_layout.h', 'nmfmu_inst_pp.hip'),
                  'fused': ('nmfmu_fused.h', 'nmfmu_layout.h', 'nmfmu_inst_r128.hip'),
                  'xb': ('nmfmu_fused.h', 'nmfmu_layout.h', 'nmfmu_inst_r128.hip'),
                  'sp': ('nmfmu_sp.h', 'nmfmu_fused.h', 'nmfmu_layout.h', 'nmfmu_inst_sp.hip')}


def kernel_source_sha(family):
    import hashlib
    h = hashlib.sha256()
    for fn in KERNEL_SOURCES[family]:
        with open(os.path.join(ROOT, 'pytorch-nmf_amd', 'csrc', fn), 'rb') as f:
            h.update(fn.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def pmc_traffic_key(key):
    """(bytes per launch | None, note) for a record key '<rows>x<cols>_r<rank>_<precision>_<pp|fused|xb>'."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        e = d.get('records', {}).get(key)
        if e is None:
            return None, 'no PMC pass recorded for this workload (profiles/pmc_traffic.json)'
        now = kernel_source_sha(key.rsplit('_', 1)[1])
        if e.get('src_sha16') != now:
            return None, (f"the recorded PMC pass ({e.get('source', '?')}) measured another build of this kernel "
                          f"(sources {e.get('src_sha16', 'unrecorded')}, this run {now}); re-run tools/gpu_prof.sh ... pmc")
        return e.get('hbm_bytes_per_launch'), f"{e.get('source', '')} -- same kernel sources as this run ({now})"
    except Exception as ex:
        return None, f'{type(ex).__name__}: {ex}'


def pmc_traffic(N, C, R, precision, family):
    return pmc_traffic_key(f'{N}x{C}_r{R}_{precision}_{family}')


def usable_cores():
    """Host threads this process may actually run on: affinity mask, capped by a cgroup CPU quota if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def pick_threads(run_probe, max_cands=None):
    """The ATen CPU kernels are far from monotone in thread count on many-core hosts (256 threads were measured ~4x
    slower than 32 on the MI355X box).  Time a probe with a few thread counts and keep the fastest, so the CPU baseline
    is the reference at its best on this box, not at its default."""
    n = usable_cores()
    cands = sorted({max(1, c) for c in (n, n // 2, n // 4, n // 8, 32, 16) if c <= n} | {min(n, 8)}, reverse=True)
    if max_cands:
        cands = cands[:max_cands]            # slow probes (seconds each): only the largest counts
    best, tried = None, {}
    for c in cands:
        torch.set_num_threads(c)
        run_probe()                      # warm the pool at this size
        t0 = time.perf_counter()
        run_probe()
        tried[c] = round(time.perf_counter() - t0, 4)
        if best is None or tried[c] < tried[best]:
            best = c
    torch.set_num_threads(best)
    return best, tried


class SmiSampler:
    """Core clock and socket power while the GPU is under the benchmark's own load, read through amdsmi (the library
    behind `amd-smi` / `rocm-smi --showpower --showclocks`) by a side thread -- the driver-visible evidence for what the
    MFMA loops run at (they are power-limited).  Everything is best effort: a box without amdsmi, or
    a metrics table without a field, yields None for it; nothing here is on the timed path."""

    def __init__(self, index=0):
        self.ok, self.err, self.h = False, None, None
        try:
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h = hs[min(index, len(hs) - 1)]
            self.ok = True
        except Exception as e:                       # no library / no permission: report why
            self.err = f'{type(e).__name__}: {e}'[:200]

    @staticmethod
    def _num(v):
        return float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else None

    def read(self):
        """One sample: {'t', 'gfxclk_mhz' (mean over the XCDs that report), 'power_w', 'energy', ...}."""
        out = {'t': time.perf_counter()}
        m = {}
        try:
            m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        except Exception as e:
            self.err = f'metrics: {type(e).__name__}: {e}'[:200]
        clks = m.get('current_gfxclks')
        clks = [float(c) for c in clks if isinstance(c, (int, float)) and 0 < c < 10000] if isinstance(clks, (list, tuple)) else []
        if not clks and self._num(m.get('current_gfxclk')):
            clks = [float(m['current_gfxclk'])]
        if not clks:
            try:
                c = self.smi.amdsmi_get_clock_info(self.h, self.smi.AmdSmiClkType.GFX)
                if self._num(c.get('clk')):
                    clks = [float(c['clk'])]
            except Exception:
                pass
        out['gfxclk_mhz'] = sum(clks) / len(clks) if clks else None
        out['gfxclk_xcd_mhz'] = clks
        pw = self._num(m.get('current_socket_power'))
        if pw is None or pw <= 0 or pw >= 65535:
            pw = self._num(m.get('average_socket_power'))
        if pw is None or pw <= 0 or pw >= 65535:
            try:
                pi = self.smi.amdsmi_get_power_info(self.h)
                for k in ('current_socket_power', 'socket_power', 'average_socket_power'):
                    if self._num(pi.get(k)) and 0 < pi[k] < 65535:
                        pw = float(pi[k])
                        break
            except Exception:
                pass
        out['power_w'] = pw if pw and 0 < pw < 65535 else None
        out['energy'] = self._num(m.get('energy_accumulator'))
        out['mem_activity'] = self._num(m.get('average_umc_activity'))
        out['temp_hotspot'] = self._num(m.get('temperature_hotspot'))
        return out

    def power_limit_w(self):
        try:
            pi = self.smi.amdsmi_get_power_info(self.h)
            v = self._num(pi.get('power_limit'))
            if v and v > 10000:          # some versions report microwatts
                v /= 1e6
            return v
        except Exception:
            return None

    def under_load(self, step, seconds=1.0, period=0.04):
        """Run `step` back to back for `seconds` (untimed) while a side thread samples; returns the summary dict."""
        if not self.ok:
            return {'source': 'amdsmi', 'available': False, 'error': self.err}
        import threading
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append(self.read())
                stop.wait(period)
        idle = self.read()
        th = threading.Thread(target=poll, daemon=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        # let the load run for a quarter of the window before the first sample counts (clock settles, SMI averages catch up)
        while time.perf_counter() - t0 < 0.25 * seconds:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
        th.start()
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < seconds:
            for _ in range(20):
                step()
            n += 20
            torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        stop.set()
        th.join()
        mean = lambda xs: (round(sum(xs) / len(xs), 1) if xs else None)
        clk = [s_['gfxclk_mhz'] for s_ in samples if s_['gfxclk_mhz']]
        pw = [s_['power_w'] for s_ in samples if s_['power_w']]
        out = {'source': 'amdsmi gpu_metrics, side thread, %d samples over %.2f s of back-to-back iterations (untimed leg)' % (len(samples), dt),
               'available': True, 'clock_mhz': mean(clk), 'clock_mhz_min_max': [min(clk), max(clk)] if clk else None,
               'power_w': mean(pw), 'power_w_max': max(pw) if pw else None, 'power_limit_w': self.power_limit_w(),
               'idle_clock_mhz': idle.get('gfxclk_mhz'), 'idle_power_w': idle.get('power_w'),
               'iters_per_s_during_sampling': round(n / dt, 1),
               'hotspot_c': mean([s_['temp_hotspot'] for s_ in samples if s_['temp_hotspot']]),
               'hbm_activity_pct': mean([s_['mem_activity'] for s_ in samples if s_['mem_activity'] is not None])}
        e = [s_['energy'] for s_ in samples if s_['energy']]
        if len(e) >= 2 and e[-1] > e[0]:
            # energy accumulator: 15.259 uJ per count (amdsmi's documented resolution for this family)
            out['power_w_from_energy_counter'] = round((e[-1] - e[0]) * 15.259e-6 / (samples[-1]['t'] - samples[0]['t']), 1)
        if self.err:
            out['last_error'] = self.err
        return out


STANDIN = False   # --standin: CPU tensors, gloo, the test-suite's oracle-backed backend (no measurement)


def dev_sync():
    if not STANDIN:
        torch.cuda.synchronize()


class HostTimer:
    """--standin only: the KernelTimer interface on the host clock (there are no device events on CPU tensors)."""

    def __init__(self, n_events):
        self.t, self.tags = [], []

    def mark(self, tag):
        self.t.append(time.perf_counter())
        self.tags.append(tag)

    def spans(self):
        out = {}
        for i in range(len(self.tags) - 1):
            a, b = self.tags[i], self.tags[i + 1]
            if a.endswith('<') and b == a[:-1] + '>':
                out.setdefault(a[:-1], []).append(1e3 * (self.t[i + 1] - self.t[i]))
        return out

    def close(self):
        pass


def preroll_steps(step, seconds):
    """Untimed iterations until `seconds` of GPU work have passed: under this load the chip takes a few tenths of a second
    to settle its clocks (the first blocks of a cold run were up to 18 % slower than the settled ones)."""
    n = 0
    dev_sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            step()
        dev_sync()
        n += 20
    return n


def timed_blocks(run, steps, min_repeats, max_repeats, barrier=None, reduce_max=None):
    """Blocks of exactly `steps` steps, each bracketed by barrier + synchronize (MAX over ranks); at least `min_repeats`,
    then more until two consecutive blocks agree within 2 % (or `max_repeats`).  Returns (median ms/step, all blocks)."""
    def sync():
        dev_sync()
        if barrier is not None:
            barrier()
    blocks = []
    while True:
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        sync()
        elapsed = time.perf_counter() - t0
        if reduce_max is not None:
            elapsed = reduce_max(elapsed)
        blocks.append(1e3 * elapsed / steps)
        n = len(blocks)
        settled = n >= 2 and abs(blocks[-1] - blocks[-2]) <= 0.02 * blocks[-1]
        if n >= max(1, min_repeats) and (settled or n >= max_repeats):
            break
    tail = blocks[-max(1, min_repeats):]          # the settled end of the run
    return sorted(tail)[len(tail) // 2], blocks


def cpu_baseline(V, W0, H0, beta, iters, betamu=False):
    """Time the reference's op sequence on the host cores (bounded sample of the same workload)."""
    from oracle import aten_port
    torch.set_flush_denormal(True)   # the reference's own advice (README.md:101-102)
    cs = min(V.shape[1], 8192)       # probe = one MU iteration on a column slice of the same workload
    Vs, Ws = V[:, :cs].contiguous(), W0[:cs].contiguous()
    run = aten_port.betamu_iterations if betamu else aten_port.mu_iterations
    cores, tried = pick_threads(lambda: run(Vs, Ws, H0, beta, 1))
    run(V, W0, H0, beta, 1)          # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    res = run(V, W0, H0, beta, iters)
    dt = (time.perf_counter() - t0) / iters
    Wr, Hr = (res[0], res[1]) if isinstance(res, tuple) and len(res) >= 2 else (None, None)
    return dt, cores, tried, Wr, Hr


def nmfd_line(a, sub=False):
    """BASELINE configs[3]: NMFD spectrogram 1025 x 8192, rank 8, T = 400, beta = 1 (replicas only: 1 GPU).  Returns the
    JSON object; sub=True is the trimmed form embedded in the default run (precision f16 headline)."""
    dev = torch.device('cuda', 0)
    from torchnmf_amd.nmfd_engine import ConvMU
    from torchnmf_amd.engine import KernelTimer
    kind, prec_head = a.workload, a.precision
    Cc, L, R, T, beta = (a.rows or 1025), (a.cols or 8192), (a.rank or 8), a.taps, a.beta   # --rows = channels, --cols = frames
    g = torch.Generator(device=dev).manual_seed(1000)
    if a.workload == 'nmf2d':       # NMF2D, the next row after NMFD (same engine, two shift axes)
        Cc, R = 64, 8
        ls, ks = (256, 512), tuple(a.kernel2d)
        V = torch.rand(1, Cc, *ls, device=dev, generator=g).bfloat16().float()
        W = torch.randn(Cc, R, *ks, device=dev, generator=g).abs_()
        H = torch.randn(1, R, *[l - k + 1 for l, k in zip(ls, ks)], device=dev, generator=g).abs_()
        L, T = ls[0] * ls[1], ks[0] * ks[1]
        title = f'NMF2D 1x{Cc}x{ls[0]}x{ls[1]} rank={R} kernel={ks[0]}x{ks[1]}'
    else:
        V = torch.rand(1, Cc, L, device=dev, generator=g).bfloat16().float()
        W = torch.randn(Cc, R, T, device=dev, generator=g).abs_()
        H = torch.randn(1, R, L - T + 1, device=dev, generator=g).abs_()
        title = f'NMFD 1x{Cc}x{L} rank={R} T={T}' + (' (BASELINE configs[3])' if (Cc, L, R, T) == (1025, 8192, 8, 400) else '')
    Vc, Wc, Hc = V.cpu(), W.cpu(), H.cpu()
    if a.precision == 'auto':          # what fit() picks for this problem (one set-up sync)
        a.precision = prec_head = ConvMU(V, W.clone(), H.clone(), beta, precision='auto').precision_name
    eng = ConvMU(V, W, H, beta, precision=a.precision)

    def step():
        eng.w_step()
        eng.h_step()
    for _ in range(a.warmup):
        step()
    preroll = preroll_steps(step, a.preroll_s)
    ms, blocks = timed_blocks(step, a.steps, a.repeats, a.max_repeats)
    flops = (4.0 if beta == 1 else 6.0) * 2.0 * Cc * L * R * T
    # second timed leg in the parity-grade mode (split bf16, 3 MFMAs per product) so that the line carries both
    pm = None
    if a.precision not in ('bf16x3', 'f16') and not a.no_parity_mode and a.workload == 'nmfd' and not sub:
        # the parity-grade mode fit() picks by itself: fp16 operands where built (beta == 1, >= 128 taps), else split bf16
        pprec = 'f16' if (beta == 1 and T >= 128 and T % 8 == 0 and L % 8 == 0) else 'bf16x3'
        Wp, Hp = W.clone(), H.clone()
        e3 = ConvMU(V, Wp, Hp, beta, precision=pprec)
        for _ in range(max(3, a.warmup // 2)):
            e3.w_step(); e3.h_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            e3.w_step(); e3.h_step()
        torch.cuda.synchronize()
        ms3 = 1e3 * (time.perf_counter() - t0) / a.steps
        pm = {'precision': pprec, 'dtype': 'f16 operands / fp32 accumulate (same MFMA rate as bf16)' if pprec == 'f16'
              else 'bf16x3 (split bf16, fp32-grade: 3 MFMAs per product)',
              'iters_per_s': round(1e3 / ms3, 2), 'ms_per_step': round(ms3, 4),
              'value': round((4.0 if beta == 1 else 6.0) * 2.0 * Cc * L * R * T / (ms3 * 1e-3) / 1e9, 1), 'unit': 'GFLOP/s'}
        del e3
    # the dominant kernel (nt_gemm) timed live: 4 launches per iteration (beta == 1), each 2*C*L*R*T algorithmic flops,
    # every one bracketed by hipEvents on the launching stream as the iteration runs it (the reconstructions run over the
    # channels that fill whole 128-row tiles; the ragged-channel kernel next to them is outside the brackets)
    eng.timer = KernelTimer(8 * a.steps + 8)
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    spans = eng.timer.spans()
    eng.timer.close()
    eng.timer = None
    tel = SmiSampler(0).under_load(step, a.telemetry_s) if a.telemetry_s > 0 else None
    fit_obj = None
    if beta == 1:      # what the user calls: NMFD.fit / NMF2D.fit end to end (200 iterations, 20 loss checkpoints)
        fit_obj = fit_leg(a, V, Wc, Hc, beta, a.precision, dev, ms, cls_name='NMFD' if a.workload == 'nmfd' else 'NMF2D')
    gflop = 2.0 * Cc * L * R * T
    per_gemm = {k: {'avg_launch_ms': round(sum(v) / len(v), 5),
                    'frac': round(gflop / (sum(v) / len(v) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)} for k, v in spans.items()}
    all_ms = [x for v in spans.values() for x in v]
    gemm_ms = sum(all_ms) / len(all_ms)
    ach = gflop / (gemm_ms * 1e-3) / 1e12
    peak = MFMA_BF16_PEAK_TFLOPS
    cpu = None
    parity = None
    cpu_iters = a.cpu_iters            # (round 5: the embedded legs run the headline's k as well; was 3)
    if cpu_iters > 0:
        from oracle import aten_port
        torch.set_flush_denormal(True)
        cores, tried = pick_threads(lambda: aten_port.mu_iterations_nmfd(Vc, Wc, Hc, beta, 1), max_cands=2 if sub else None)
        aten_port.mu_iterations_nmfd(Vc, Wc, Hc, beta, 1)
        t0 = time.perf_counter()
        Wr, Hr = aten_port.mu_iterations_nmfd(Vc, Wc, Hc, beta, cpu_iters)
        dt = (time.perf_counter() - t0) / cpu_iters
        # in-run parity: the same k iterations on the GPU from the same V, W0, H0, per precision mode
        def rel(x, y):
            return float((x.double() - y.double()).norm() / y.double().norm())
        modes = {}
        for prec in dict.fromkeys([prec_head] + (['f16'] if kind == 'nmfd' and beta == 1 and T >= 128 else []) + (['bf16'] if sub else ['bf16x3'])):
            Wg, Hg = Wc.clone().to(dev), Hc.clone().to(dev)
            e2 = ConvMU(V, Wg, Hg, beta, precision=prec)
            for _ in range(cpu_iters):
                e2.w_step()
                e2.h_step()
            torch.cuda.synchronize()
            rw, rh = rel(Wg.cpu(), Wr), rel(Hg.cpu(), Hr)
            modes[prec] = {'rel_W': float(f'{rw:.4g}'), 'rel_H': float(f'{rh:.4g}'), 'meets_1e-4': bool(max(rw, rh) < 1e-4)}
            del e2
        parity = {'k': cpu_iters, 'reference': 'oracle/aten_port.py (fp32, same V, W0, H0), the timed CPU iterations',
                  'bar': 1e-4, 'modes': modes}
        cpu = {'value': round(flops / dt / 1e9, 2), 'unit': 'GFLOP/s', 'cores': cores, 'kind': 'port',
               'host_cores': usable_cores(), 'thread_probe_s': tried, 'iters_per_s': round(1 / dt, 4), 'sample': f'{cpu_iters} timed MU iterations (+1 warm-up) of the same '
               f'workload, fp32, F.conv{2 if a.workload == "nmf2d" else 1}d + two backward passes (oracle/aten_port.py)'}
    return {
        'metric': f'MU GFLOP/s (algorithmic 8*C*L*R*T per iteration), {title} beta={beta:g}',
        'value': round(flops / (ms * 1e-3) / 1e9, 1), 'unit': 'GFLOP/s', 'iters_per_s': round(1e3 / ms, 2), 'n_gpus': 1,
        'steps': a.steps, 'warmup': a.warmup, 'preroll_steps': preroll, 'ms_per_step': round(ms, 4),
        'blocks_ms_per_step': [round(x, 4) for x in blocks], 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': {'bf16': 'bf16', 'f16': 'f16', 'bf16x3': 'bf16x3 (split bf16, fp32-grade)'}[a.precision],
        'data': 'synthetic',
        'config': {'workload': f'{title} beta={beta:g}',
                   'precision': a.precision, 'parallelism': 'single GPU (replicas only)', 'launch': 'eager launches'},
        'roofline': {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': round(ach / peak, 4), 'traffic': None,
                     'kernel': 'nmfmu::nt_gemm_kernel (mean over the GEMM launches of an iteration: reconstruction + ratio of both '
                               'half-steps, W numerator, H numerator ' + ('as the window-operand GEMM over shifted ratio rows)'
                                                                          if getattr(eng, 'h_rows', False) else 'with the fold epilogue)'),
                     'avg_launch_ms': round(gemm_ms, 5), 'per_gemm': per_gemm,
                     'clock_mhz': tel.get('clock_mhz') if tel else None, 'power_w': tel.get('power_w') if tel else None,
                     'telemetry': tel,
                     'note': 'bf16x3 issues 3 MFMAs per algorithmic product: hardware MFMA rate is 3x achieved'
                     if a.precision == 'bf16x3' else ''},
        'cpu_baseline': cpu, 'parity': parity, 'parity_mode': pm, 'fit': fit_obj}


def main_nmfd(a):
    print(json.dumps(nmfd_line(a)))


def main_sparse(a):
    """NMF.fit on a sparse-COO target (nmf.py:351-398, 602-638); no reference headline.
    Default: 32768 x 32768, 1 % stored entries, rank 64, beta = 1."""
    dev = torch.device('cuda', 0)
    from torchnmf_amd.sparse_engine import SparseMU
    N = a.rows if a.rows is not None else 32768
    Cc = a.cols if a.cols is not None else 32768
    R = a.rank if a.rank is not None else 64
    beta = a.beta
    g = torch.Generator(device=dev).manual_seed(1000)
    nnz = int(N * Cc * a.density)
    flat = torch.randint(0, N * Cc, (nnz,), device=dev, generator=g).unique()
    idx = torch.stack([flat // Cc, flat % Cc])
    vals = torch.rand(flat.numel(), device=dev, generator=g) + 1e-3
    V = torch.sparse_coo_tensor(idx, vals, (N, Cc)).coalesce()
    nnz = V._nnz()
    W = torch.randn(Cc, R, device=dev, generator=g).abs_()
    H = torch.randn(N, R, device=dev, generator=g).abs_()
    Vc, Wc, Hc = V.cpu(), W.cpu(), H.cpu()
    eng = SparseMU(V, W, H, beta)

    def step():
        eng.w_step()
        eng.h_step()
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    flops = 8.0 * nnz * R if beta == 1 else 4.0 * nnz * R      # per iteration: (dot + axpy) x 2 half-steps (beta 2: axpy only)
    # dominant kernel (sp_partial_kernel) timed live: one half-step's numerator pass
    st, csr = eng.step_h, eng.csr_h
    from torchnmf_amd import _capi

    def partial():
        _capi.check(eng.lib.nmfmu_sp_partial(csr[0].data_ptr(), csr[1].data_ptr(), csr[2].data_ptr(), st.owner.rows,
                                             st.owner.f.data_ptr(), st.panel.f.data_ptr(), R, beta, st.slab_num.data_ptr(),
                                             eng.r_pad, torch.cuda.current_stream().cuda_stream), 'nmfmu_sp_partial')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    partial()
    ev[0].record()
    for _ in range(a.steps):
        partial()
    ev[1].record()
    torch.cuda.synchronize()
    k_ms = ev[0].elapsed_time(ev[1]) / a.steps
    bytes_per_launch = nnz * (R * 4 + 8.0) + N * R * 8.0     # one panel row + (column index, value) per entry; owner in/out
    cpu = None
    if a.cpu_iters > 0:
        from oracle import aten_port
        torch.set_flush_denormal(True)
        cores, tried = pick_threads(lambda: aten_port.sp_mu_iterations(Vc, Wc, Hc, beta, 1), max_cands=2)
        t0 = time.perf_counter()
        aten_port.sp_mu_iterations(Vc, Wc, Hc, beta, a.cpu_iters)
        dt = (time.perf_counter() - t0) / a.cpu_iters
        cpu = {'value': round(flops / dt / 1e9, 2), 'unit': 'GFLOP/s', 'cores': cores, 'kind': 'port',
               'host_cores': usable_cores(), 'thread_probe_s': tried, 'iters_per_s': round(1 / dt, 4),
               'sample': f'{a.cpu_iters} timed MU iterations (+1 probe per thread count) of the same sparse workload, fp32, '
                         f"the reference's scalar-objective op sequence (oracle/aten_port.py: sp_mu_iterations)"}
    print(json.dumps({
        'metric': f'MU GFLOP/s (algorithmic 8*nnz*R per iteration), sparse NMF {N}x{Cc}, {nnz} stored entries, rank-{R} '
                  f'beta={beta:g}; MU iterations/s alongside',
        'value': round(flops / (ms * 1e-3) / 1e9, 1), 'unit': 'GFLOP/s', 'iters_per_s': round(1e3 / ms, 2), 'n_gpus': 1,
        'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'sparse NMF {N}x{Cc} density={nnz / (N * Cc):.4f} rank={R} beta={beta:g} (SURVEY 8 row f3)',
                   'nnz': nnz, 'parallelism': 'single GPU'},
        'roofline': {'bound': 'hbm', 'achieved': round(bytes_per_launch / (k_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': round(bytes_per_launch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     'traffic': None, 'kernel': 'nmfmu::sp_partial_kernel', 'avg_launch_ms': round(k_ms, 5),
                     'note': 'algorithmic bytes = one panel row (4R B) + index and value (8 B) per stored entry; panel rows '
                             'are re-read from L2 / MALL when columns repeat, so HBM traffic is below this figure'},
        'cpu_baseline': cpu}))


def main_plca(a):
    """PLCA's EM iteration (plca.py:248-290) on the fused kernels; shape = BASELINE configs[1]."""
    dev = torch.device('cuda', 0)
    from torchnmf_amd.plca import _PlcaEM, get_norm
    N, Cc, R = a.rows or 4096, a.cols or 65536, a.rank or 128
    g = torch.Generator(device=dev).manual_seed(1000)
    V = torch.rand(N, Cc, device=dev, generator=g).bfloat16().float()
    W = torch.rand(Cc, R, device=dev, generator=g)
    H = torch.rand(N, R, device=dev, generator=g)
    Z = torch.rand(R, device=dev, generator=g)
    Vc, Wc, Hc, Zc = (V.cpu(), W.cpu(), H.cpu(), Z.cpu()) if a.cpu_iters > 0 else (None,) * 4
    for t_ in (W, H, Z):
        t_.div_(get_norm(t_))
    em = _PlcaEM((V / V.sum()).contiguous(), W, H, Z, a.precision)
    del V

    def step():
        em.em_step(True, True, True, 1.0, 1.0, 1.0)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    flops = 6.0 * N * Cc * R          # as the reference computes it: one reconstruction + the two backward products
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    em.be.mu_partial(em.step_h)
    ev[0].record()
    for _ in range(a.steps):
        em.be.mu_partial(em.step_h)
    ev[1].record()
    torch.cuda.synchronize()
    k_ms = ev[0].elapsed_time(ev[1]) / a.steps
    ach = 4.0 * N * Cc * R / (k_ms * 1e-3) / 1e12
    cpu = None
    if a.cpu_iters > 0:
        from oracle import aten_port
        torch.set_flush_denormal(True)
        cs = min(Cc, 8192)
        cores, tried = pick_threads(lambda: aten_port.plca_iterations(Vc[:, :cs].contiguous(), Wc[:cs].contiguous(), Hc, Zc, 1))
        aten_port.plca_iterations(Vc, Wc, Hc, Zc, 1)
        t0 = time.perf_counter()
        aten_port.plca_iterations(Vc, Wc, Hc, Zc, a.cpu_iters)
        dt = (time.perf_counter() - t0) / a.cpu_iters
        cpu = {'value': round(flops / dt / 1e9, 2), 'unit': 'GFLOP/s', 'cores': cores, 'kind': 'port',
               'host_cores': usable_cores(), 'thread_probe_s': tried, 'iters_per_s': round(1 / dt, 4),
               'sample': f"{a.cpu_iters} timed EM iterations (+1 warm-up) of the same workload, fp32, the reference's op "
                         f'sequence (oracle/aten_port.py: plca_iterations)'}
    print(json.dumps({
        'metric': f'EM GFLOP/s (algorithmic 6*N*C*R per iteration), PLCA {N}x{Cc} rank-{R}; EM iterations/s alongside',
        'value': round(flops / (ms * 1e-3) / 1e9, 1), 'unit': 'GFLOP/s', 'iters_per_s': round(1e3 / ms, 2), 'n_gpus': 1,
        'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': {'bf16': 'bf16', 'f16': 'f16', 'bf16x3': 'bf16x3 (split bf16, fp32-grade)'}[a.precision],
        'data': 'synthetic',
        'config': {'workload': f'PLCA {N}x{Cc} rank={R} (SURVEY 8 row f4; shape of BASELINE configs[1])',
                   'precision': a.precision, 'parallelism': 'single GPU'},
        'roofline': {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None, 'kernel': 'nmfmu::fused_kernel',
                     'avg_launch_ms': round(k_ms, 5),
                     'note': 'each EM iteration launches the fused kernel twice (both contractions recompute the '
                             'reconstruction: 8*N*C*R executed for 6*N*C*R algorithmic)'},
        'cpu_baseline': cpu}))


def dense_leg(a, V, W0, H0, beta, precision, group, world, dev, want_roofline, betamu=False, blocks_min=None, telemetry=False,
              gram=False):
    """W warm-up steps, an untimed pre-roll, then blocks of exactly K steps, each bracketed by barrier + synchronize; the
    block time is the MAX over ranks, the reported ms/step the median of the settled blocks.  Returns a dict."""
    from torchnmf_amd.engine import DenseMU, KernelTimer
    N, C = V.shape
    R = W0.shape[1]
    flops_per_iter_gpu = (8.0 if beta == 1 else 12.0) * N * C * R     # 4 (6) contractions of 2NCR
    if gram and beta == 2:
        flops_per_iter_gpu = 4.0 * N * C * R                         # executed: two X @ panel GEMMs (never priced as 12 N C R)
    W, H = W0.clone(), H0.clone()
    if betamu:
        # trainer.BetaMu.step(closure) on one NMF layer; one step = W update + H update
        assert world == 1 and group is None, 'BetaMu runs on one GPU'
        from torchnmf_amd.nmf import NMF
        from torchnmf_amd.trainer import BetaMu
        layer = NMF(W=W.cpu(), H=H.cpu()).to(dev)
        trainer = BetaMu(layer.parameters(), beta, precision=precision)

        def closure():
            trainer.zero_grad()
            return V, (layer() if a.materialise else layer)

        def step():
            trainer.step(closure)
        step()
        eng = next(iter(trainer._engines.values()))[0]
    else:
        eng = DenseMU(V, W, H, beta, precision=precision, group=group, block_rows=a.block_rows, allow_gram=gram)
        assert eng.gram_path == (gram and beta == 2 and not (precision == 'f16x' and R > 128))

        def step():
            eng.w_step()
            eng.h_step()
    dev_sync()
    for _ in range(a.warmup):
        step()
    if world > 1:
        # every rank must run the SAME number of steps (each holds collectives): a fixed pre-roll instead of a timed one
        preroll = 60
        for _ in range(preroll):
            step()
        dev_sync()
    else:
        preroll = preroll_steps(step, a.preroll_s)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dev_sync()

    def reduce_max(elapsed):
        if world == 1:
            return elapsed
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())
    # (multi-rank: a fixed block count, so that every rank runs the same number of collectives)
    ms, blocks = timed_blocks(step, a.steps, blocks_min or a.repeats, (blocks_min or a.repeats) if world > 1 else a.max_repeats,
                              barrier, reduce_max)
    out = {'precision': precision, 'ms_per_step': ms, 'blocks_ms_per_step': [round(x, 4) for x in blocks],
           'preroll_steps': preroll, 'gflops': flops_per_iter_gpu * world / (ms * 1e-3) / 1e9, 'eng': eng}
    # ---- roofline leg: K more steps, right after the timed region, with hipEvents around every fused launch
    # (recording inside the timed region costs ~2 % and folds the dispatch gap in front of each kernel into its
    # span; these spans agree with the rocprofv3 kernel-trace durations)
    if want_roofline:
        nst = max(a.steps, 40)
        eng.timer = (HostTimer if STANDIN else KernelTimer)(8 * nst + 8)
        for _ in range(nst):
            step()
        dev_sync()
        spans = eng.timer.spans()
        eng.timer.close()
        eng.timer = None
        if 'h' not in spans and 'h0' in spans:      # sharded path: the H half-step runs as two row halves
            spans['h'] = [x + y for x, y in zip(spans['h0'], spans['h1'])]
        all_ms = spans.get('w', []) + spans.get('h', [])
        avg_ms = sum(all_ms) / len(all_ms)
        flops_per_launch = flops_per_iter_gpu / 2.0                    # one half-step = 2 (3) contractions
        elt = {'bf16x3': 4, 'f16x': 4, 'f16r': 3}.get(eng.precision_name, 2)
        bytes_per_launch = N * C * elt + 1.5 * (C * R + N * R) * 4    # one read of V + half the factor traffic
        ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
        # which kernel ran (include/nmfmu.h: NMFMU_KERNEL_*): the ping-pong kernel, the software-pipelined rank-256 kernel
        # (round 6) or the four-wave kernel
        fam = 1 if eng.step_h.block_rows == 256 else 0
        if fam == 0 and hasattr(eng.be, 'kernel_family') and not (gram and beta == 2):
            fam = eng.be.kernel_family(eng.r_pad, eng.precision, float(beta))
        family = {0: 'fused', 1: 'pp', 2: 'sp'}[fam]
        roof = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': pmc_traffic(N, C, R, eng.precision_name, family)[0],
                'traffic_source': pmc_traffic(N, C, R, eng.precision_name, family)[1],
                'kernel': f'nmfmu::{family}_kernel', 'launches_timed': len(all_ms),
                'avg_launch_ms': round(avg_ms, 5),
                'avg_launch_ms_w_step': round(sum(spans['w']) / len(spans['w']), 5),
                'avg_launch_ms_h_step': round(sum(spans['h']) / len(spans['h']), 5),
                'outside_fused_kernels_ms': round(ms - 2 * avg_ms, 5),
                'flops_per_launch': flops_per_launch,
                'note': 'W-step launches carry the MU apply (nmf.py:78-92) in their epilogue when the contraction is not '
                        'split; H-step launches are the MFMA main loop + slab stores',
                'achieved_main_loop_only': round(flops_per_launch / (sum(spans['h']) / len(spans['h']) * 1e-3) / 1e12, 2),
                'hbm': {'achieved': round(bytes_per_launch / (avg_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s', 'frac': round(bytes_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        'algorithmic_bytes_per_launch': int(bytes_per_launch)}}
        if telemetry and a.telemetry_s > 0 and world == 1:
            # driver-visible clock / power evidence: what the chip grants under THIS loop
            tel = SmiSampler(dev.index or 0).under_load(step, a.telemetry_s)
            roof['clock_mhz'], roof['power_w'] = tel.get('clock_mhz'), tel.get('power_w')
            if tel.get('clock_mhz'):
                pk = MFMA_BF16_PEAK_TFLOPS * tel['clock_mhz'] / 2400.0
                roof['peak_at_measured_clock'] = round(pk, 1)
                roof['frac_of_peak_at_measured_clock'] = round(ach / pk, 4)
            roof['telemetry'] = tel
        if telemetry and world == 1 and hasattr(eng.be, 'lib') and hasattr(eng.be.lib, 'nmfmu_ubench_mfma_hbm2') and not betamu:
            try:
                ceil = ceiling_leg(a, eng, dev)
            except Exception as ex:          # a diagnostic leg must never take the bench line down with it
                ceil = None
                roof['ceiling_error'] = f'{type(ex).__name__}: {ex}'[:200]
            if ceil is not None:
                roof['ceiling'] = ceil
                roof['ceiling_tflops'] = ceil['with_stream']['tflops']
                roof['frac_of_ceiling'] = round(ach / ceil['with_stream']['tflops'], 4)
                roof['ceiling_note'] = ('ceiling_tflops = what a loop with NO overhead (no LDS operand reads, no elementwise stage, '
                                        'no barriers, fixed operands) sustains on this box in this run at the MU step\'s X bytes per '
                                        'MFMA, over a loop long enough that launch ramp and drain are < 2 % (ceiling.*.mfma_busy_frac '
                                        'is its matrix pipe\'s busy fraction from in-kernel stamps); `frac` stays priced against the '
                                        'nominal dense peak')
        if telemetry and world == 1 and not STANDIN and family in ('pp', 'sp') and not betamu:
            try:
                ik = in_kernel_leg(eng, step, dev, family)
                if ik is not None:
                    roof['in_kernel'] = ik
                    if roof.get('ceiling'):
                        roof['ceiling']['shipped_kernel'] = ik
            except Exception as ex:
                roof['in_kernel_error'] = f'{type(ex).__name__}: {ex}'[:200]
        if 'ar' in spans:
            roof['avg_allreduce_ms'] = round(sum(spans['ar']) / len(spans['ar']), 5)
            roof['allreduce_note'] = ('exposed part: the first row half of the H numerators is reduced behind the '
                                      'second half\'s kernel' if 'h0' in spans else 'blocking, after the H half-step kernel')
        out['roofline'] = roof
    return out


def in_kernel_leg(eng, step, dev, family):
    """The shipped kernel's own clocks (nmfmu_step.stamps): workgroup 0 stamps shader cycles and the
    constant 100 MHz clock at the start and the end of its tile loop -- cycles per tile, the core clock INSIDE the kernel
    and the matrix pipe's busy fraction = 32 cycles x MFMAs per SIMD and tile / cycles per tile.  H half-step (the plain
    tile loop + slab stores); a few iterations after the timed legs, median."""
    import numpy as np
    st = eng.step_h
    nwg = (st.owner.rows_pad // st.block_rows) * st.nsplit
    buf = torch.zeros(64 + 5 * nwg, dtype=torch.int64, device=dev)
    mfma_per_tile_simd = (2 if family == 'pp' else 1) * (eng.r_pad // 4)
    res = []
    try:
        st.struct.stamps = buf.data_ptr()
        for _ in range(7):
            for _ in range(25):          # back to back: the clock the sustained loop runs at (a lone launch clocks higher)
                step()
            torch.cuda.synchronize()
            v = buf[:8].cpu().numpy().reshape(2, 4)          # slots 0 (loop start) and 1 (loop end): cycles, 100 MHz ticks, tiles
            nt = int(v[0, 2])
            cyc, ticks = int(v[1, 0] - v[0, 0]), int(v[1, 1] - v[0, 1])
            if nt > 0 and cyc > 0 and ticks > 0:
                res.append((cyc / nt, cyc / ticks * 100.0))
    finally:
        st.struct.stamps = None
    if not res:
        return None
    r = np.array(res)
    cpt, clk = float(np.median(r[:, 0])), float(np.median(r[:, 1]))
    return {'what': 'clock stamps of workgroup 0 around its tile loop (nmfmu_step.stamps; H half-step; the last launch of %d bursts of 25 back-to-back iterations, median)' % len(res),
            'kernel': f'nmfmu::{family}_kernel', 'tiles_per_workgroup': nt, 'cycles_per_tile': round(cpt, 1),
            'mfma_cycles_per_tile_and_simd': 32 * mfma_per_tile_simd,
            'mfma_busy_frac': round(32 * mfma_per_tile_simd / cpt, 4), 'clock_mhz_in_kernel': round(clk, 1),
            'ns_per_tile': round(cpt / clk * 1e3, 1)}


def ceiling_leg(a, eng, dev):
    """The zero-overhead ceiling of the fused MU step on THIS box, in THIS run: nmfmu_ubench_mfma_hbm2
    = 8 waves per CU issuing 32 MFMAs per wave and tile on fixed fragments read from the live W image (the real operand
    distribution) next to an independent non-temporal stream of the live packed X at the MU step's bytes per MFMA; no LDS
    reads, no VALU, no barriers.  Round 6: the loops run LONG (1 024 tiles without the stream, 512 with it,
    wrapping over X) so that ramp, operand fetch and drain are < 2 % of a launch, and every workgroup stamps its loop: the
    matrix pipe's busy fraction (32 cycles x MFMAs per SIMD / loop cycles) and the in-kernel clock are reported per arm.
    Timed like the kernels (hipEvents on the launching stream, after a pre-roll), with the same clock / power telemetry."""
    import ctypes as C
    import numpy as np
    from torchnmf_amd import _capi
    lib = eng.be.lib
    st = eng.step_h
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    f16 = int(eng.precision in (_capi.PREC_F16, getattr(_capi, 'PREC_F16X', -1)))
    img, xp = eng.fW.p1_hi, st.xp
    kib = {128: 4, 256: 2, 64: 8}.get(eng.r_pad)
    if kib is None or img.numel() < 65536:
        return None
    waves = 8
    wrap = int(xp.numel() // (ncu * waves * kib * 1024))     # tiles per wave that one pass over the packed X holds
    if wrap < 8:
        return None
    sink = torch.empty(ncu * waves * 64, dtype=torch.float32, device=dev)
    stamps = torch.zeros(4 * ncu * waves, dtype=torch.int64, device=dev)

    def run(k, tiles):
        _capi.check(lib.nmfmu_ubench_mfma_hbm2(img.data_ptr(), img.numel(), f16, xp.data_ptr(), k, waves, tiles, wrap, ncu,
                                               sink.data_ptr(), stamps.data_ptr(), torch.cuda.current_stream().cuda_stream),
                    'nmfmu_ubench_mfma_hbm2')
    out = {'what': 'MFMA loop on the live W image fragments (32 per wave and tile, 8 waves per CU) beside an independent nt LDS-DMA '
                   f'stream of the live packed X at {kib} KiB per wave and tile = the MU step\'s flop : byte ratio; no LDS reads, '
                   'no VALU, no barriers (nmfmu_ubench_mfma_hbm2); long loops, in-kernel stamps',
           'grid': ncu, 'waves_per_workgroup': waves, 'one_pass_over_x_tiles': wrap}
    for name, k, tiles in (('with_stream', kib, max(wrap, 512)), ('mfma_only', 0, 1024)):
        flops = float(ncu) * waves * tiles * 32 * 32768.0
        preroll_steps(lambda: run(k, tiles), min(a.preroll_s, 0.3))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 60
        ev[0].record()
        for _ in range(reps):
            run(k, tiles)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        sv = stamps.cpu().numpy().reshape(ncu, waves, 4).astype(np.float64)
        clk = float(np.median(sv[:, :, 0] / np.maximum(sv[:, :, 1], 1.0))) * 100.0          # MHz inside the loops
        span = sv[:, :, 3].max(axis=1) - sv[:, :, 2].min(axis=1)                             # per workgroup, 100 MHz ticks
        span_cyc = float(np.median(span)) * clk / 100.0
        mfma_cyc = 32.0 * (waves // 4) * tiles * 32                                          # per SIMD
        ent = {'tiles': tiles, 'avg_launch_ms': round(ms, 5), 'tflops': round(flops / (ms * 1e-3) / 1e12, 1),
               'flops_per_launch': flops,
               'mfma_busy_frac': round(mfma_cyc / span_cyc, 4) if span_cyc > 0 else None,
               'mfma_busy_frac_of_launch': round(mfma_cyc / (ms * 1e-3 * clk * 1e6), 4) if clk > 0 else None,
               'clock_mhz_in_kernel': round(clk, 1),
               'loop_us_median_workgroup': round(float(np.median(span)) * 0.01, 1)}
        if k:
            ent['stream_gbs'] = round(ncu * waves * tiles * k * 1024 / (ms * 1e-3) / 1e9, 1)
        if a.telemetry_s > 0:
            tel = SmiSampler(dev.index or 0).under_load(lambda: run(k, tiles), min(a.telemetry_s, 0.5))
            ent['clock_mhz'], ent['power_w'] = tel.get('clock_mhz'), tel.get('power_w')
        out[name] = ent
    return out


def parity_leg(a, V, W0, H0, Vc, beta, precision, Wr, Hr, k, dev, gram=False):
    """k MU iterations from the same (V, W0, H0) as the CPU reference leg, compared factor by factor."""
    from torchnmf_amd.engine import DenseMU
    N, C = V.shape
    W, H = W0.clone(), H0.clone()
    eng = DenseMU(V, W, H, beta, precision=precision, allow_gram=gram)
    for _ in range(k):
        eng.w_step()
        eng.h_step()
    torch.cuda.synchronize()
    loss = eng.divergence()
    Wc, Hc = W.cpu(), H.cpu()
    rel = lambda x, y: float((x - y).norm() / y.norm())
    # reconstruction on a 512-row slice (the full 4096 x 65536 product is 1 GiB on either side)
    rows = slice(0, min(N, 512))
    rec, rec_r = Hc[rows] @ Wc.t(), Hr[rows] @ Wr.t()
    from oracle import mu_oracle as O
    # reference loss accumulated in float64 over 512-row blocks: an fp32 sum over 2.7e8 terms wanders by ~1e-4..1e-3
    # with the host's thread count (seen: the same GPU value 8e-6 off on one box, 5e-4 on another)
    loss_r = None
    if N * C <= 4096 * 65536:
        loss_r = 0.0
        for r0 in range(0, N, 512):
            loss_r += float(O.beta_div((Hr[r0:r0 + 512] @ Wr.t()).double(), Vc[r0:r0 + 512].double(), beta))
    d = {'rel_W': rel(Wc, Wr), 'rel_H': rel(Hc, Hr), 'rel_recon': rel(rec, rec_r),
         'rel_loss': (abs(loss - loss_r) / abs(loss_r)) if loss_r else None}
    d = {kk: (None if v is None else float(f'{v:.3e}')) for kk, v in d.items()}
    d['meets_1e-4'] = all(v is not None and v < 1e-4 for v in (d['rel_W'], d['rel_H'], d['rel_recon']))
    del eng
    return d


def fit_leg(a, V, W0, H0, beta, precision, dev, engine_ms, cls_name='NMF'):
    """What the user calls: NMF.fit(V, beta, tol -> never stops, max_iter = 200) end to end -- engine
    construction (packing V twice, validation), the initial loss, 200 iterations with the 20 loss evaluations + host syncs of
    nmf.py:393-407 -- wall clock, next to the engine-step figure of the headline."""
    from torchnmf_amd import nmf as _nmf
    m = getattr(_nmf, cls_name)(W=W0, H=H0).to(dev)

    def run(max_iter):
        m.W.data.copy_(W0)
        m.H.data.copy_(H0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = m.fit(V, beta=beta, tol=-1e9, max_iter=max_iter, **({} if precision is None else {'precision': precision}))
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0), n
    run(20)                                        # warm: allocator, kernels' first launches
    setup = min(run(0)[0] for _ in range(3))       # max_iter = 0: everything fit() does outside its loop
    walls = []
    for _ in range(3):
        w, n = run(200)
        assert n == 200
        walls.append(w)
    wall = sorted(walls)[1]
    loop = wall - setup
    return {'call': f"{cls_name}.fit(V, beta={beta:g}, tol=-1e9, max_iter=200" + ("" if precision is None else f", precision='{precision}'") + ")",
            'iterations': 200,
            'wall_ms': round(wall, 3), 'wall_ms_runs': [round(x, 3) for x in walls], 'setup_ms': round(setup, 3),
            'iters_per_s_whole_call': round(200e3 / wall, 1), 'iters_per_s_loop': round(200e3 / loop, 1),
            'ms_per_iter_loop': round(loop / 200, 4), 'engine_step_ms': round(engine_ms, 4),
            'loop_over_engine_step': round(loop / 200 / engine_ms, 4),
            'note': 'loop = 200 MU iterations + the 19 loss checkpoints of nmf.py:393-407 (no host sync; beta == 1 on the ping-pong '
                    'kernel: the loss rides in the W half-step behind the checkpoint, otherwise one loss pass each) in ONE cold call, '
                    'against engine_step_ms from long pre-rolled blocks; setup = packing V in both orientations, validation '
                    'flags, precision admission test (one kernel pass), initial loss'}


def ref_notebook_leg(a, dev, do_cpu):
    """The ONE workload the reference publishes numbers for (examples/benchmarks/benchmark.ipynb cells 3-4): NMF.fit on a
    5168 x 1025 magnitude spectrogram, 88 components, beta in {0, 0.5, 1, 1.5, 2}, max_iter = 60, tol = 1e-4, timed as the
    notebook times it -- wall clock of the whole fit() call divided by the iterations it returns.  The notebook's audio file
    is not available (no network): the target here is the magnitude of complex white noise of that shape (Rayleigh
    distributed, what |STFT| of noise is), same init law as the reference (|N(0,1)|, nmf.py:221).  Reported per beta: what
    precision='auto' picks and its s/iteration, the same fit forced to the 1x fp16 mode 'f16x' (what 'auto' would need the
    F16_MIN_DIM rule relaxed for), both with their relative error against the reference's op sequence on the host after the
    same number of iterations -- so the record says whether the 1x mode would be admissible at this shape.  The notebook's
    own figures (RTX 3070 / i7-4790K, torchnmf 0.3.4) are quoted as context, not as a baseline (other hardware, real audio)."""
    from torchnmf_amd.nmf import NMF
    from torchnmf_amd.engine import DenseMU
    N, C, R = 5168, 1025, 88
    g = torch.Generator(device=dev).manual_seed(5168)
    V = torch.randn(2, N, C, device=dev, generator=g).pow_(2).sum(0).sqrt_()
    W0 = torch.randn(C, R, device=dev, generator=g).abs_()
    H0 = torch.randn(N, R, device=dev, generator=g).abs_()
    notebook = {'0': (0.2972, 0.2081, 0.001958), '0.5': (0.4571, 0.2477, 0.002170), '1': (0.1713, 0.1546, 0.001306),
                '1.5': (0.3474, 0.2535, 0.002192), '2': (0.03827, 0.08189, 0.001327)}
    m = NMF(W=W0, H=H0).to(dev)

    def run(beta, precision, max_iter=60, tol=1e-4):
        m.W.data.copy_(W0)
        m.H.data.copy_(H0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = m.fit(V, beta=beta, tol=tol, max_iter=max_iter, **({} if precision is None else {'precision': precision}))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / max(n, 1), n
    out = {'workload': f'NMF.fit, {N}x{C} (frames x bins), rank {R}, max_iter=60, tol=1e-4, time / returned iterations '
                       '(/root/reference/examples/benchmarks/benchmark.ipynb:108-130)',
           'target': 'synthetic |complex white noise| (Rayleigh), fp32 -- the notebook\'s MAPS recording is not available',
           'f16_min_dim_rule': f"'auto' admits the 1x fp16 modes from min(V.shape) >= {DenseMU.F16_MIN_DIM} (engine.py); here "
                               f"min(V.shape) = {C}", 'betas': {}}
    Vc = W0c = H0c = None
    if do_cpu:
        from oracle import aten_port
        torch.set_flush_denormal(True)
        Vc, W0c, H0c = V.cpu(), W0.cpu(), H0.cpu()
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    for beta in (0.0, 0.5, 1.0, 1.5, 2.0):
        ent = {}
        for tag, prec in (('auto', None), ('f16x_forced', 'f16x')):
            run(beta, prec, 10)                        # warm (allocator, first launches of this beta's kernels)
            s_it, n = min(run(beta, prec) for _ in range(3))
            e = {'s_per_iter': float(f'{s_it:.4g}'), 'iters_per_s': round(1.0 / s_it, 1), 'n_iter': n,
                 'precision': m.last_precision}
            if do_cpu:
                # parity over the iterations this fit ran (no early stop on either side: tol -> never)
                k = 60
                run(beta, prec, k, -1e9)
                Wg, Hg = m.W.data.cpu(), m.H.data.cpu()
                if 'ref' not in ent:
                    t0 = time.perf_counter()
                    ent['ref'] = aten_port.mu_iterations(Vc, W0c, H0c, beta, k)
                    ent['cpu_port_s_per_iter'] = float(f'{(time.perf_counter() - t0) / k:.4g}')
                Wr, Hr = ent['ref'][0], ent['ref'][1]
                e['parity_k60'] = {'rel_W': float(f'{rel(Wg, Wr):.3e}'), 'rel_H': float(f'{rel(Hg, Hr):.3e}')}
                e['parity_k60']['meets_1e-4'] = max(e['parity_k60']['rel_W'], e['parity_k60']['rel_H']) < 1e-4
            ent[tag] = e
        ent.pop('ref', None)
        sk, tcpu, tcuda = notebook[f'{beta:g}']
        ent['notebook_context_s_per_iter'] = {'sklearn_i7_4790K': sk, 'torchnmf_cpu_i7_4790K': tcpu, 'torchnmf_cuda_rtx3070': tcuda}
        out['betas'][f'{beta:g}'] = ent
    if do_cpu:
        out['cpu_port_threads'] = torch.get_num_threads()
    return out


DTYPE_NAME = {'bf16': 'bf16', 'f16': 'f16', 'bf16x3': 'bf16x3 (split bf16, fp32-grade)', 'f16x': 'f16 operands, f32 target',
              'f16r': 'f16 operands, 3-byte target (top 24 bits of the fp32)'}
DTYPE_LONG = {'f16': 'f16 operands and target / fp32 accumulate (same MFMA rate as bf16; meets the 1e-4 parity bar)',
              'f16x': 'f16 operands, fp32 target / fp32 accumulate (1x MFMA work, twice the V stream; for targets fp16 does not hold exactly)',
              'f16r': 'f16 operands, 3-byte target (the fp32 rounded to its top 24 bits: 16 significant bits, fp32 range) / '
                      'fp32 accumulate (1x MFMA work, 1.5x the V stream; what auto takes for targets fp16 does not hold exactly, beta != 2)',
              'bf16': 'bf16 operands and target / fp32 accumulate (the type configs[1] names; factors ~2e-4 after 3 iterations)',
              'bf16x3': 'split bf16 (3 MFMAs per product, fp32 target): fp32-grade'}


def _spawned(local_rank, nprocs, port, argv):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(nprocs), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.argv = [sys.argv[0]] + list(argv)
    main()


def main():
    global STANDIN
    a = parse()
    STANDIN = bool(a.standin)
    if STANDIN:
        # TEST ONLY: the host logic of this script over gloo on CPU tensors; compute by the test-suite's stand-in backend
        tdir = os.path.join(ROOT, 'tests')
        if tdir not in sys.path:
            sys.path.insert(0, tdir)
        from cpu_backend import OracleBackend
        from torchnmf_amd import engine as _engine
        _engine.DEFAULT_BACKEND_FACTORY = OracleBackend
        torch.set_num_threads(1)
        assert a.workload == 'nmf', '--standin covers the dense NMF path'
    if a.precision is None:
        # (betamu: the optimizer's own default -- 'auto' resolves like NMF.fit's since round 5)
        a.precision = 'f16' if a.workload in ('nmf', 'nmfd') else ('auto' if a.workload == 'betamu' else 'bf16')
    if a.workload == 'plca':
        assert int(os.environ.get('WORLD_SIZE', '1')) == 1, 'PLCA is not sharded'
        torch.cuda.set_device(0)
        return main_plca(a)
    if a.workload == 'sparse':
        assert int(os.environ.get('WORLD_SIZE', '1')) == 1, 'the sparse path is not sharded'
        torch.cuda.set_device(0)
        return main_sparse(a)
    if a.workload in ('nmfd', 'nmf2d'):
        assert int(os.environ.get('WORLD_SIZE', '1')) == 1, 'NMFD / NMF2D are not sharded (replicas only)'
        torch.cuda.set_device(0)
        return main_nmfd(a)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N` (no launcher): spawn the N ranks ourselves, one per GPU, rendezvous on localhost
        import socket
        import torch.multiprocessing as mp
        assert STANDIN or torch.cuda.device_count() >= a.gpus, f'--gpus {a.gpus} but {torch.cuda.device_count()} devices are visible'
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
        mp.spawn(_spawned, args=(a.gpus, port, sys.argv[1:]), nprocs=a.gpus, join=True)
        return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert STANDIN or torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks')
    if STANDIN:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    group = None
    nranks = 1
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.force_dist and 'RANK' not in os.environ:          # plain `python bench.py --force-dist`: a world of one
            os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
            os.environ.setdefault('MASTER_PORT', '29533')
        if STANDIN:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)   # RCCL
        group = dist.group.WORLD
        nranks = dist.get_world_size()                   # as RCCL's communicator reports it

    # ---- workload preset.  1 GPU: BASELINE configs[1].  N > 1 GPUs: configs[4], the config the ">= 6x at 8 GPUs"
    # target is quoted on -- every rank owns an 8192 x 262144 column shard at rank 256 (weak scaling).  Explicit
    # --rows / --cols / --rank override the preset.
    preset = a.config or ('cfg5' if world > 1 else 'cfg1')
    pr = {'cfg1': (4096, 65536, 128), 'cfg5': (8192, 262144, 256), 'cfg5full': (8192, 2097152, 256)}[preset]
    if STANDIN:
        pr = (64, 96, 8)        # (the CPU oracle computes these iterations: sizes of a unit test)
    N = a.rows if a.rows is not None else pr[0]
    C = a.cols if a.cols is not None else pr[1]
    R = a.rank if a.rank is not None else pr[2]
    beta = a.beta
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    if N * C > (1 << 32):
        # (cfg5full: 64 GiB of fp32 -- drawn in column chunks so that the temporaries stay small; same law)
        V = torch.empty(N, C, device=dev)
        for c0 in range(0, C, 65536):
            V[:, c0:c0 + 65536] = torch.rand(N, min(65536, C - c0), device=dev, generator=g).bfloat16().float()
    else:
        V = torch.rand(N, C, device=dev, generator=g).bfloat16().float()   # bf16-representable U[0,1): exact in fp16 too
    if beta <= 0:
        V.clamp_(min=2.0 ** -7)                                        # strictly positive for beta <= 0 (nmf.py:332-336)
    gw = torch.Generator(device=dev).manual_seed(2000 + rank)
    W0 = torch.randn(C, R, device=dev, generator=gw).abs_()            # the reference's init law (nmf.py:221)
    gh = torch.Generator(device=dev).manual_seed(3000)
    H0 = torch.randn(N, R, device=dev, generator=gh).abs_()            # replicated
    do_cpu = rank == 0 and world == 1 and a.cpu_iters > 0
    betamu = a.workload == 'betamu'
    flops_per_iter_gpu = (8.0 if beta == 1 else 12.0) * N * C * R

    head = dense_leg(a, V, W0, H0, beta, a.precision, group, world, dev, not a.no_roofline, betamu, telemetry=True,
                     gram=a.gram and beta == 2 and world == 1)
    # ---- N > 1: the 1-GPU denominator of THIS workload, measured in THIS run (the default N = 1 line is
    # configs[1] at rank 128, the N > 1 lines run configs[4]'s shard at rank 256 -- a curve through the default lines would
    # divide two different workloads).  Rank 0 times its own shard unsharded (no collective, the fused apply instead of
    # slab reduction + all-reduce + apply) while the other ranks wait at a barrier.
    same1 = None
    if world > 1 and (group is not None):
        import torch.distributed as dist
        if rank == 0:
            one = dense_leg(a, V, W0, H0, beta, a.precision, None, 1, dev, False, blocks_min=3)
            same1 = {'what': f"rank 0's own shard ({N}x{C} rank {R}) as an unsharded 1-GPU problem, same precision, timed in this run "
                             'between the sharded legs (the other ranks idle at a barrier)',
                     'ms_per_step': round(one['ms_per_step'], 4), 'iters_per_s': round(1e3 / one['ms_per_step'], 2),
                     'value': round(one['gflops'], 1), 'unit': 'GFLOP/s', 'blocks_ms_per_step': one['blocks_ms_per_step']}
            del one
        dist.barrier()
    asked = a.precision
    if a.precision == 'auto':            # betamu: report what the optimizer's 'auto' resolved to
        a.precision = head['eng'].precision_name
    # secondary, clearly labelled object: the other single-plane operand type, timed in the same run
    other = {'f16': 'bf16', 'bf16': 'f16'}.get(a.precision)
    second = None
    if other and not betamu and not a.no_parity_mode and world == 1:
        pm = dense_leg(a, V, W0, H0, beta, other, group, world, dev, not a.no_roofline)
        second = {'precision': other, 'dtype': DTYPE_LONG[other], 'value': round(pm['gflops'], 1), 'unit': 'GFLOP/s',
                  'iters_per_s': round(1e3 / pm['ms_per_step'], 2), 'ms_per_step': round(pm['ms_per_step'], 4),
                  'blocks_ms_per_step': pm['blocks_ms_per_step'], 'roofline': pm.get('roofline')}
        del pm

    cpu = None
    parity = None
    Vc = None
    if do_cpu:
        Vc, W0c, H0c = V.cpu(), W0.cpu(), H0.cpu()
        dt, cores, tried, Wr, Hr = cpu_baseline(Vc, W0c, H0c, beta, a.cpu_iters, betamu)
        cpu = {'value': round(flops_per_iter_gpu / dt / 1e9, 2), 'unit': 'GFLOP/s', 'cores': cores, 'kind': 'port',
               'host_cores': usable_cores(), 'thread_probe_s': tried,
               'iters_per_s': round(1.0 / dt, 4), 's_per_iter': round(dt, 4),
               'sample': f'{a.cpu_iters} timed MU iterations (+1 warm-up) of the same {N}x{C} rank-{R} beta={beta:g} '
                         f'workload, fp32, reference op sequence (oracle/aten_port.py), loss evaluation excluded'}
        if not betamu:
            modes = [a.precision] + ([other] if second is not None else [])
            parity = {'k': a.cpu_iters, 'reference': 'oracle/aten_port.py (fp32, same V, W0, H0), the timed CPU iterations',
                      'bar': 1e-4, 'modes': {m: parity_leg(a, V, W0, H0, Vc, beta, m, Wr, Hr, a.cpu_iters, dev) for m in modes}}

    # ---- the other BASELINE configs the driver's default command should see (1 GPU, default workload only):
    # configs[2] = the beta sweep at this shape, configs[3] = NMFD.  Each with its own in-run parity check (k = --cpu-iters, like the headline).
    beta_sweep = None
    nmfd = None
    nmf2d = None
    default_run = (world == 1 and not betamu and not a.no_sweep and beta == 1 and (N, C, R) == (4096, 65536, 128)
                   and a.precision in ('f16', 'bf16'))
    if default_run:
        from oracle import aten_port
        beta_sweep = {'shape': f'{N}x{C} rank={R}', 'precision': a.precision,
                      'flops_per_iteration': '12*N*C*R (6 contractions, as the reference computes them)', 'betas': {}}
        for b in (2.0, 0.5, 0.0):
            Vb = V.clamp(min=2.0 ** -7) if b <= 0 else V
            leg = dense_leg(a, Vb, W0, H0, b, a.precision, None, 1, dev, True, blocks_min=3, telemetry=True)
            ent = {'iters_per_s': round(1e3 / leg['ms_per_step'], 2), 'ms_per_step': round(leg['ms_per_step'], 4),
                   'value': round(leg['gflops'], 1), 'unit': 'GFLOP/s', 'blocks_ms_per_step': leg['blocks_ms_per_step'],
                   'kernel': leg['roofline']['kernel'], 'kernel_frac': leg['roofline']['frac'],
                   'kernel_avg_launch_ms': leg['roofline']['avg_launch_ms'], 'kernel_tflops': leg['roofline']['achieved'],
                   'clock_mhz': leg['roofline'].get('clock_mhz'), 'power_w': leg['roofline'].get('power_w')}
            del leg
            gleg = None
            if b == 2.0:
                # what fit() runs for beta == 2 (round 4): no reconstruction -- numerator X @ panel (one streaming MFMA GEMM),
                # denominator owner @ (panel^T panel).  4 N C R flops executed per iteration; HBM-bound on the X stream, so it
                # is priced against the HBM roofline -- the 12 N C R kernel above stays the MFMA-roofline line of configs[2].
                gleg = dense_leg(a, Vb, W0, H0, b, a.precision, None, 1, dev, True, blocks_min=3, gram=True, telemetry=True)
                rf = gleg['roofline']
                ent['gram_path'] = {
                    'what': 'NMF.fit(beta=2): num = X @ panel (kModeXB, one pass over X), den = owner @ (panel^T panel) via the MFMA '
                            'Gram kernel; no N x C reconstruction (nmf.py:61-63 puts no eps inside its grad_outputs)',
                    'iters_per_s': round(1e3 / gleg['ms_per_step'], 2), 'ms_per_step': round(gleg['ms_per_step'], 4),
                    'blocks_ms_per_step': gleg['blocks_ms_per_step'],
                    'flops_executed_per_iteration': '4*N*C*R (two X @ panel GEMMs) + 4*(N+C)*R*R (Gram matrices and owner @ G)',
                    'kernel': 'nmfmu::fused_kernel<.., kEuc, .., kModeXB>', 'kernel_avg_launch_ms': rf['avg_launch_ms'],
                    'kernel_avg_launch_ms_w_step': rf['avg_launch_ms_w_step'], 'kernel_avg_launch_ms_h_step': rf['avg_launch_ms_h_step'],
                    'roofline': {'bound': 'hbm', 'achieved': rf['hbm']['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                 'frac': rf['hbm']['frac'], 'algorithmic_bytes_per_launch': rf['hbm']['algorithmic_bytes_per_launch'],
                                 'traffic': pmc_traffic_key(f'{N}x{C}_r{R}_{a.precision}_xb')[0],
                                 'traffic_source': pmc_traffic_key(f'{N}x{C}_r{R}_{a.precision}_xb')[1]},
                    'speedup_over_12NCR_kernel': round(ent['ms_per_step'] / gleg['ms_per_step'], 3),
                    'clock_mhz': rf.get('clock_mhz'), 'power_w': rf.get('power_w')}
                del gleg
            if do_cpu:
                k = a.cpu_iters        # round 5: the sweep legs carry the same k as the headline (was 3)
                Vbc = Vc.clamp(min=2.0 ** -7) if b <= 0 else Vc
                Wr, Hr = aten_port.mu_iterations(Vbc, W0.cpu(), H0.cpu(), b, k)
                ent['parity'] = dict(k=k, **parity_leg(a, Vb, W0, H0, Vbc, b, a.precision, Wr, Hr, k, dev))
                if b == 2.0:
                    ent['gram_path']['parity'] = dict(k=k, **parity_leg(a, Vb, W0, H0, Vbc, b, a.precision, Wr, Hr, k, dev, gram=True))
                del Vbc
            del Vb
            beta_sweep['betas'][f'{b:g}'] = ent
        import argparse
        na = argparse.Namespace(**vars(a))
        na.workload, na.precision, na.rows, na.cols, na.rank, na.beta, na.taps = 'nmfd', 'f16', None, None, None, 1.0, 400
        if not do_cpu:
            na.cpu_iters = 0
        line = nmfd_line(na, sub=True)
        nmfd = {k: line[k] for k in ('metric', 'value', 'unit', 'iters_per_s', 'ms_per_step', 'blocks_ms_per_step', 'dtype',
                                     'roofline', 'parity', 'cpu_baseline', 'fit')}
        # row f2 (no reference headline): NMF2D on a 64-channel 256 x 512 frame, rank 8, 8 x 16 kernel, in the
        # mode fit() picks there (fp16 operands: every contraction has >= 1024 terms); its in-run parity covers bf16 as well
        na = argparse.Namespace(**vars(a))
        na.workload, na.precision, na.rows, na.cols, na.rank, na.beta = 'nmf2d', 'auto', None, None, None, 1.0
        na.telemetry_s = 0
        if not do_cpu:
            na.cpu_iters = 0
        line = nmfd_line(na, sub=True)
        nmf2d = {k: line[k] for k in ('metric', 'value', 'unit', 'iters_per_s', 'ms_per_step', 'blocks_ms_per_step', 'dtype',
                                      'roofline', 'parity', 'cpu_baseline', 'fit')}

    # ---- fit(): the call a torchnmf user makes, end to end; real_data_mode: a target fp16 does NOT hold exactly
    fit_obj, real = None, None
    if default_run:
        from torchnmf_amd.engine import DenseMU
        fit_obj = fit_leg(a, V, W0, H0, beta, a.precision, dev, head['ms_per_step'])
        gr = torch.Generator(device=dev).manual_seed(4000)
        Vr = torch.rand(N, C, device=dev, generator=gr)             # plain fp32 U[0,1): fp16 would round it
        be = head['eng'].be
        # what fit()'s 'auto' resolves to on this target: ask an engine (one admission test; 'f16r' since round 6)
        picks = DenseMU(Vr, W0.clone(), H0.clone(), beta, precision='auto', allow_f16=True).precision_name
        leg = dense_leg(a, Vr, W0, H0, beta, picks, None, 1, dev, True, blocks_min=3)
        rf = leg['roofline']
        real = {'target': 'plain fp32 U[0,1) (not exactly representable in fp16)', 'auto_picks': picks, 'dtype': DTYPE_LONG[picks],
                'iters_per_s': round(1e3 / leg['ms_per_step'], 2), 'ms_per_step': round(leg['ms_per_step'], 4),
                'value': round(leg['gflops'], 1), 'unit': 'GFLOP/s', 'blocks_ms_per_step': leg['blocks_ms_per_step'],
                'kernel': rf['kernel'], 'kernel_avg_launch_ms': rf['avg_launch_ms'], 'kernel_frac_mfma': rf['frac'],
                'hbm': rf['hbm']}
        del leg
        if do_cpu:
            from oracle import aten_port
            k = a.cpu_iters
            Vrc = Vr.cpu()
            Wr, Hr = aten_port.mu_iterations(Vrc, W0.cpu(), H0.cpu(), beta, k)
            real['parity'] = dict(k=k, **parity_leg(a, Vr, W0, H0, Vrc, beta, picks, Wr, Hr, k, dev))
            del Vrc
        fitr = fit_leg(a, Vr, W0, H0, beta, None, dev, 1e3 / real['iters_per_s'])   # precision=None: fit()'s own default ('auto')
        real['fit'] = fitr
        del Vr

    ref_nb = None
    if default_run or a.ref_notebook:
        try:
            ref_nb = ref_notebook_leg(a, dev, do_cpu)
        except Exception as ex:              # (context leg: report the failure instead of losing the whole line)
            ref_nb = {'error': f'{type(ex).__name__}: {ex}'[:300]}

    if rank == 0:
        ms_per_step = head['ms_per_step']
        eng = head['eng']
        out = {
            'metric': f'MU GFLOP/s (algorithmic {"8" if beta == 1 else "12"}*N*C*R per iteration), dense NMF '
                      f'{N}x{C} rank-{R} beta={beta:g}; MU iterations/s alongside',
            'value': round(head['gflops'], 1), 'unit': 'GFLOP/s', 'iters_per_s': round(1e3 / ms_per_step, 2),
            'n_gpus': world, 'nranks': nranks, 'steps': a.steps, 'warmup': a.warmup, 'preroll_steps': head['preroll_steps'],
            'ms_per_step': round(ms_per_step, 4),
            'repeats': len(head['blocks_ms_per_step']), 'blocks_ms_per_step': head['blocks_ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': DTYPE_NAME[a.precision], 'dtype_note': DTYPE_LONG[a.precision],
            'data': 'synthetic' if not STANDIN else 'synthetic -- TEST STAND-IN (--standin: CPU tensors, gloo, oracle-backed backend): NOT a measurement',
            'config': {'workload': (f'NMF {N}x{C * world} rank={R} beta={beta:g}, V column-sharded {world} x {C}, '
                                    f'H replicated, all-reduce of the H numerators per iteration' + (' (BASELINE configs[4])' if (N, C * world, R, beta) == (8192, 2097152, 256, 1.0) else ''))
                       if world > 1 else
                       ('trainer.BetaMu.step on ' if betamu else '') + f'NMF {N}x{C} rank={R} beta={beta:g}' +
                       (' (BASELINE configs[1])' if (N, C, R, beta) == (4096, 65536, 128, 1.0) else ''),
                       'preset': preset, 'rows': N, 'cols_per_gpu': C, 'rank': R, 'beta': beta, 'precision': a.precision,
                       'parallelism': f'column-shard x{world}' if world > 1 else 'single GPU',
                       'nsplit_h': eng.step_h.nsplit, 'nsplit_w': eng.step_w.nsplit, 'block_rows_h': eng.step_h.block_rows, 'block_rows_w': eng.step_w.block_rows,
                       'launch': 'eager launches'},
            'roofline': head.get('roofline'), 'cpu_baseline': cpu, 'parity': parity,
            ('bf16_mode' if other == 'bf16' else 'parity_mode'): second,
            'beta_sweep': beta_sweep, 'nmfd': nmfd, 'nmf2d': nmf2d, 'fit': fit_obj, 'real_data_mode': real,
            'ref_notebook': ref_nb,
        }
        if same1 is not None:
            out['same_shard_1gpu'] = same1
            out['efficiency'] = round(same1['ms_per_step'] / ms_per_step, 4)
            out['scaling_note'] = ('weak scaling: every rank owns one shard of this size; efficiency = same_shard_1gpu.ms_per_step / '
                                   'ms_per_step (1.0 = the all-reduce and the unfused apply cost nothing).  NOTE the default N = 1 line '
                                   f'of this script is another workload (configs[1], 4096x65536 rank 128); the 1-GPU point of THIS '
                                   f'curve is same_shard_1gpu (or `--gpus 1 --config {preset}`)')
        if betamu:
            out['config']['precision_asked'] = asked
            out['config']['closure'] = 'returns m() (reconstruction materialised)' if a.materialise else \
                'returns the layer (deferred reconstruction)'
        print(json.dumps(out))
    if group is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
