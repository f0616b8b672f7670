"""Per-workgroup timeline of the software-pipelined rank-256 kernel on configs[4]'s shard (round 6; VERDICT r5 item 1: "if it
lands below, commit the wave-timeline that shows where the gaps went").  The product build stamps, for every workgroup, the
constant 100 MHz clock at kernel entry, loop start, loop end and exit, and where it ran (nmfmu_step.stamps, layout in
include/nmfmu.h): this script runs the W half-step (2 048 workgroups of 128 tiles, fused apply in the epilogue: eight rounds of one
workgroup per CU) and the H half-step (256 workgroups of 1 024 tiles, slab stores) back to back and prints where a launch's
time goes: prologue / loop / epilogue per workgroup, the dead time between two workgroups on the same CU, the rounds.

    python tools/sp_timeline.py [--rows 8192 --cols 262144 --rank 256] [--iters 12] [--beta 1] [--precision f16]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-nmf_amd'))
from torchnmf_amd.engine import DenseMU  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=8192)
ap.add_argument('--cols', type=int, default=262144)
ap.add_argument('--rank', type=int, default=256)
ap.add_argument('--iters', type=int, default=12)
ap.add_argument('--beta', type=float, default=1.0)
ap.add_argument('--precision', default='f16')
a = ap.parse_args()
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(5)
V = torch.rand(a.rows, a.cols, device=dev, generator=g).half().float().clamp(min=2.0 ** -7 if a.beta <= 0 else 0.0)
W = torch.rand(a.cols, a.rank, device=dev, generator=g) + 0.1
H = torch.rand(a.rows, a.rank, device=dev, generator=g) + 0.1
eng = DenseMU(V, W, H, a.beta, precision=a.precision)
del V


def timeline(st, step, name):
    nwg = (st.owner.rows_pad // st.block_rows) * st.nsplit
    buf = torch.zeros(64 + 5 * nwg, dtype=torch.int64, device=dev)
    st.struct.stamps = buf.data_ptr()
    try:
        for _ in range(a.iters):
            eng.w_step(), eng.h_step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
    finally:
        st.struct.stamps = None
    v = buf[64:].cpu().numpy().reshape(nwg, 5)
    if not v.any():
        print(f'== {name}: this half-step\'s kernel records no stamps (only the ping-pong and the rank-256 pipelined kernel do)')
        return 0.0
    t = (v[:, :4] - v[:, 2].min()) / 100.0              # us since the first entry; columns: loop start, loop end, entry, exit
    start, end, entry, exit_ = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    where = v[:, 4]
    xcc, hw = (where >> 32) & 0xf, where & 0xffffffff
    cu = ((xcc << 16) | (((hw >> 13) & 0x7) << 8) | ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4)).astype(np.int64)   # XCC, SE, CU, SH
    print(f'== {name}: {nwg} workgroups, {len(np.unique(cu))} distinct CUs seen, launch {e0.elapsed_time(e1) * 1e3:.0f} us (hipEvents), '
          f'first entry -> last exit {exit_.max():.0f} us')
    q = lambda x: f'median {np.median(x):7.1f}  p10 {np.percentile(x, 10):7.1f}  p90 {np.percentile(x, 90):7.1f}  max {x.max():7.1f}'
    print('  prologue  (entry -> loop start) us:', q(start - entry))
    print('  tile loop (start -> end)        us:', q(end - start), f'  = {np.median(end - start) / max(1, st.panel.rows_pad // 64 // st.nsplit) * 1e3:.0f} ns per tile')
    print('  epilogue  (loop end -> exit)    us:', q(exit_ - end))
    gaps, rounds = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(entry[idx])]
        rounds.append(len(idx))
        gaps += list(entry[idx][1:] - exit_[idx][:-1])
    if gaps:
        gaps = np.array(gaps)
        print(f'  workgroups per CU: median {np.median(rounds):.0f} (min {min(rounds)}, max {max(rounds)});  exit -> next entry on the same CU us:', q(gaps))
        per_cu_busy = np.median(rounds) * np.median(end - start)
        print(f'  a CU\'s launch = {np.median(rounds):.0f} x (prologue {np.median(start - entry):.1f} + loop {np.median(end - start):.1f} + epilogue '
              f'{np.median(exit_ - end):.1f} + hand-over {np.median(gaps):.1f}) us; loops alone {per_cu_busy:.0f} us of {exit_.max():.0f}')
    # the rounds: entry times cluster when the chip runs in lock step
    order = np.sort(entry)
    k = max(1, len(order) // max(1, int(np.median(rounds)) if gaps is not None and len(rounds) else 1))
    marks = [order[i * k] for i in range(min(len(order) // k, 12))]
    print('  entry time of the first workgroup of each round us:', ' '.join(f'{m:.0f}' for m in marks))
    return exit_.max()


timeline(eng.step_w, eng.w_step, 'W half-step (fused apply in the epilogue)')
timeline(eng.step_h, eng.h_step, 'H half-step (slab stores)')
