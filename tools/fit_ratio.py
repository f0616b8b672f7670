"""fit() wall time per iteration with one of the round-6 loss-checkpoint switches on / off, configs[1] shape:
    python tools/fit_ratio.py [--iters 200] [--toggle TORCHNMF_AMD_RIDING_LOSS | TORCHNMF_AMD_FUSED_CHECKPOINT]
Prints, per target kind (fp16-exact -> 'f16', plain floats -> 'f16r') and per mode, the median of three whole calls."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-nmf_amd'))
from torchnmf_amd.nmf import NMF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=200)
ap.add_argument('--beta', type=float, default=1.0)
ap.add_argument('--toggle', default='TORCHNMF_AMD_RIDING_LOSS')
ap.add_argument('--auto', action='store_true', help="one pass per target with precision='auto' (the admission test runs): whole call and loop")
a = ap.parse_args()
dev = torch.device('cuda:0')
N, C, R = 4096, 65536, 128
g = torch.Generator(device=dev).manual_seed(1)
Vf = torch.rand(N, C, device=dev, generator=g)
targets = {'exact': Vf.half().float(), 'plain': Vf}
W0 = torch.rand(C, R, device=dev, generator=g) + 0.1
H0 = torch.rand(N, R, device=dev, generator=g) + 0.1
for kind, V in targets.items():
    for rep in range(1 if a.auto else 2):
        for mode in (('1',) if a.auto else ('1', '0')):
            os.environ[a.toggle] = mode
            ts = []
            for _ in range(3):
                m = NMF(W=W0.clone(), H=H0.clone()).to(dev)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = m.fit(V, a.beta, -1e9, a.iters)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            print(f'{kind} ({m.last_precision}) {a.toggle}={mode}: whole call {ts[1]:.2f} ms for {n} iterations = {ts[1] / n:.4f} ms/iter', flush=True)
