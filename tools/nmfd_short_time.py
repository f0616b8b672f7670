"""MU iterations per second of NMFD with a short kernel (1 x 513 x 4096 spectrogram, rank 16, T = 32, beta = 1), engine level:
the window-operand H numerator + general contraction split of round 4 against the store-then-fold path
(TORCHNMF_AMD_NMFD_H_ROWS=0 TORCHNMF_AMD_NMFD_KSPLIT=0).  Usage: python tools/nmfd_short_time.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch-nmf_amd'))
from torchnmf_amd.nmfd_engine import ConvMU  # noqa: E402

dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(7)
Cc, L, R, T = 513, 4096, 16, 32
V = torch.rand(1, Cc, L, device=dev, generator=g) + 1e-3
out = {}
for prec in os.environ.get('PRECISIONS', 'auto,bf16x3,bf16').split(','):
    W = torch.randn(Cc, R, T, device=dev, generator=g).abs_()
    H = torch.randn(1, R, L - T + 1, device=dev, generator=g).abs_()
    eng = ConvMU(V, W, H, 1.0, precision=prec)
    for _ in range(5):
        eng.w_step(); eng.h_step()
    torch.cuda.synchronize()
    n = int(os.environ.get('STEPS', '60'))
    t0 = time.perf_counter()
    for _ in range(n):
        eng.w_step(); eng.h_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out[prec] = {'precision': eng.precision_name, 'iters_per_s': round(1 / dt, 1), 'ms_per_iter': round(1e3 * dt, 4),
                 'implicit': bool(eng.implicit), 'h_rows': bool(eng.h_rows), 'fold': getattr(eng, 'wk_fold', None),
                 'w_ksplit': eng.w_ksplit, 'finite': bool(torch.isfinite(W).all() and torch.isfinite(H).all())}
    del eng
print(json.dumps({'workload': f'NMFD 1x{Cc}x{L} rank {R} T={T} beta=1', 'modes': out}))
