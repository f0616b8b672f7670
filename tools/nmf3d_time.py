"""MU iterations per second of NMF3D (1 x 32 x 64 x 64 x 64 volume, rank 8, 4 x 4 x 8 kernel, beta = 1), engine level, in the
mode fit() picks ('auto') and in split bf16; plus explicit operands + store-then-fold for comparison
(TORCHNMF_AMD_NMFD_EXPLICIT=1 TORCHNMF_AMD_NMFD_H_ROWS=0 TORCHNMF_AMD_NMFD_KSPLIT=0).  Usage: python tools/nmf3d_time.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch-nmf_amd'))
from torchnmf_amd.nmfd_engine import ConvMU  # noqa: E402

dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(6)
ls, ks, Cc, R = (64, 64, 64), (4, 4, 8), 32, 8
V = torch.rand(1, Cc, *ls, device=dev, generator=g) + 1e-3
out = {}
for prec in os.environ.get('PRECISIONS', 'auto,bf16x3').split(','):
    W = torch.randn(Cc, R, *ks, device=dev, generator=g).abs_()
    H = torch.randn(1, R, *[l - k + 1 for l, k in zip(ls, ks)], device=dev, generator=g).abs_()
    eng = ConvMU(V, W, H, 1.0, precision=prec)
    for _ in range(5):
        eng.w_step(); eng.h_step()
    torch.cuda.synchronize()
    n = int(os.environ.get('STEPS', '40'))
    t0 = time.perf_counter()
    for _ in range(n):
        eng.w_step(); eng.h_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out[prec] = {'precision': eng.precision_name, 'iters_per_s': round(1 / dt, 1), 'ms_per_iter': round(1e3 * dt, 4),
                 'implicit': bool(eng.implicit), 'h_rows': bool(eng.h_rows), 'fold': getattr(eng, 'wk_fold', None),
                 'w_ksplit': eng.w_ksplit, 'c_rows': eng.c_rows,
                 'finite': bool(torch.isfinite(W).all() and torch.isfinite(H).all())}
    del eng
print(json.dumps({'workload': 'NMF3D 1x32x64x64x64 rank 8 kernel 4x4x8 beta=1', 'modes': out}))
