"""EM iterations per second of SIPLCA2 (row f4's shift-invariant member) on the NMF2D bench shape
(1 x 64 x 256 x 512, rank 8, 8 x 16 kernel), engine level: round-4 paths (window tables with several shift axes,
window-operand GEMM for G W, contraction-split G^T H) against the explicit-operand / store-then-fold path
(TORCHNMF_AMD_NMFD_EXPLICIT=1 TORCHNMF_AMD_NMFD_H_ROWS=0 TORCHNMF_AMD_NMFD_KSPLIT=0).  Usage: python tools/siplca2_time.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch-nmf_amd'))
from torchnmf_amd.plca import SIPLCA2  # noqa: E402

dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(5)
V = torch.rand(1, 64, 256, 512, device=dev, generator=g)
m = SIPLCA2(V.shape, rank=8, kernel_size=(8, 16)).to(dev)
em = m._make_em((V / V.sum()).contiguous(), os.environ.get('PRECISION', 'bf16x3'))
for _ in range(5):
    em.em_step(True, True, True, 1.0, 1.0, 1.0)
torch.cuda.synchronize()
n = int(os.environ.get('STEPS', '30'))
t0 = time.perf_counter()
for _ in range(n):
    em.em_step(True, True, True, 1.0, 1.0, 1.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
e = em.eng
print(json.dumps({'workload': 'SIPLCA2 1x64x256x512 rank 8 kernel 8x16', 'precision': e.precision_name,
                  'em_iters_per_s': round(1 / dt, 1), 'ms_per_iter': round(1e3 * dt, 4), 'implicit': bool(e.implicit),
                  'h_rows': bool(e.h_rows), 'w_ksplit': e.w_ksplit, 'c_rows': e.c_rows,
                  'finite': bool(torch.isfinite(m.W).all() and torch.isfinite(m.H).all())}))
