"""Where do the NMFD GPU tests spend their time?  (They are ~95 % of the GPU suite's wall clock.)  cProfile of one engine
construction + two iterations + one loss at a mid-size test shape; run on the GPU box: python tools/nmfd_test_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch-nmf_amd'))
from torchnmf_amd.nmfd_engine import ConvMU  # noqa: E402

dev = torch.device('cuda', 0)
B, Cc, L, R, T = 1, 1025, 520, 3, 136
g = torch.Generator().manual_seed(0)
V = torch.rand(B, Cc, L, generator=g) + 1e-3
W0 = torch.randn(Cc, R, T, generator=g).abs()
H0 = torch.randn(B, R, L - T + 1, generator=g).abs()


def once(prec):
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = ConvMU(V.to(dev), W, H, 1, 0.01, 0.02, precision=prec)
    for _ in range(2):
        eng.w_step()
        eng.h_step()
    return eng.divergence()


for prec in ('bf16x3', 'bf16x3', 'f16', 'f16'):
    torch.cuda.synchronize()
    t0 = time.time()
    once(prec)
    torch.cuda.synchronize()
    print(prec, 'engine + 2 iterations + loss: %.3f s' % (time.time() - t0), flush=True)
pr = cProfile.Profile()
pr.enable()
once('bf16x3')
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
