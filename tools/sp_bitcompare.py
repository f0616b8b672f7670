"""Bit-compare two builds of the library on the same dense MU iterations (round 6: nmfmu::sp_kernel against the four-wave
kernel it replaces at padded rank 256 -- the MFMA order per accumulator is the same, so the factors must be bit-identical).
    NMFMU_LIB=.../libnmfmu_nosp.so python tools/sp_bitcompare.py --save /tmp/ref.pt
    python tools/sp_bitcompare.py --compare /tmp/ref.pt
TORCHNMF_AMD_NSPLIT forces the same contraction split on both sides (the two kernels' own choices differ)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd')):
    sys.path.insert(0, p)
import torch

from torchnmf_amd.engine import DenseMU

ap = argparse.ArgumentParser()
ap.add_argument('--save')
ap.add_argument('--compare')
ap.add_argument('--shapes', default='8192x16384x256,300x65600x200,1000x3000x256')
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--beta', type=float, default=1.0)
a = ap.parse_args()
dev = torch.device('cuda', 0)
out = {}
for sh in a.shapes.split(','):
    N, C, R = (int(x) for x in sh.split('x'))
    g = torch.Generator().manual_seed(N + C + R)
    V = (torch.rand(N, C, generator=g).bfloat16().float() + (2.0 ** -7 if a.beta <= 0 else 0.0)).to(dev)
    W = torch.randn(C, R, generator=g).abs().to(dev)
    H = torch.randn(N, R, generator=g).abs().to(dev)
    eng = DenseMU(V, W, H, a.beta, precision='f16')
    for _ in range(a.iters):
        eng.w_step()
        eng.h_step()
    torch.cuda.synchronize()
    out[sh] = (W.cpu(), H.cpu(), eng.divergence(), (eng.step_w.nsplit, eng.step_h.nsplit))
    del eng
if a.save:
    torch.save(out, a.save)
    print('saved', {k: v[3] for k, v in out.items()})
if a.compare:
    ref = torch.load(a.compare)
    ok = True
    for sh, (W, H, loss, ns) in out.items():
        Wr, Hr, lr, nsr = ref[sh]
        eq = torch.equal(W, Wr) and torch.equal(H, Hr)
        rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
        nbad_w = int((W != Wr).any(dim=1).sum())
        nbad_h = int((H != Hr).any(dim=1).sum())
        print(f'{sh}: nsplit {ns} vs {nsr}: bit-identical={eq} relW={rel(W, Wr):.3e} relH={rel(H, Hr):.3e} loss {loss} vs {lr}; '
              f'rows of W / H that differ: {nbad_w} / {nbad_h}; max abs dW {float((W - Wr).abs().max()):.3e}')
        ok &= eq or ns != nsr
    sys.exit(0 if ok else 1)
