"""Localise a fused-apply epilogue error: master, both images and column sums after one W half-step, then H."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
from test_layout_emulation import p1_offset, p2_offset
from torchnmf_amd.engine import DenseMU
from oracle import mu_oracle as O
dev = torch.device('cuda', 0)
N, C, R = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (200, 330, 24))]
g = torch.Generator().manual_seed(N + R)
V = torch.rand(N, C, generator=g)
W0 = torch.randn(C, R, generator=g).abs()
H0 = torch.randn(N, R, generator=g).abs()
for prec in ('bf16', 'f16'):
    dt = torch.bfloat16 if prec == 'bf16' else torch.float16
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = DenseMU(V.to(dev), W, H, 1.0, precision=prec)
    eng.w_step(); torch.cuda.synchronize()
    Wr = O.nmf_w_step(V, W0, H0, 1, 1.0)
    W1 = W.cpu()
    print(f'{prec}: r_pad={eng.r_pad} W-step nsplit={eng.step_w.nsplit} fuse={getattr(eng.step_w, "fuse_apply", None)} relW={float((W1 - Wr).norm() / Wr.norm()):.3e} finite={bool(torch.isfinite(W1).all())}')
    fb = eng.fw if hasattr(eng, 'fw') else None
    for name in ('fW',):
        if hasattr(eng, name): fb = getattr(eng, name)
    if fb is None:
        print('  attrs:', [a for a in dir(eng) if not a.startswith('__')])
    else:
        p1 = fb.p1_hi.view(dt).float().cpu().numpy().ravel(); p2 = fb.p2_hi.view(dt).float().cpu().numpy().ravel()
        want = W1.to(dt).float().numpy()
        b1 = b2 = 0; first = []
        for row in range(C):
            for r in range(R):
                if p1[p1_offset(row, r, eng.r_pad)] != want[row, r]: b1 += 1
                if p2[p2_offset(row, r, eng.r_pad)] != want[row, r]:
                    b2 += 1
                    if len(first) < 8: first.append((row, r, float(p2[p2_offset(row, r, eng.r_pad)]), float(want[row, r])))
        print(f'  image mismatches: P1 {b1} P2 {b2} of {C * R}; first P2: {first}')
        cs = fb.colsum.cpu()[:R] if hasattr(fb, 'colsum') else None
        if cs is not None: print('  colsum rel err', float((cs - W1.sum(0)).abs().max() / W1.sum(0).abs().max()))
    eng.h_step(); torch.cuda.synchronize()
    Hr = O.nmf_h_step(V, Wr, H0, 1, 1.0)
    H1 = H.cpu()
    bad = ~torch.isfinite(H1) | ((H1 - Hr).abs() > 1e-2 * Hr.abs().clamp_min(1e-3))
    print(f'  H-step nsplit={eng.step_h.nsplit} relH={float((H1 - Hr).norm() / Hr.norm()):.3e} bad={int(bad.sum())} rows={sorted(set(bad.nonzero()[:, 0].tolist()))[:20]} cols={sorted(set(bad.nonzero()[:, 1].tolist()))}')
