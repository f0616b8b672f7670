"""Localise the fp16-mode error: numerator slabs of one W half-step against a CPU evaluation with the same roundings."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd')):
    sys.path.insert(0, p)
import torch
from torchnmf_amd import engine
from torchnmf_amd.engine import DenseMU
EPS = float(torch.finfo(torch.float32).eps)
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(448)
N, C, R = 384, 1100, 64
V = torch.rand(N, C, generator=g)
W0 = torch.randn(C, R, generator=g).abs()
H0 = torch.randn(N, R, generator=g).abs()
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engine.HipBackend.choose_nsplit = lambda self, m, k, br, d: min(ns, max(1, k // 64))
for prec, q in (('bf16', lambda x: x.bfloat16().float()), ('f16', lambda x: x.half().float())):
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = DenseMU(V.to(dev), W, H, 1.0, precision=prec)
    st = eng.step_w
    eng.be.mu_partial(st)
    torch.cuda.synchronize()
    r_pad = eng.r_pad
    num = st.slab_num.view(st.nsplit, -1, r_pad).sum(0).cpu()[:C, :R]
    S = q(W0) @ q(H0).t() + EPS                      # (C, N)
    Gn = q(q(V).t() / S)
    want = Gn @ q(H0)
    err = (num - want).abs() / want.abs().clamp_min(1e-6)
    print(f'{prec}: nsplit={st.nsplit} block_rows={st.block_rows} rel err of numerators: max {err.max():.3e} mean {err.mean():.3e}')
    if err.max() > 1e-2:
        bad = (err > 1e-2)
        print('  bad fraction per 32-row group (first 16):', [round(float(bad[i*32:(i+1)*32].float().mean()), 2) for i in range(16)])
        print('  bad fraction per 8-col rank group:', [round(float(bad[:, i*8:(i+1)*8].float().mean()), 2) for i in range(R // 8)])
        print('  ratio num/want sample:', (num / want)[:4, :8])
    l = eng.divergence()
    from oracle import mu_oracle as O
    print(f'  loss {l:.6f} oracle {float(O.beta_div(O.nmf_reconstruct(H0, W0), V, 1)):.6f}')
