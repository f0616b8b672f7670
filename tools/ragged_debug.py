import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from oracle import mu_oracle as O
from torchnmf_amd.nmfd_engine import ConvMU
shape = (1, 136, 600, 2, 400)
B, Cc, L, R, T = shape
g = torch.Generator().manual_seed(sum(shape))
V = torch.rand(B, Cc, L, generator=g) + 1e-3
W0 = torch.randn(Cc, R, T, generator=g).abs()
H0 = torch.randn(B, R, L - T + 1, generator=g).abs()
dev = torch.device('cuda', 0)
rec = O.nmfd_reconstruct(H0, W0)
for rep in range(3):
    W, H = W0.clone().to(dev), H0.clone().to(dev)
    eng = ConvMU(V.to(dev), W, H, 1, 0.01, 0.02, precision='bf16x3')
    torch.cuda.synchronize(); print('ctor ok', eng.ragged, eng.fused_sums, eng.c_pad, eng.rp_pad, eng.bl_pad, flush=True)
    l0 = eng.divergence()
    torch.cuda.synchronize(); print('div ok', flush=True)
    print('loss_part', eng.loss_part.shape, eng.loss_part.data_ptr() % 256, eng._loss_main, flush=True)
    c1 = eng.loss_part.clone(); torch.cuda.synchronize(); print('clone ok', flush=True)
    lp = c1.double(); torch.cuda.synchronize(); print('double ok', flush=True)
    lp = lp.cpu()
    main = float(lp[:eng._loss_main].sum()); rag = lp[eng._loss_main:]
    print(f'rep {rep}: l0={l0:.3f} main={main:.3f} (oracle {float(O.beta_div(rec[:, :128], V[:, :128], 1)):.3f}) ragged={float(rag.sum()):.3f} (oracle {float(O.beta_div(rec[:, 128:], V[:, 128:], 1)):.3f}) nrag={rag.numel()} stale tail={float(lp[eng._loss_main + 80:].abs().sum())}')
    per = rag[:80].view(8, 10).sum(1)
    want = torch.tensor([float(O.beta_div(rec[:, 128 + i:129 + i], V[:, 128 + i:129 + i], 1)) for i in range(8)])
    print('   per channel rel err', ((per - want) / want).tolist())
