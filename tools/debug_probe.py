"""GPU debugging aid: decode which X[n][c] the fused kernel pairs with which panel row (one tile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-nmf_amd'))
import torch
from torchnmf_amd.engine import DenseMU
dev = torch.device('cuda:0')
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)

def probe(prec, half, N=128, C=64, R=32):
    X = (torch.arange(C)[None, :] + (64 * torch.arange(N)[:, None] if prec == 'bf16x3' else 0)).float()
    X = (X + 1.0).expand(N, C).contiguous()   # strictly positive
    W = torch.zeros(C, R)
    for r in range(R):
        W[r + 32 * half, r] = 1.0
    H = torch.full((N, R), 1.0 / R)
    Wd, Hd = W.to(dev).contiguous(), H.to(dev).contiguous()
    eng = DenseMU(X.to(dev), Wd, Hd, 1, precision=prec, update_W=False)
    st = eng.step_h
    eng.be.mu_partial(st)
    torch.cuda.synchronize()
    num = st.slab_num.view(st.nsplit, -1, st.r_pad).sum(0).cpu()
    dec = num[:N, :R] / 32.0 - 1.0   # = X[n][sigma(r)] - 1
    want = X[:, 32 * half: 32 * half + R] - 1.0
    bad = (dec - want).abs() > 0.5
    print(f'--- prec={prec} half={half}: mismatches {int(bad.sum())} / {bad.numel()}')
    for n in (0, 1, 5, 37, 100):
        row = dec[n] - (64 * n if prec == 'bf16x3' else 0)
        print(f'n={n:3d}:', ' '.join(f'{v:6.1f}' for v in row.tolist()))

for prec in ('bf16', 'bf16x3'):
    for half in (0, 1):
        probe(prec, half)
