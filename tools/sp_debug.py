"""Localise an error of the software-pipelined rank-256 kernel: the H half-step's partial sums (nmfmu_mu_partial) of a small
problem against a host reference with the kernel's own rounding points, by contraction column, by rank tile, by owner row."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd')):
    sys.path.insert(0, p)
import torch

from torchnmf_amd.engine import DenseMU

dev = torch.device('cuda', 0)
N, C, R = 128, int(os.environ.get('SP_COLS', '512')), 256
g = torch.Generator().manual_seed(1)
V = torch.rand(N, C, generator=g).half().float()
W = (torch.rand(C, R, generator=g) + 0.5).half().float()
H = (torch.rand(N, R, generator=g) + 0.5).half().float()
eps = 2.0 ** -23


def run(Vx):
    eng = DenseMU(Vx.to(dev), W.clone().to(dev), H.clone().to(dev), 1.0, precision='f16')
    st = eng.step_h
    eng.be.mu_partial(st)
    torch.cuda.synchronize()
    num = st.slab_num.view(st.nsplit, st.owner.rows_pad, st.r_pad).sum(0)[:N, :R].cpu()
    return num, st.nsplit


def ref(Vx):
    S = H.double() @ W.double().t() + eps
    Gn = (Vx.double() / S).half().double()
    return (Gn @ W.double()).float()


num, ns = run(V)
want = ref(V)
err = (num - want).abs() / want.abs().clamp_min(1e-6)
print('nsplit', ns, 'C', C, 'tiles', C // 64, 'max rel err', float(err.max()), 'mean', float(err.mean()))
print('rel err by rank tile   :', [f'{float(err[:, 32 * rt:32 * rt + 32].mean()):.2e}' for rt in range(8)])
print('rel err by row group/32:', [f'{float(err[32 * w:32 * w + 32].mean()):.2e}' for w in range(4)])
# by contraction column group of 8 (one MFMA k-slice), via targets that are nonzero in one group only
bad = []
for k0 in range(0, min(C, 256), 8):
    Vx = torch.zeros_like(V)
    Vx[:, k0:k0 + 8] = V[:, k0:k0 + 8]
    n1, _ = run(Vx)
    w1 = ref(Vx)
    e1 = float(((n1 - w1).abs().max()) / w1.abs().max())
    if e1 > 2e-3:
        bad.append((k0, f'{e1:.2e}'))
print('column groups (of 8) with wrong contributions:', bad if bad else 'none')
