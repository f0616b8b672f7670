"""Where does the reduced-precision error of the fused MU step enter?  (CPU emulation, build container only.)

Emulates the fused kernel's arithmetic for beta = 1 with a chosen storage format at each of its four rounding
points and reports the factor error against the fp32 iteration after `iters` iterations:

    S   = own_q(A) . pan1_q(B)^T + eps      GEMM1  (fp32 accumulate)
    Gn  = gn_q(X / S)                        elementwise
    num = Gn . pan2_q(B)                     GEMM2  (fp32 accumulate)

Formats: 'f32' (exact), 'bf16', 'f16', 'bf16x2' (hi + lo bf16 planes = 16 significant bits), 'f16x2'.
Usage: python tools/precision_probe.py            (prints a table; results quoted in DESIGN.md section 4)
"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
EPS = float(torch.finfo(torch.float32).eps)


def q(x, fmt):
    if fmt == 'f32':
        return x
    if fmt == 'bf16':
        return x.bfloat16().float()
    if fmt == 'f16':
        return x.half().float()
    if fmt == 'bf16x2':
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    if fmt == 'f16x2':
        hi = x.half().float()
        return hi + (x - hi).half().float()
    raise ValueError(fmt)


def half_step(X, A, B, fm):
    """owner A (M,R), panel B (K,R), X (M,K)."""
    own, pan1, gnf, pan2 = fm
    S = q(A, own) @ q(B, pan1).t() + EPS
    Gn = q(X / S, gnf)
    num = Gn @ q(B, pan2)
    neg = num.relu() + EPS
    pos = B.sum(0, keepdim=True)
    return A * (neg / pos)


def run(V, W0, H0, fm, iters):
    W, H = W0.clone(), H0.clone()
    Vt = V.t().contiguous()
    for _ in range(iters):
        W = half_step(Vt, W, H, fm)
        H = half_step(V, H, W, fm)
    return W, H


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    torch.set_num_threads(8)
    cases = []
    g = np.load('tests/golden/g2_cfg1.npz')
    V = torch.from_numpy(g['V_bf16_bits'].astype(np.int16)).view(torch.bfloat16).float()
    cases.append(('cfg1 256x512 r16 (golden init)', V, torch.from_numpy(g['W0']), torch.from_numpy(g['H0']), 50))
    gen = torch.Generator().manual_seed(0)
    for (n, c, r, it) in ((1024, 2048, 64, 50), (512, 4096, 128, 50)):
        V = torch.rand(n, c, generator=gen).bfloat16().float()
        cases.append((f'{n}x{c} r{r}', V, torch.randn(c, r, generator=gen).abs(), torch.randn(n, r, generator=gen).abs(), it))
    modes = [
        ('bf16 everywhere (shipped bf16)', ('bf16', 'bf16', 'bf16', 'bf16')),
        ('only owner bf16', ('bf16', 'f32', 'f32', 'f32')),
        ('only GEMM1 panel bf16', ('f32', 'bf16', 'f32', 'f32')),
        ('only Gn bf16', ('f32', 'f32', 'bf16', 'f32')),
        ('only GEMM2 panel bf16', ('f32', 'f32', 'f32', 'bf16')),
        ('owner bf16x2, rest bf16', ('bf16x2', 'bf16', 'bf16', 'bf16')),
        ('owner+panel1 bf16x2, rest bf16', ('bf16x2', 'bf16x2', 'bf16', 'bf16')),
        ('GEMM1 bf16x2, Gn bf16, panel2 bf16x2', ('bf16x2', 'bf16x2', 'bf16', 'bf16x2')),
        ('all bf16x2 (shipped bf16x3)', ('bf16x2', 'bf16x2', 'bf16x2', 'bf16x2')),
        ('f16 everywhere', ('f16', 'f16', 'f16', 'f16')),
        ('f16 operands, Gn bf16', ('f16', 'f16', 'bf16', 'f16')),
        ('owner f16x2, rest f16', ('f16x2', 'f16', 'f16', 'f16')),
        ('owner f16x2, panels f16, Gn bf16', ('f16x2', 'f16', 'bf16', 'f16')),
    ]
    for name, V, W0, H0, it in cases:
        Wr, Hr = run(V, W0, H0, ('f32',) * 4, it)
        print(f'== {name}, {it} iterations')
        for mname, fm in modes:
            W, H = run(V, W0, H0, fm, it)
            print(f'  {mname:42s} relW={rel(W, Wr):.2e} relH={rel(H, Hr):.2e}')


if __name__ == '__main__':
    main()
