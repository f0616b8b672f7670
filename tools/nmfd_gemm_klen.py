"""Where does an NMFD GEMM launch spend its time?  Launch time as a function of the contraction length (configs[3] shapes,
the engine's own descriptors with k_len shortened): slope = the k loop per k-tile, intercept = prologue + epilogue (ratio /
fp32 stores / fold) + launch ramp.  round 5, DESIGN.md section 3.4.

    python tools/nmfd_gemm_klen.py [f16|bf16|bf16x3] > gpurun_out/nmfd_klen.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch

from torchnmf_amd import _capi
from torchnmf_amd.nmfd_engine import ConvMU

dev = torch.device('cuda', 0)
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16'
Cc, L, R, T = 1025, 8192, 8, 400
g = torch.Generator(device=dev).manual_seed(1000)
V = torch.rand(1, Cc, L, device=dev, generator=g).bfloat16().float()
W = torch.randn(Cc, R, T, device=dev, generator=g).abs_()
H = torch.randn(1, R, L - T + 1, device=dev, generator=g).abs_()
eng = ConvMU(V, W, H, 1.0, precision=prec)
for _ in range(20):
    eng.w_step(); eng.h_step()


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return 1e3 * ev[0].elapsed_time(ev[1]) / reps      # us


legs = {
    'recon_w': (lambda k: eng._gemm(eng.wm, eng.hu, _capi.EPI_RATIO, x=eng.x_w, gn=eng.gn, m_rows=eng.c_main, ragged=eng.ragged_in_grid, k_len=k), 3200),
    'recon_h': (lambda k: eng._gemm(eng.hu, eng.wm, _capi.EPI_RATIO, x=eng.x_h, gn=eng.gnt, n_rows=eng.c_main, ragged=eng.ragged_in_grid, k_len=k), 3200),
    'loss': (lambda k: eng._gemm(eng.wm, eng.hu, _capi.EPI_LOSS, x=eng.x_w, out=eng.loss_part, m_valid=eng.C, n_valid=eng.B * eng.L, m_rows=eng.c_main, k_len=k), 3200),
    'num_w': (lambda k: eng._gemm(eng.gn, eng.hut, _capi.EPI_F32, out=eng.num_w, k_split=eng.w_ksplit, k_len=k), 8192),
    'num_h': (lambda k: eng._gemm(eng.wmt, eng.gnt, _capi.EPI_FOLD if eng.fold_parts else _capi.EPI_F32, out=eng.y, k_len=k), 1088),
}
out = {'precision': prec, 'shape': f'NMFD 1x{Cc}x{L} rank {R} T {T}', 'winstage': os.environ.get('TORCHNMF_AMD_NMFD_WINSTAGE', '1'), 'legs': {}}
for name, (fn, kmax) in legs.items():
    ks = sorted({max(128, (kmax * f // 5) // 128 * 128) for f in (1, 2, 3, 4)} | {kmax})
    if name == 'num_h':
        ks = [64, 320, 576, 832, 1088]
    us = [timed(lambda k=k: fn(k)) for k in ks]
    A = np.vstack([np.ones(len(ks)), np.array(ks) / 64.0]).T
    (a, b), *_ = np.linalg.lstsq(A, np.array(us), rcond=None)
    out['legs'][name] = {'k_len': ks, 'launch_us': [round(u, 2) for u in us], 'intercept_us': round(float(a), 2),
                         'us_per_ktile': round(float(b), 4), 'ktiles_full': kmax // 64,
                         'loop_share_at_full_k': round(float(b) * (kmax / 64) / us[-1], 3)}
    print(name, out['legs'][name], file=sys.stderr)
print(json.dumps(out))
