"""Where a launch of the beta = 2 stream kernel (kModeXB, nmfmu_fused.h) spends its time: entry / loop start / loop end /
exit of every workgroup on the constant 100 MHz clock, and which CU it ran on.  Needs a diagnostic build:
    make -C pytorch-nmf_amd/csrc VARIANT=_dbg EXTRA=-DNMFMU_DEBUG_HOOKS
    NMFMU_LIB=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu_dbg.so python tools/xb_timeline.py f16 [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch

from torchnmf_amd import _capi
from torchnmf_amd.engine import DenseMU

dev = torch.device('cuda', 0)
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
Cc, R = 65536, 128
g = torch.Generator(device=dev).manual_seed(0)
V = torch.rand(N, Cc, device=dev, generator=g).bfloat16().float()
W = torch.randn(Cc, R, device=dev, generator=g).abs_()
H = torch.randn(N, R, device=dev, generator=g).abs_()
lib = _capi.load()
buf = torch.zeros(64 + 5 * 4096, dtype=torch.int64, device=dev)
_capi.check(lib.nmfmu_debug_set_buffer(buf.data_ptr()), 'debug (diagnostic build needed)')
eng = DenseMU(V, W, H, 2.0, precision=prec, allow_gram=True)
assert eng.gram_path
for _ in range(30):
    eng.w_step(); eng.h_step()
q = lambda x: ' / '.join(f'{v:.1f}' for v in np.percentile(x, [0, 10, 50, 90, 100]))
for which in ('w', 'h'):
    st = eng.step_w if which == 'w' else eng.step_h
    nwg = (st.owner.rows_pad // st.block_rows) * st.nsplit
    for _ in range(3):
        eng.w_step(); eng.h_step()
    if which == 'w':
        eng.w_step()
    else:
        eng.w_step(); eng.h_step()
    torch.cuda.synchronize()
    wg = buf.cpu().numpy()[64:64 + 5 * nwg].reshape(nwg, 5)
    tt = (wg[:, [2, 0, 1, 3]] - wg[:, 2].min()) * 0.01
    hw = wg[:, 4] & 0xffffffff
    xcc = (wg[:, 4] >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)       # HW_ID: CU_ID [11:8], SE_ID [15:13]
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    print(f'{prec} {which}-step rows={N}: {nwg} workgroups x {st.panel.rows_pad // 64 // st.nsplit} tiles, nsplit {st.nsplit}; '
          f'distinct CUs {len(per_cu)}, workgroups per CU min/max {per_cu.min()}/{per_cu.max()}')
    print(f'   us (min / p10 / median / p90 / max): entry {q(tt[:, 0])}; loop start {q(tt[:, 1])}; loop end {q(tt[:, 2])}; exit {q(tt[:, 3])}')
    print(f'   prologue {q(tt[:, 1] - tt[:, 0])}; loop {q(tt[:, 2] - tt[:, 1])}; epilogue {q(tt[:, 3] - tt[:, 2])}; '
          f'per tile in the loop (median) {np.median(tt[:, 2] - tt[:, 1]) / (st.panel.rows_pad // 64 // st.nsplit) * 1e3:.0f} ns')
    loop = tt[:, 2] - tt[:, 1]
    print('   loop length, median per XCC: ' + ', '.join(f'{int(x)}:{np.median(loop[xcc == x]):.1f}' for x in np.unique(xcc)) +
          ' | max exit per XCC: ' + ', '.join(f'{int(x)}:{tt[xcc == x, 3].max():.1f}' for x in np.unique(xcc)))
    se = (hw >> 13) & 0x7
    print('   loop length, median per SE (all XCCs): ' + ', '.join(f'{int(x)}:{np.median(loop[se == x]):.1f}' for x in np.unique(se)))
    bid = np.arange(nwg)
    oct_ = bid * 8 // nwg
    print('   loop length, median per eighth of the grid (blockIdx order): ' + ', '.join(f'{np.median(loop[oct_ == x]):.1f}' for x in range(8)))
    order = np.argsort(loop)
    print('   slowest 8 workgroups: ' + ', '.join(f'b{int(b)}(xcc{int(xcc[b])},cu{int(cu[b]) & 0xff:02x}):{loop[b]:.0f}' for b in order[-8:]) +
          ' | fastest 8: ' + ', '.join(f'b{int(b)}(xcc{int(xcc[b])},cu{int(cu[b]) & 0xff:02x}):{loop[b]:.0f}' for b in order[:8]))
    # the two workgroups of one CU: do they finish together?
    pair = {}
    for b in range(nwg):
        pair.setdefault(int(cu[b]), []).append(loop[b])
    d = np.array([abs(v[0] - v[1]) for v in pair.values() if len(v) == 2])
    m = np.array([np.mean(v) for v in pair.values() if len(v) == 2])
    print(f'   same-CU pairs: |difference| median {np.median(d):.1f} us; CU means min / median / max {m.min():.1f} / {np.median(m):.1f} / {m.max():.1f}')
    late = tt[:, 0] > 5.0
    print(f'   workgroups entering later than 5 us after the first: {int(late.sum())}' + (f' (median entry {np.median(tt[late, 0]):.1f} us)' if late.any() else ''))
