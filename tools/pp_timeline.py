"""Cycles per tile and core frequency of the ping-pong kernel's tile loop, prologue / loop / epilogue of every workgroup.
Needs a diagnostic build of the library (clock stamps compiled in):
    make -C pytorch-nmf_amd/csrc VARIANT=_dbg EXTRA=-DNMFMU_DEBUG_HOOKS
    NMFMU_LIB=$PWD/pytorch-nmf_amd/torchnmf_amd/libnmfmu_dbg.so python tools/pp_timeline.py f16
Waves 0 and 4 of workgroup 0 stamp the shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) at kernel
entry, at the start and the end of the loop and at exit; every workgroup stamps the 100 MHz clock at the same points."""
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
N, Cc, R = 4096, int(os.environ.get('PP_COLS', '65536')), 128
g = torch.Generator(device=dev).manual_seed(0)
V = torch.rand(N, Cc, device=dev, generator=g).bfloat16().float()
W = torch.randn(Cc, R, device=dev, generator=g).abs_()
H = torch.randn(N, R, device=dev, generator=g).abs_()
lib = _capi.load()
buf = torch.zeros(64 + 5 * 4096, dtype=torch.int64, device=dev)
import ctypes
tm = ctypes.c_void_p()
from torchnmf_amd.engine import KernelTimer
_capi.check(lib.nmfmu_debug_set_buffer(buf.data_ptr()), 'debug')
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
eng = DenseMU(V, W, H, 1.0, precision=prec)
for _ in range(30):
    eng.w_step(); eng.h_step()
for which in os.environ.get('PP_STEPS', 'h,w').split(','):
    res = []
    for rep in range(5):
        for _ in range(3):
            eng.w_step(); eng.h_step()
        (eng.h_step if which == 'h' else eng.w_step)()
        torch.cuda.synchronize()
        st = buf.cpu().numpy()[:32].reshape(2, 4, 4)[0]
        nt = int(st[0, 2])
        cyc = int(st[1, 0] - st[0, 0]); ref = int(st[1, 1] - st[0, 1])
        res.append((cyc / nt, cyc / max(ref, 1) * 100.0, ref / nt * 10.0, (st[0, 1] - st[2, 1]) * 0.01, (st[3, 1] - st[1, 1]) * 0.01,
                    (st[3, 1] - st[2, 1]) * 0.01))
    r = np.array(res)
    # all workgroups of the last launch: entry / loop start / loop end / exit relative to the earliest entry
    grid = (eng.step_h if which == 'h' else eng.step_w)
    nwg = (grid.owner.rows_pad // grid.block_rows) * grid.nsplit
    wg = buf.cpu().numpy()[64:64 + 5 * nwg].reshape(nwg, 5)
    tt = (wg[:, [2, 0, 1, 3]] - wg[:, 2].min()) * 0.01
    xcc = (wg[:, 4] >> 32) & 0xf
    q = lambda x: ' / '.join(f'{v:.1f}' for v in np.percentile(x, [0, 50, 90, 100]))
    print(f'   all {nwg} workgroups (us; min / median / p90 / max): entry {q(tt[:, 0])}; loop start {q(tt[:, 1])}; loop end {q(tt[:, 2])}; '
          f'exit {q(tt[:, 3])}; loop length {q(tt[:, 2] - tt[:, 1])}; epilogue {q(tt[:, 3] - tt[:, 2])}')
    print('   per XCC: median loop length ' + ', '.join(f'{int(x)}:{np.median((tt[:, 2] - tt[:, 1])[xcc == x]):.1f}' for x in np.unique(xcc)) +
          ' | max exit ' + ', '.join(f'{int(x)}:{tt[xcc == x, 3].max():.1f}' for x in np.unique(xcc)))
    print(f'{prec} lib={os.path.basename(_capi.LIB_PATH)} {which}-step cols={Cc}: {nt} tiles/WG, cycles/tile {np.median(r[:, 0]):.0f}, '
          f'core clock {np.median(r[:, 1]):.0f} MHz, {np.median(r[:, 2]):.1f} ns/tile  '
          f'=> tile loop {np.median(r[:, 2]) * nt / 1e3:.1f} us; prologue {np.median(r[:, 3]):.1f} us, epilogue {np.median(r[:, 4]):.1f} us, '
          f'workgroup 0 entry->exit {np.median(r[:, 5]):.1f} us')
_capi.check(lib.nmfmu_debug_set_buffer(None), 'debug')
