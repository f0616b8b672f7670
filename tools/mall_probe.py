"""Go / no-go for "read V from HBM once" (VERDICT r4 item 1a): does the ping-pong kernel's tile loop run faster when its
X stream comes out of the 256 MiB Infinity Cache instead of HBM?

The SHIPPED H half-step kernel (nmfmu_mu_partial on step_h: pp_kernel, 256 workgroups = full chip) is launched back to
back on problems whose packed X is 64 / 128 / 256 / 512 MiB (C = 8192 ... 65536 at N = 4096, rank 128).  Nothing else
runs in between -- no W half-step, the factors never change -- so a packed X of <= 128 MiB stays resident in the MALL from
one launch to the next, while 512 MiB (configs[1]) streams from HBM every time.  Reported per size: launch time
(hipEvents), ns per tile and core clock from the kernel's own stamps (diagnostic build, see tools/pp_timeline.py),
socket power and clock from amdsmi (bench.SmiSampler).  The tile loop is isolated from prologue / epilogue by the stamps
and, without them, by the slope of launch time over tiles per workgroup.

    make -C pytorch-nmf_amd/csrc VARIANT=_dbg EXTRA=-DNMFMU_DEBUG_HOOKS
    make -C pytorch-nmf_amd/csrc VARIANT=_dbgx EXTRA="-DNMFMU_DEBUG_HOOKS -DNMFMU_PP_X_DEFAULT_POLICY"   # X loads without `nt`
    NMFMU_LIB=.../libnmfmu_dbg.so python tools/mall_probe.py f16 > gpurun_out/mall_nt.json
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
from torchnmf_amd.engine import DenseMU
import bench

dev = torch.device('cuda', 0)
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16'
lib = _capi.load()
buf = torch.zeros(64 + 5 * 4096, dtype=torch.int64, device=dev)
has_dbg = lib.nmfmu_debug_set_buffer(buf.data_ptr()) == 0      # product builds answer NMFMU_ERR_UNSUPPORTED: no stamps
lib.nmfmu_debug_set_buffer(None)
N, R = 4096, 128
rows = []
for Cc in (8192, 16384, 32768, 65536):
    g = torch.Generator(device=dev).manual_seed(0)
    V = torch.rand(N, Cc, device=dev, generator=g).bfloat16().float()
    W = torch.randn(Cc, R, device=dev, generator=g).abs_()
    H = torch.randn(N, R, device=dev, generator=g).abs_()
    eng = DenseMU(V, W, H, 1.0, precision=prec)
    st = eng.step_h
    assert st.block_rows == 256, 'ping-pong kernel expected'
    nwg = (st.owner.rows_pad // st.block_rows) * st.nsplit
    tiles = st.panel.rows_pad // 64 // st.nsplit
    x_mib = st.owner.rows_pad * st.panel.rows_pad * 2 / 2 ** 20

    def launch():
        eng.be.mu_partial(st)
    for _ in range(200):          # warm: clocks, MALL contents
        launch()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 400
    ev[0].record()
    for _ in range(reps):
        launch()
    ev[1].record()
    torch.cuda.synchronize()
    us = 1e3 * ev[0].elapsed_time(ev[1]) / reps
    rec = {'cols': Cc, 'x_mib': x_mib, 'workgroups': nwg, 'tiles_per_wg': tiles, 'launch_us': round(us, 2),
           'us_per_tile_launch_over_tiles': round(us / tiles, 4)}
    if has_dbg:
        _capi.check(lib.nmfmu_debug_set_buffer(buf.data_ptr()), 'debug')
        res = []
        for _ in range(7):
            for _ in range(5):
                launch()
            torch.cuda.synchronize()
            s = buf.cpu().numpy()[:32].reshape(2, 4, 4)[0]
            nt = int(s[0, 2])
            cyc, ref = int(s[1, 0] - s[0, 0]), int(s[1, 1] - s[0, 1])
            res.append((cyc / nt, cyc / max(ref, 1) * 100.0, ref / nt * 10.0))
        wg = buf.cpu().numpy()[64:64 + 5 * nwg].reshape(nwg, 5)
        loop_us = (wg[:, 1] - wg[:, 0]) * 0.01
        r = np.median(np.array(res), axis=0)
        rec.update(stamp_tiles=nt, cycles_per_tile=round(float(r[0]), 1), core_clock_mhz_in_loop=round(float(r[1]), 1),
                   ns_per_tile=round(float(r[2]), 2), loop_us_median_over_wgs=round(float(np.median(loop_us)), 2),
                   loop_us_max_over_wgs=round(float(loop_us.max()), 2))
        _capi.check(lib.nmfmu_debug_set_buffer(None), 'debug')
    tel = bench.SmiSampler(0).under_load(launch, 0.8)
    rec.update(smi_clock_mhz=tel.get('clock_mhz'), smi_power_w=tel.get('power_w'), hbm_activity_pct=tel.get('hbm_activity_pct'))
    rows.append(rec)
    print(json.dumps(rec), file=sys.stderr)
    del eng, V, W, H
# slope of the launch time over tiles per workgroup between the two ends = loop time per tile without any stamps
big, small = rows[-1], rows[1]
out = {'lib': os.path.basename(_capi.LIB_PATH), 'precision': prec, 'kernel': 'pp_kernel (H half-step, slab stores)', 'rows': rows,
       'note': 'X <= 128 MiB stays in the 256 MiB Infinity Cache between back-to-back launches; 512 MiB = configs[1] streams from HBM'}
if has_dbg:
    out['ns_per_tile_mall_resident_128MiB'] = small['ns_per_tile']
    out['ns_per_tile_hbm_512MiB'] = big['ns_per_tile']
    out['mall_speedup_per_tile'] = round(big['ns_per_tile'] / small['ns_per_tile'], 4)
print(json.dumps(out))
