"""What amdsmi reports on this box, idle and under the configs[1] MU loop (checks bench.py's SmiSampler field names)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pytorch-nmf_amd')):
    sys.path.insert(0, p)
import torch

import bench
from torchnmf_amd.engine import DenseMU

s = bench.SmiSampler(0)
print('sampler ok:', s.ok, s.err)
if s.ok:
    m = s.smi.amdsmi_get_gpu_metrics_info(s.h)
    print('idle metrics:', json.dumps({k: v for k, v in m.items() if not k.startswith('common_header')}, default=str)[:3000])
    try:
        print('power_info:', s.smi.amdsmi_get_power_info(s.h))
        print('clock_info GFX:', s.smi.amdsmi_get_clock_info(s.h, s.smi.AmdSmiClkType.GFX))
    except Exception as e:
        print('power/clock info failed:', e)
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
V = torch.rand(4096, 65536, device=dev, generator=g).bfloat16().float()
W = torch.randn(65536, 128, device=dev, generator=g).abs_()
H = torch.randn(4096, 128, device=dev, generator=g).abs_()
eng = DenseMU(V, W, H, 1.0, precision=sys.argv[1] if len(sys.argv) > 1 else 'f16')


def step():
    eng.w_step()
    eng.h_step()


for _ in range(50):
    step()
print('under load:', json.dumps(s.under_load(step, 1.5)))
if s.ok:
    m = s.smi.amdsmi_get_gpu_metrics_info(s.h)
    print('metrics right after load:', json.dumps({k: v for k, v in m.items() if 'clk' in k or 'power' in k or 'energy' in k or 'throttle' in k or 'activity' in k}, default=str)[:3000])
