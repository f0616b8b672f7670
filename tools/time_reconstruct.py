"""Time NMF.reconstruct at BASELINE configs[1] size on the GPU box (prints ms and GB/s of the fp32 store)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-nmf_amd'))
from torchnmf_amd.nmf import NMF
dev = torch.device('cuda:0')
H, W = torch.rand(4096, 128, device=dev), torch.rand(65536, 128, device=dev)
for _ in range(3):
    y = NMF.reconstruct(H, W)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10):
    y = NMF.reconstruct(H, W)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 10
print(f'reconstruct 4096x65536 r128: {ms:.3f} ms  {y.numel() * 4 / ms / 1e6:.0f} GB/s store  {2 * y.numel() * 128 / ms / 1e9:.1f} TFLOP/s fp32')
