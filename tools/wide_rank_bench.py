"""NMF with rank > 256 (GEMM engine, nmfd_engine.WideRankMU): iterations/s with 128 x 128 vs 256 x 256 GEMM tiles."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-nmf_amd'))
import torch
N, C, R = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 32768, 512)))
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(0)
V = torch.rand(N, C, device=dev, generator=g)
for mode in ('128', '256', '128', '256'):
    os.environ['TORCHNMF_AMD_NMFD_TILE'] = mode
    from torchnmf_amd.nmfd_engine import WideRankMU
    W = torch.rand(C, R, device=dev, generator=g); H = torch.rand(N, R, device=dev, generator=g)
    eng = WideRankMU(V, W, H, 1.0, precision='bf16')
    for _ in range(3):
        eng.w_step(); eng.h_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        eng.w_step(); eng.h_step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f'rank {R} {N}x{C} tile={mode} (engine tile {eng.eng.tile}): {dt * 1e3:.3f} ms/iteration = {8.0 * N * C * R / dt / 1e12:.0f} TFLOP/s, loss {eng.divergence():.6g}')
