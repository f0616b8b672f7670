"""``torchnmf.metrics``-compatible divergences, evaluated on the ROCm device.

Reference: torchnmf/metrics.py:6-96.  ``input`` is the reconstruction, ``target``
the data.  Each call launches one fused elementwise-reduction HIP kernel
(``nmfmu_beta_div``) and returns a 0-dim float32 tensor on the inputs' device;
there is no CPU path and no autograd (the fit loop never differentiates them).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _capi

__all__ = ['kl_div', 'euclidean', 'is_div', 'beta_div', 'sparseness']


def beta_div(input: Tensor, target: Tensor, beta: float = 2) -> Tensor:
    """beta-divergence (metrics.py:60-96); beta = 2 / 1 / 0 are the Euclidean, KL and Itakura-Saito cases."""
    if input.device.type != 'cuda' or target.device.type != 'cuda':
        raise _capi.NmfmuError('beta_div: tensors must live on the ROCm device (no CPU fallback)')
    assert input.shape == target.shape, 'input and target must have the same shape'
    lib = _capi.load()
    x = input.detach().float().contiguous().reshape(-1)
    y = target.detach().float().contiguous().reshape(-1)
    part = torch.empty(1024, dtype=torch.float64, device=x.device)
    out = torch.zeros(1, dtype=torch.float64, device=x.device)
    _capi.check(lib.nmfmu_beta_div(x.data_ptr(), y.data_ptr(), x.numel(), float(beta), part.data_ptr(), out.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), 'nmfmu_beta_div')
    return out[0].float()


def kl_div(input: Tensor, target: Tensor) -> Tensor:
    """Generalised Kullback-Leibler divergence = beta_div(beta=1) (metrics.py:6-22)."""
    return beta_div(input, target, 1)


def euclidean(input: Tensor, target: Tensor) -> Tensor:
    """Half squared Euclidean distance = beta_div(beta=2) (metrics.py:25-39)."""
    return beta_div(input, target, 2)


def is_div(input: Tensor, target: Tensor) -> Tensor:
    """Itakura-Saito divergence = beta_div(beta=0) (metrics.py:42-57)."""
    return beta_div(input, target, 0)


def sparseness(x: Tensor) -> Tensor:
    """Hoyer's sparseness measure ``(sqrt(N) - |x|_1 / |x|_2) / (sqrt(N) - 1)`` (metrics.py:99-115)."""
    if x.device.type != 'cuda':
        raise _capi.NmfmuError('sparseness: tensors must live on the ROCm device (no CPU fallback)')
    lib = _capi.load()
    xf = x.detach().float().contiguous().reshape(-1)
    part = torch.empty(1024, dtype=torch.float64, device=x.device)
    out = torch.zeros(2, dtype=torch.float64, device=x.device)
    _capi.check(lib.nmfmu_norms(xf.data_ptr(), xf.numel(), part.data_ptr(), out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream), 'nmfmu_norms')
    n = xf.numel()
    return ((n ** 0.5 - out[0] / out[1].sqrt()) / (n ** 0.5 - 1)).float()
