"""``torchnmf.nmf``-compatible module surface over the MI355X MU engine.

Mirrors the public behaviour of the reference's ``BaseComponent`` / ``NMF`` /
``NMFD`` (torchnmf/nmf.py:173-409, 641-779): same constructor arguments,
attributes (``W``, ``H``, ``rank``, ``out_channels``, ``kernel_size``),
``forward`` / ``reconstruct`` / ``fit`` signatures, return values and error
types.  What differs is *where* the arithmetic happens: every compute entry
point runs hand-written HIP kernels on the module's ROCm device through the C
ABI of ``include/nmfmu.h``; there is no CPU path (a CPU-resident module raises
on ``forward`` / ``fit``).

``NMF2D`` / ``NMF3D`` (nmf.py:782-942) and sparse-COO targets of ``NMF.fit`` (nmf.py:351-398, 602-638) run on
the same engines.  Out of scope: ``sparse_fit`` (Hoyer-projected gradient) and autograd through
``forward``.
"""
from __future__ import annotations

import os
from collections.abc import Iterable
from typing import Optional

import torch
from torch import Tensor, nn

from . import _capi
from .constants import eps  # noqa: F401  (re-exported like the reference)

__all__ = ['BaseComponent', 'NMF', 'NMFD', 'NMF2D', 'NMF3D']      # torchnmf/nmf.py:16-18


def _sqrt2(div: float) -> float:
    """``sqrt(2 * divergence)`` of nmf.py:362 / 402.  A divergence that fp32 rounding pushed slightly below zero gives
    NaN, as ``Tensor.sqrt`` does in the reference (a Python float power would return a complex number and make the
    ``< tol`` comparison raise)."""
    d = 2.0 * div
    return d ** 0.5 if d >= 0 else float('nan')


def _new_factor(spec, trainable: bool, label: str):
    """Turn a constructor argument into (Parameter | None, inferred rank | None) -- nmf.py:213-237."""
    if isinstance(spec, Tensor):
        assert bool(torch.all(spec >= 0.)), f"Tensor {label} should be non-negative."
        p = nn.Parameter(torch.empty(*spec.size()), requires_grad=trainable)
        p.data.copy_(spec)
        return p, p.shape[1]
    if isinstance(spec, Iterable):
        shape = tuple(spec)
        # like the reference, a shape always yields a trainable factor drawn from |N(0, 1)|
        return nn.Parameter(torch.randn(*shape).abs()), shape[1]
    return None, None


def _require_device(t: Tensor, what: str) -> None:
    if t.device.type != 'cuda':
        raise _capi.NmfmuError(f'{what}: tensors live on {t.device}; torchnmf_amd computes on an MI355X only -- move '
                               f'the module and its inputs with .cuda() (there is no CPU fallback)')


class BaseComponent(nn.Module):
    """Base of the NMF modules (reference: nmf.py:173-292)."""

    def __init__(self, rank: Optional[int] = None, W=None, H=None, trainable_W: bool = True, trainable_H: bool = True):
        super().__init__()
        w_param, w_rank = _new_factor(W, trainable_W, 'W')
        h_param, h_rank = _new_factor(H, trainable_H, 'H')
        self.register_parameter('W', w_param)
        self.register_parameter('H', h_param)
        inferred = h_rank if h_rank is not None else w_rank
        if inferred is None:
            assert rank, "A rank should be given when W and H are not available!"
        else:
            if h_param is not None:
                assert h_param.shape[1] == inferred, "Latent size of H does not match with others!"
            if w_param is not None:
                assert w_param.shape[1] == inferred, "Latent size of W does not match with others!"
                self.out_channels = w_param.shape[0]
                if w_param.ndim > 2:
                    self.kernel_size = tuple(w_param.shape[2:])
            rank = inferred
        self.rank = rank

    def extra_repr(self) -> str:
        parts = [str(self.rank)]
        if self.W is not None:
            parts.append(f'out_channels={self.out_channels}')
            if hasattr(self, 'kernel_size'):
                parts.append(f'kernel_size={self.kernel_size}')
        return ', '.join(parts)

    def forward(self, H: Tensor = None, W: Tensor = None) -> Tensor:
        """Reconstruction with the module's own factors substituted for missing arguments (nmf.py:261-284)."""
        H = self.H if H is None else H
        W = self.W if W is None else W
        assert H is not None
        assert W is not None
        out = self.reconstruct(H, W)
        # provenance for trainer.BetaMu's single-layer path: which layer, and which factors, produced this tensor
        out._nmf_source = (self, H, W)
        return out

    @staticmethod
    def reconstruct(H: Tensor, W: Tensor) -> Tensor:
        raise NotImplementedError

    def _make_engine(self, V, beta, l1, l2, precision, group, allreduce=None):
        raise NotImplementedError

    def sparse_fit(self, *args, **kwargs):
        raise NotImplementedError('sparse_fit (Hoyer-projected gradient, nmf.py:411-599) is outside the MU hot path '
                                  'this engine implements')

    @torch.no_grad()
    def fit(self, V: Tensor, beta: float = 1, tol: float = 1e-4, max_iter: int = 200, verbose: bool = False,
            alpha: float = 0, l1_ratio: float = 0, *, precision: Optional[str] = None, process_group=None,
            allreduce: Optional[str] = None) -> int:
        """Minimise the beta-divergence between ``V`` and the model by multiplicative updates.

        Same contract as the reference (nmf.py:297-409): W half-step, then H
        half-step with the new W, every iteration; every 10th iteration the loss
        ``sqrt(2 * beta_div)`` is evaluated and the loop stops once
        ``(previous - loss) / loss_init < tol``.  Returns the number of iterations.

        Extra keyword-only arguments (not in the reference):
          precision      None / 'auto' (default): the fastest mode that meets the reference's 1e-4 bar.  With both
                         dimensions >= 4096, rank <= 256 and the data inside fp16's range that is a single-plane fp16
                         mode at bf16's MFMA rate: 'f16' (fp16 operands AND target) when V is exactly representable in
                         fp16, else 'f16x' (fp16 operands, V stays fp32 in HBM: nothing is rounded that the reference
                         does not round, twice the V stream).  Otherwise 'bf16x3' (split-bf16 MFMA, 3x the MFMA work,
                         matches the fp32 reference to ~1e-5; above rank 128 on the GEMM engine).  Never plain bf16.
                         Explicit: 'f16' (a V that fp16 does not hold exactly is rounded to 11 significant bits: a few
                         1e-5 per iteration, ~3e-4 after 200), 'f16x', 'bf16x3', 'bf16' (V and operands rounded to
                         bf16: objective within 1e-4, factors ~1e-3).
                         The convolutive models (NMFD / NMF2D / NMF3D): 'f16' for beta == 1 when taps and frames of
                         the last shift axis are multiples of 8, every contraction has >= 1024 terms and the data fit
                         fp16's range; otherwise 'bf16x3'.
                         The environment variable TORCHNMF_AMD_PRECISION overrides the default.  After the call
                         ``self.last_precision`` names the mode that ran.
          process_group  a torch.distributed group: V and W are then this rank's column shard
                         (V[:, Cg], W[Cg]); H is replicated.
          allreduce      sharded fits only: 'single' = ONE all-reduce of the packed [numerator | denominator] buffer per
                         iteration; 'overlap' = the H half-step in two row halves, the first half's all-reduce travelling
                         behind the second half's kernel (two collectives); 'direct' = 'single' with the whole half-step
                         (kernel, slab reduction, RCCL all-reduce, apply) enqueued by ONE C call on the compute stream
                         through the library's own communicator (nmfmu_mu_step_allreduce) instead of torch.distributed;
                         None = TORCHNMF_AMD_AR_OVERLAP / TORCHNMF_AMD_COMM; default 'single' over torch.distributed, the
                         form BASELINE's north_star names (round 5: 'overlap' was the default before, with no N > 1
                         measurement behind it).  'direct' raises when the backend has no communicator entry.
        """
        sparse = V.is_sparse
        if sparse and not isinstance(self, NMF):
            raise NotImplementedError('sparse targets are supported by NMF only (as in the reference)')
        W, H = self.W, self.H
        assert W is not None and H is not None
        _require_device(V, 'fit')
        _require_device(W, 'fit')
        _require_device(H, 'fit')
        if W.dtype != torch.float32 or H.dtype != torch.float32:
            # The reference runs in whatever dtype the module was cast to (m.double(); nmf.py:216-221).  The engine's masters
            # are fp32: fit on fp32 working copies of the factors and store the result in the module's dtype (round 6;
            # float64 / float16 / bfloat16 modules -- against the reference in float64 the factors agree like the fp32 ones do).
            if not (W.dtype.is_floating_point and H.dtype.is_floating_point):
                raise NotImplementedError(f'factors must be floating point; got W {W.dtype}, H {H.dtype}')
            keep = (W.data, H.data)
            W.data, H.data = W.data.float().contiguous(), H.data.float().contiguous()
            try:
                return self.fit(V, beta, tol, max_iter, verbose, alpha, l1_ratio, precision=precision,
                                process_group=process_group, allreduce=allreduce)
            finally:
                w32, h32 = W.data, H.data
                W.data, H.data = keep
                W.data.copy_(w32)
                H.data.copy_(h32)
        if precision is None:
            precision = os.environ.get('TORCHNMF_AMD_PRECISION', 'auto')
        beta = float(beta)
        l1 = float(alpha * l1_ratio)        # nmf.py:348-349
        l2 = float(alpha * (1 - l1_ratio))
        V = V.detach()
        if V.dtype != torch.float32:
            V = V.float()
        if sparse:
            if process_group is not None:
                raise NotImplementedError('sparse targets are not sharded')
            from .sparse_engine import SparseMU
            eng = SparseMU(V, W.data, H.data, beta, l1, l2, update_W=W.requires_grad, update_H=H.requires_grad)
        else:
            if allreduce not in (None, 'single', 'overlap', 'direct'):
                raise ValueError(f"allreduce must be None, 'single', 'overlap' or 'direct', got {allreduce!r}")
            if allreduce is not None and (process_group is None or not isinstance(self, NMF)):
                # (the choice used to be ignored silently here)
                import warnings
                warnings.warn(f"torchnmf_amd: allreduce={allreduce!r} only applies to a column-sharded NMF.fit "
                              "(process_group=...); ignored", stacklevel=2)
            eng = self._make_engine(V, beta, l1, l2, precision, process_group, allreduce)
        self.last_precision = getattr(eng, 'precision_name', None)   # what 'auto' resolved to (plain attribute, not state)

        has_bad, has_zero = eng.target_flags()   # nmf.py:329-336, computed during packing
        assert not has_bad, "Target should be non-negative."
        if has_zero and beta <= 0:
            raise ValueError("When beta <= 0 and V contains zeros, the training process may diverge. "
                             "Please add small values to V, or use a positive beta value.")

        loss_init = _sqrt2(eng.divergence())   # nmf.py:355-363
        previous = loss_init
        pbar = None
        if verbose:
            from tqdm import tqdm
            pbar = tqdm(total=max_iter)
        n_iter = -1
        warned = False
        # Loss checkpoints without a host sync (engines with AsyncLossMixin; unsharded, no progress bar): the loss of
        # checkpoint k is enqueued -- with a snapshot of W and H -- and judged at checkpoint k + 10, when its value has long
        # arrived; if the stop rule of nmf.py:405 had fired at k, the snapshot is restored and k + 1 returned: the same
        # factors and the same count as the synchronous loop, without draining the launch queue every ten iterations.
        use_async = (process_group is None and not verbose and hasattr(eng, 'checkpoint_begin')
                     and os.environ.get('TORCHNMF_AMD_ASYNC_LOSS', '1') != '0')
        pending = None           # iteration index of the checkpoint whose loss is still in flight

        def range_warning():
            nonlocal warned
            import warnings
            warned = True
            warnings.warn("torchnmf_amd: a factor grew beyond fp16's range (65504) during the fit; its fp16 "
                          "operand image is clamped from here on and the updates no longer follow the "
                          "reference.  Re-run with precision='bf16x3' (or rescale V).")

        def judge(loss, left):
            """nmf.py:402-407 for one checkpoint; True = stop there."""
            nonlocal previous
            if left and not warned:
                range_warning()
            if (previous - loss) / loss_init < tol:
                return True
            previous = loss
            return False

        def settle():
            """Judge the pending checkpoint; on stop restore its factors and return its iteration index, else None."""
            nonlocal pending
            k, pending = pending, None
            div, left = eng.checkpoint_result()
            if judge(_sqrt2(div), left):
                eng.rollback()
                return k
            return None
        try:
            for n_iter in range(max_iter):
                if W.requires_grad:
                    eng.w_step()
                if H.requires_grad:
                    eng.h_step()
                if n_iter % 10 == 9:
                    if use_async:
                        if pending is not None:
                            stopped_at = settle()
                            if stopped_at is not None:
                                n_iter = stopped_at
                                break
                        if n_iter + 1 < max_iter:     # (the last iteration's loss decides nothing: n_iter + 1 is returned either way)
                            eng.checkpoint_begin()
                            pending = n_iter
                        continue
                    loss = _sqrt2(eng.divergence())
                    left = getattr(eng, 'left_f16_range', None) is not None and not warned and eng.left_f16_range()
                    if pbar is not None:
                        pbar.set_postfix(loss=loss)
                        pbar.update(10)
                    if judge(loss, left):
                        break
            if pending is not None:                   # max_iter reached with one checkpoint still unjudged
                stopped_at = settle()
                if stopped_at is not None:
                    n_iter = stopped_at
            # fits shorter than ten iterations never reach a checkpoint: ask once more at the end
            if not warned and getattr(eng, 'left_f16_range', None) is not None and eng.left_f16_range():
                range_warning()
        finally:
            if pbar is not None:
                pbar.close()
        return n_iter + 1


class NMF(BaseComponent):
    """Non-negative matrix factorisation ``V (N, C) ~ H (N, R) @ W (C, R)^T`` (reference: nmf.py:641-697)."""

    def __init__(self, Vshape=None, rank: Optional[int] = None, **kwargs):
        if isinstance(Vshape, Iterable):
            n_rows, n_cols = Vshape
            rank = rank if rank else n_cols
            kwargs['W'] = (n_cols, rank)
            kwargs['H'] = (n_rows, rank)
        super().__init__(rank, **kwargs)

    @staticmethod
    def reconstruct(H: Tensor, W: Tensor) -> Tensor:
        """``H @ W.T`` (nmf.py:691-693) by an exact-fp32 MFMA kernel on the device."""
        _require_device(H, 'reconstruct')
        _require_device(W, 'reconstruct')
        assert H.dim() >= 2 and W.dim() == 2 and H.shape[-1] == W.shape[1]
        lib = _capi.load()
        lead = tuple(H.shape[:-1])                       # F.linear accepts leading batch dimensions (nmf.py:693)
        Hc = H.detach().float().reshape(-1, H.shape[-1]).contiguous()
        Wc = W.detach().float().contiguous()
        out = torch.empty(Hc.shape[0], Wc.shape[0], dtype=torch.float32, device=H.device)
        _capi.check(lib.nmfmu_reconstruct(Hc.data_ptr(), Hc.shape[0], Wc.data_ptr(), Wc.shape[0], Hc.shape[1],
                                          out.data_ptr(), out.stride(0), torch.cuda.current_stream().cuda_stream),
                    'nmfmu_reconstruct')
        out = out.reshape(lead + (Wc.shape[0],))
        return out if H.dtype == torch.float32 else out.to(H.dtype)      # (a module cast to another dtype answers in it)

    def _make_engine(self, V, beta, l1, l2, precision, group, allreduce=None):
        from .engine import DenseMU
        assert V.dim() == 2 and V.shape == (self.H.shape[0], self.W.shape[0]), \
            f'V must be {(self.H.shape[0], self.W.shape[0])}, got {tuple(V.shape)}'
        for p in (self.W, self.H):
            if not p.data.is_contiguous():
                p.data = p.data.contiguous()
        R = self.W.shape[1]
        # The fused kernels keep rank-wide accumulators in registers: single-plane operands (fp16 / bf16) up to rank 256,
        # the fp32-grade split-bf16 mode up to rank 128.  Everything else runs on the GEMM engine (NMF = the T = 1 member
        # of the NMFD family), which has no rank limit but is not sharded.  'auto' means "meets the 1e-4 parity bar":
        # at rank 129..256 that is the fp16 mode of the fused kernel when DenseMU's size / range test admits it, else
        # (unsharded) the GEMM engine, else (sharded) an error -- never silently the plain bf16 mode.
        auto = precision in (None, 'auto')
        wide = R > 256 or (R > 128 and precision == 'bf16x3')
        if R > 128 and R <= 256 and auto:
            # one admission test for both engines (DenseMU.auto_mode; sharded: its answer is all-reduced, so every
            # rank takes the same branch -- including the raise below)
            from .engine import DenseMU as _D, DEFAULT_BACKEND_FACTORY
            be = DEFAULT_BACKEND_FACTORY()
            single = _D.auto_mode(V, self.W.data, self.H.data, be.pad_rank(R), be, beta, group)
            if single is not None:
                precision = single
            elif group is None:
                wide = True
            else:
                raise NotImplementedError(
                    "precision='auto' on a column-sharded fit at rank 129..256 needs a single-plane fp16 mode (both "
                    f"dimensions >= {_D.F16_MIN_DIM} on every rank, data within fp16's range); pass precision='bf16' "
                    f"(factors ~1e-3), 'f16' or 'f16x' explicitly")
        if wide and group is not None:
            raise NotImplementedError('column sharding is implemented for the fused kernels (rank <= 256; bf16x3 up to '
                                      'rank 128)')
        if wide:
            from .nmfd_engine import WideRankMU
            return WideRankMU(V, self.W.data, self.H.data, beta, l1, l2, precision=precision,
                              update_W=self.W.requires_grad, update_H=self.H.requires_grad)
        return DenseMU(V, self.W.data, self.H.data, beta, l1, l2, precision=precision, group=group,
                       update_W=self.W.requires_grad, update_H=self.H.requires_grad, allow_f16=True,
                       ar_overlap=None if allreduce is None else allreduce == 'overlap', allow_gram=True,
                       ar_direct=None if allreduce is None else allreduce == 'direct')


class NMFD(BaseComponent):
    """1-D convolutive NMF ``V[b,c,l] ~ sum_t sum_r W[c,r,t] H[b,r,l-t]`` (reference: nmf.py:700-779)."""

    def __init__(self, Vshape=None, rank: Optional[int] = None, T=1, **kwargs):
        if isinstance(Vshape, Iterable):
            if isinstance(T, Iterable):
                T, = tuple(T)
            batch, n_chan, length = Vshape
            rank = rank if rank else n_chan
            kwargs['W'] = (n_chan, rank, T)
            kwargs['H'] = (batch, rank, length - T + 1)
        super().__init__(rank, **kwargs)

    @staticmethod
    def reconstruct(H: Tensor, W: Tensor) -> Tensor:
        from .nmfd_engine import reconstruct as _recon
        return _recon(H, W)

    def _make_engine(self, V, beta, l1, l2, precision, group, allreduce=None):
        from .nmfd_engine import ConvMU
        if group is not None:
            raise NotImplementedError('NMFD is not sharded (replicas only): sharding L needs a (T-1)-column halo')
        for p in (self.W, self.H):                      # the engine works on the parameters' storage in place
            if not p.data.is_contiguous():
                p.data = p.data.contiguous()
        return ConvMU(V, self.W.data, self.H.data, beta, l1, l2, precision=precision,
                      update_W=self.W.requires_grad, update_H=self.H.requires_grad)


def _ntuple(x, n):
    return tuple(x) if isinstance(x, Iterable) else (x,) * n


class NMF2D(NMFD):
    """2-D convolutive NMF ``V[b,c,l,m] ~ sum W[c,r,i,j] H[b,r,l-i,m-j]`` (reference: nmf.py:782-865).  Same engine
    as NMFD with two shift axes (explicit unfold / fold kernels, nmfmu_convnd_*)."""

    def __init__(self, Vshape=None, rank: Optional[int] = None, kernel_size=1, **kwargs):
        if isinstance(Vshape, Iterable):
            kernel_size = _ntuple(kernel_size, 2)
            batch, channel, K, M = Vshape
            rank = rank if rank else K
            kwargs['W'] = (channel, rank) + kernel_size
            kwargs['H'] = (batch, rank, K - kernel_size[0] + 1, M - kernel_size[1] + 1)
        BaseComponent.__init__(self, rank, **kwargs)


class NMF3D(NMFD):
    """3-D convolutive NMF (reference: nmf.py:868-942); three shift axes."""

    def __init__(self, Vshape=None, rank: Optional[int] = None, kernel_size=1, **kwargs):
        if isinstance(Vshape, Iterable):
            kernel_size = _ntuple(kernel_size, 3)
            batch, channel, N, K, M = Vshape
            rank = rank if rank else K
            kwargs['W'] = (channel, rank) + kernel_size
            kwargs['H'] = (batch, rank, N - kernel_size[0] + 1, K - kernel_size[1] + 1, M - kernel_size[2] + 1)
        BaseComponent.__init__(self, rank, **kwargs)
