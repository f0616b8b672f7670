"""``BetaMu`` -- the optimizer-style entry to the same multiplicative update (reference: trainer.py:8-121).

The reference's ``BetaMu.step(closure)`` back-propagates ``output_neg`` / ``output_pos`` through whatever graph the
closure built.  When the closure's prediction is the direct output of ONE ``NMF`` layer those two backward passes
are exactly the numerator / denominator contractions of the fused HIP kernel, so this
class runs them there and applies trainer.py:93-112 in ``nmfmu_trainer_apply``.  Anything else (stacked layers,
arithmetic on the prediction, NMFD) is outside this engine and raises ``NotImplementedError`` -- there is no autograd
fallback.

    trainer = BetaMu(m.parameters(), beta=1)
    def closure():
        trainer.zero_grad()
        return V, m()          # or ``return V, m``: hands over the layer and skips materialising H @ W^T
    trainer.step(closure)
"""
from typing import Dict, Tuple

import torch
from torch.optim.optimizer import Optimizer

from .engine import DenseMU
from . import nmf as _nmf
from .nmf import NMF, BaseComponent

__all__ = ['BetaMu']


class BetaMu(Optimizer):
    """Multiplicative updater minimising the beta-divergence (same arguments and checks as trainer.py:24-33).

    ``precision`` (keyword-only, not in the reference) selects the MFMA operand format as in ``NMF.fit``.
    """

    def __init__(self, params, beta=1, l1_reg=0, l2_reg=0, orthogonal=0, *, precision='auto'):
        if not 0.0 <= l1_reg:
            raise ValueError("Invalid l1_reg value: {}".format(l1_reg))
        if not 0.0 <= l2_reg:
            raise ValueError("Invalid l2_reg value: {}".format(l2_reg))
        if not 0.0 <= orthogonal:
            raise ValueError("Invalid orthogonal value: {}".format(orthogonal))
        defaults = dict(beta=beta, l1_reg=l1_reg, l2_reg=l2_reg, orthogonal=orthogonal)
        super().__init__(params, defaults)
        self._precision = precision
        self._engines: Dict[Tuple, Tuple[DenseMU, list]] = {}

    # ------------------------------------------------------------------
    @staticmethod
    def _source(pred):
        """(layer, H, W) behind the closure's prediction, or raise."""
        if isinstance(pred, BaseComponent):            # deferred form: the layer itself
            layer, H, W = pred, pred.H, pred.W
        else:
            src = getattr(pred, '_nmf_source', None)
            if src is None:
                raise NotImplementedError(
                    'BetaMu: the closure must return (target, prediction) where prediction is the direct output of '
                    'one torchnmf_amd NMF layer (m() or m itself); general autograd graphs are outside this engine')
            layer, H, W = src
        if not isinstance(layer, NMF):
            raise NotImplementedError(f'BetaMu: only NMF layers are supported, got {type(layer).__name__}')
        assert H is not None and W is not None
        return layer, H, W

    def _engine(self, V, H, W, beta, l1, l2) -> DenseMU:
        """DenseMU bound to (V, W, H), rebuilt when any of them is replaced and refreshed when edited in place."""
        key = (V.data_ptr(), tuple(V.shape), W.data_ptr(), H.data_ptr(), float(beta), float(l1), float(l2))
        hit = self._engines.get(key)
        versions = [V._version, W._version, H._version]
        if hit is not None and hit[1][0] == versions[0]:
            eng, seen = hit
            if seen[1:] != versions[1:]:               # someone else edited W / H since our last update
                eng.refresh_images()
                seen[1:] = versions[1:]
            return eng
        self._engines.clear()                          # one live binding: packed V is as large as V
        eng = DenseMU(V, W.data, H.data, beta, l1, l2, precision=self._precision)
        bad, _ = eng.target_flags()
        assert not bad, "Target should be non-negative."
        self._engines[key] = (eng, versions)
        return eng

    @torch.no_grad()
    def step(self, closure):
        """One multiplicative update of every parameter (trainer.py:36-121).

        ``closure() -> (target, prediction)`` is re-evaluated before each parameter's update, like the reference
        (trainer.py:72), so later parameters see the earlier ones already updated.
        """
        status_cache = {}
        for group in self.param_groups:
            for p in group['params']:
                status_cache[id(p)] = p.requires_grad
                p.requires_grad = False
        try:
            for group in self.param_groups:
                beta, l1, l2, ortho = group['beta'], group['l1_reg'], group['l2_reg'], group['orthogonal']
                for p in group['params']:
                    if not status_cache[id(p)]:
                        continue
                    p.requires_grad = True
                    V, pred = closure()
                    layer, H, W = self._source(pred)
                    which = 'W' if p is W else ('H' if p is H else None)
                    if which is None:                  # p does not feed this prediction (trainer.py:73-75)
                        p.requires_grad = False
                        continue
                    for t, what in ((V, 'BetaMu target'), (W, 'BetaMu W'), (H, 'BetaMu H')):
                        _nmf._require_device(t, what)
                    V = V.detach()
                    if V.dtype != torch.float32 or not V.is_contiguous():
                        V = V.float().contiguous()
                    for q in (W, H):
                        if not q.data.is_contiguous():
                            q.data = q.data.contiguous()
                    assert V.dim() == 2 and V.shape == (H.shape[0], W.shape[0]), \
                        f'target must be {(H.shape[0], W.shape[0])}, got {tuple(V.shape)}'
                    eng = self._engine(V, H, W, beta, l1, l2)
                    if p.grad is None or p.grad.shape != p.shape or not p.grad.is_contiguous():
                        p.grad = torch.empty_like(p.data)
                    eng.trainer_step(which, ortho, p.grad)
                    p.requires_grad = False
        finally:
            for group in self.param_groups:
                for p in group['params']:
                    p.requires_grad = status_cache[id(p)]
        return None
