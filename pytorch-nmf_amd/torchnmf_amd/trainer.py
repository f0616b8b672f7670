"""``BetaMu`` -- the optimizer-style entry to the same multiplicative update (reference: trainer.py:8-121).

The reference's ``BetaMu.step(closure)`` back-propagates ``output_neg`` / ``output_pos`` through whatever graph the
closure built.  Two graph shapes are recognised through the provenance tag ``forward()`` leaves on its output:

* ONE ``NMF`` layer: the two backward passes are exactly the numerator / denominator contractions of the fused HIP
  kernel; they run there and trainer.py:93-112 is applied by ``nmfmu_trainer_apply``.
* a CHAIN of ``NMF`` layers (``nn.Sequential``, each layer's output feeding the next one's ``H`` -- the reference's own
  trainer test, tests/test_trainer.py:10-32): prediction = X0 W1^T W2^T ...; the forward products, the back-propagated
  seeds ``G_{k-1} = G_k W_k`` and the parameter gradients ``G_k^T X_{k-1}`` are exact-fp32 MFMA products
  (``nmfmu_reconstruct``), the seeds come from ``nmfmu_mu_terms`` and the update from ``nmfmu_trainer_update``.

* ONE convolutive layer (``NMFD`` / ``NMF2D`` / ``NMF3D``, round 6): the backward passes through the convolution are the
  numerator / denominator GEMMs of the convolutive engine (``nmfd_engine.ConvMU``: reconstruction + ratio planes, then
  ratio x Toeplitz operand for W and ratio x W for H); their results, in the parameter's
  own layout, go through ``nmfmu_trainer_update`` like a chain's.

Anything else (arithmetic on the prediction, chains that contain a convolutive layer) raises ``NotImplementedError`` --
there is no autograd fallback.

    trainer = BetaMu(m.parameters(), beta=1)
    def closure():
        trainer.zero_grad()
        return V, m()          # or ``return V, m``: hands over the layer and skips materialising H @ W^T
    trainer.step(closure)
"""
from typing import Dict, List, Tuple

import torch
from torch.optim.optimizer import Optimizer

from .engine import DenseMU
from . import nmf as _nmf
from .nmf import NMF, BaseComponent

__all__ = ['BetaMu', 'SparsityProj']


class SparsityProj(Optimizer):
    """Placeholder for the reference's Hoyer-projected gradient optimizer (torchnmf/trainer.py:124-226).  It is a
    different algorithm (serial projection loop) outside the MU hot path this engine implements;
    importing the name works, constructing it says so."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('trainer.SparsityProj (Hoyer-projected gradient, trainer.py:124-226) is outside the MU '
                                  'hot path torchnmf_amd implements; use the reference package for sparse_fit')


class _ConvBinding:
    """One convolutive layer bound to a target for ``BetaMu``: the convolutive engine drives the GEMMs (no fused update:
    ``own_loop=False``), this class turns their outputs into (neg, pos) of trainer.py:93-97 in the parameter's layout."""

    def __init__(self, V, W, H, beta, precision):
        from .nmfd_engine import ConvMU
        from . import _capi
        self._capi = _capi
        # 'auto' = the fp32-grade split mode (the engine's own fp16 admission belongs to its fused fit loop)
        self.eng = e = ConvMU(V, W, H, beta, precision=('bf16x3' if precision in (None, 'auto') else precision), own_loop=False)
        self.lib, self.beta = e.lib, float(beta)
        dev = V.device
        self.numh = torch.empty(e.B * e.R * e.Lh, dtype=torch.float32, device=dev)
        self.denh = None if e.kl else torch.empty_like(self.numh)

    @property
    def precision_name(self):
        return self.eng.precision_name

    def _s(self):
        return torch.cuda.current_stream().cuda_stream

    def _w_plane(self, buf):
        """[c][(r, t)] GEMM output (row pitch rp_pad, split-K slabs summed) -> tensor of W's shape."""
        e = self.eng
        if e.w_ksplit > 1:
            self._capi.check(self.lib.nmfmu_slab_sum(buf.data_ptr(), (e.c_rows or e.c_pad) * e.rp_pad, e.w_ksplit, self._s()),
                             'nmfmu_slab_sum')
        rows = e.c_rows or e.c_pad
        return buf[:rows * e.rp_pad].view(rows, e.rp_pad)[:e.C, :e.R * e.T].reshape(e.W.shape).contiguous()

    def grads_w(self):
        """(neg, pos) for W: trainer.py:93-97 through conv's backward = ratio planes x the Toeplitz operand of H."""
        e, c = self.eng, self._capi
        e.recon_ratio_w()
        e._gemm(e.gn, e.hut, c.EPI_F32, out=e.num_w, k_split=e.w_ksplit, m_rows=e.c_rows)
        neg = self._w_plane(e.num_w)
        if e.kl:   # beta == 1 back-propagates ones: sum_{b,j} H[b][r][j] for every (c, t) (all taps see the whole of H)
            pos = e.sum_h.view(1, e.R, *([1] * e.nd)).expand(e.W.shape).contiguous()
        else:
            e._gemm(e.gp, e.hut, c.EPI_F32, out=e.den_w, k_split=e.w_ksplit, m_rows=e.c_rows)
            pos = self._w_plane(e.den_w)
        return neg, pos

    def _h_fold(self, planes, rows_out, y_out, flat):
        e, c = self.eng, self._capi
        if e.h_rows:      # window-operand GEMM over shifted rows of the ratio planes, then the tap fold of its columns
            e._gemm_win(planes, rows_out)
            c.check(self.lib.nmfmu_conv_rows_fold(flat.data_ptr(), e.B, e.R, e.Lh // e.lhs[-1], e.lhs[-1], e.wk_fold,
                                                  rows_out.data_ptr(), e.wk_rows, self._s()), 'nmfmu_conv_rows_fold')
        else:             # Y [(r,t)][(b,l)] and the col2im sum of conv's backward
            e._gemm(e.wmt, planes, c.EPI_F32, out=y_out)
            c.check(self.lib.nmfmu_convnd_fold(flat.data_ptr(), e.B, e.R, e.nd, e._lh_arr, e._t_arr, y_out.data_ptr(), e.bl_pad,
                                               self._s()), 'nmfmu_convnd_fold')
        return flat.view(e.H.shape)

    def grads_h(self):
        e, c = self.eng, self._capi
        e._gemm(e.hu, e.wm, c.EPI_RATIO, x=e.x_h, gn=e.gnt, gp=e.gpt, n_rows=e.c_rows)
        neg = self._h_fold(e.gnt, e.hnum if e.h_rows else None, e.y, self.numh)
        if e.kl:
            pos = e.sum_w.view(1, e.R, *([1] * e.nd)).expand(e.H.shape).contiguous()
        else:
            pos = self._h_fold(e.gpt, e.hden if e.h_rows else None, e.y_den, self.denh)
        return neg, pos


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
        self.last_precision = None                  # the operand mode of the last fused binding ('chain': exact chain path)
        self._engines: Dict[Tuple, Tuple] = {}      # key -> (DenseMU | None, versions, target)

    # ------------------------------------------------------------------
    @staticmethod
    def _chain(pred):
        """(X0, [W1, ..., Wn], None) behind the closure's prediction ``X0 @ W1.T @ ... @ Wn.T``, or (H, [W], layer) for one
        convolutive layer, or raise."""
        if isinstance(pred, BaseComponent):            # deferred form: the layer itself
            node = (pred, pred.H, pred.W)
        else:
            node = getattr(pred, '_nmf_source', None)
            if node is None:
                raise NotImplementedError(
                    'BetaMu: the closure must return (target, prediction) where prediction is the output of a torchnmf_amd '
                    'NMF layer or of a chain of them (m(), nn.Sequential(...)(None), or the layer itself); general '
                    'autograd graphs are outside this engine')
        Ws: List = []
        while True:
            layer, H, W = node
            if isinstance(layer, (_nmf.NMFD, _nmf.NMF2D, _nmf.NMF3D)):
                if Ws or getattr(H, '_nmf_source', None) is not None:
                    raise NotImplementedError('BetaMu: a convolutive layer is supported on its own, not inside a chain')
                return H, [W], layer
            if not isinstance(layer, NMF):
                raise NotImplementedError(f'BetaMu: only NMF / NMFD / NMF2D / NMF3D layers are supported, got {type(layer).__name__}')
            assert H is not None and W is not None
            Ws.append(W)
            node = getattr(H, '_nmf_source', None)
            if node is None:
                return H, Ws[::-1], None

    def _engine(self, V_user, V, converted, H, W, beta, l1, l2) -> DenseMU:
        """DenseMU bound to (target, W, H), rebuilt when any of them is replaced and refreshed when W / H are edited in
        place.  ``V_user`` is the tensor the closure returned, ``V`` its detached fp32 contiguous form (``converted`` says
        whether that needed a copy).  The cache entry holds a reference to ``V_user`` and is keyed on that object's identity
        and ``_version`` -- never on the address of a converted temporary, whose storage the caching allocator hands out
        again -- and a target that had to be converted is packed afresh on every step INTO THE SAME
        BUFFERS (its source may have been edited in place without bumping anything we can see; the engine, its images
        and slabs are kept).  One engine per (target, W, H, beta, l1, l2): param groups with
        different hyper-parameters keep their own packed target instead of evicting each other."""
        if W.dtype != torch.float32 or H.dtype != torch.float32:
            raise NotImplementedError('BetaMu needs float32 factors (the update is applied in place to the parameters, whose '
                                      f'storage is the engine\'s fp32 master); got W {W.dtype}, H {H.dtype}')
        key = (id(V_user), tuple(V.shape), W.data_ptr(), H.data_ptr(), float(beta), float(l1), float(l2))
        hit = self._engines.get(key)
        versions = [V_user._version, W._version, H._version]
        if hit is not None and hit[2] is V_user and (converted or hit[1][0] == versions[0]):
            eng, seen, _ = hit
            if eng is None:                            # rank 129..256 without a parity-grade fused mode: chain path
                return None
            if converted:                              # same buffers, fresh contents: pack + validate, nothing else
                eng.repack_target(V)
                bad, _ = eng.target_flags()
                assert not bad, "Target should be non-negative."
            if seen[1:] != versions[1:]:               # someone else edited W / H since our last update
                eng.refresh_images()
                seen[1:] = versions[1:]
            return eng
        # drop bindings of other targets / other factor storage (a packed V is as large as V); same-target bindings of
        # other hyper-parameter sets stay
        for k in [k for k, v in self._engines.items() if v[2] is not V_user or k[2:4] != key[2:4]]:
            del self._engines[k]
        # precision='auto' resolves like NMF.fit's (one admission test, DenseMU.auto_mode): a single-plane fp16 mode
        # at 1x MFMA work where it meets the 1e-4 bar -- 'f16' for an fp16-exact target, else 'f16r' ('f16x' for beta == 2) -- else split
        # bf16 (rank <= 128).  Rank 129..256 without an admissible fp16 mode has no parity-grade fused mode: the binding is
        # remembered as None and step() takes the exact chain path.
        precision = self._precision
        if precision in (None, 'auto') and W.shape[1] > 128:
            from .engine import DEFAULT_BACKEND_FACTORY
            be = DEFAULT_BACKEND_FACTORY()
            precision = DenseMU.auto_mode(V, W.data, H.data, be.pad_rank(W.shape[1]), be, beta)
            if precision is None:
                self._engines[key] = (None, versions, V_user)
                self.last_precision = 'chain'
                return None
        if self._precision in (None, 'auto') and converted and W.shape[1] <= 128:
            #  a target that is re-packed from its source on every step may CHANGE between steps, and 'f16'
            # is admitted on the first one's contents (fp16-exact, in range): resolve 'auto' here and take the mode that keeps
            # the target in fp32 instead
            from .engine import DEFAULT_BACKEND_FACTORY
            be = DEFAULT_BACKEND_FACTORY()
            precision = DenseMU.auto_mode(V, W.data, H.data, be.pad_rank(W.shape[1]), be, beta) or 'bf16x3'
        if self._precision in (None, 'auto') and converted and precision == 'f16':
            precision = 'f16r' if float(beta) != 2.0 else 'f16x'   # 24-bit / fp32 target: nothing to admit per step
        eng = DenseMU(V, W.data, H.data, beta, l1, l2, precision=precision, allow_f16=True)
        self.last_precision = eng.precision_name       # what 'auto' resolved to (plain attribute, like NMF.last_precision)
        bad, _ = eng.target_flags()
        assert not bad, "Target should be non-negative."
        self._engines[key] = (eng, versions, V_user)
        return eng

    def _chain_step(self, V, X0, Ws, p, beta, l1, l2, ortho):
        """trainer.py:72-112 for parameter ``p`` of the chain prediction = X0 W1^T ... Wn^T (see the module docstring)."""
        from . import _capi
        from .engine import mu_gamma
        lib = _capi.load()
        s = torch.cuda.current_stream().cuda_stream
        R = NMF.reconstruct                                     # A @ B.T, exact-fp32 MFMA kernel
        xs = [X0.detach().float().contiguous()]
        for W in Ws:
            xs.append(R(xs[-1], W))
        pred = xs[-1]
        assert V.shape == pred.shape, f'target must be {tuple(pred.shape)}, got {tuple(V.shape)}'
        gn, gp = torch.empty_like(pred), torch.empty_like(pred)
        _capi.check(lib.nmfmu_mu_terms(pred.data_ptr(), V.data_ptr(), pred.numel(), float(beta), gn.data_ptr(),
                                       gp.data_ptr(), s), 'nmfmu_mu_terms')

        def back(G):
            k = len(Ws)
            while True:
                if p is Ws[k - 1]:                               # dW_k = G_k^T X_{k-1}
                    return R(G.t().contiguous(), xs[k - 1].t().contiguous())
                G = R(G, Ws[k - 1].detach().t().contiguous())    # G_{k-1} = G_k W_k
                k -= 1
                if k == 0:
                    return G                                     # dX0
        neg, pos = back(gn), back(gp)
        assert neg.shape == p.shape
        _capi.check(lib.nmfmu_trainer_update(p.data.data_ptr(), p.shape[0], p.shape[1], neg.data_ptr(), pos.data_ptr(),
                                             float(l1), float(l2), float(ortho), mu_gamma(float(beta)), p.grad.data_ptr(),
                                             s), 'nmfmu_trainer_update')

    def _watch_range(self, eng):
        """ the fp16 modes are admitted on the data of the step that built the engine; a factor that later grows
        beyond 65504 is clamped in its fp16 image and the kernels say so in ``nmfmu_step.status``.  fit() reads that word at
        its loss checkpoints; the trainer has none, so every 16th update of an fp16-mode engine does (one small device read)
        and warns once."""
        if getattr(self, '_range_warned', False) or not hasattr(eng, 'left_f16_range'):
            return
        self._range_tick = getattr(self, '_range_tick', 0) + 1
        if self._range_tick % 16 == 0 and eng.left_f16_range():
            import warnings
            warnings.warn("torchnmf_amd: a factor grew beyond fp16's range (65504) during BetaMu steps; its fp16 operand image "
                          "is clamped from here on and the updates stop following the reference.  Construct the optimizer with "
                          "precision='bf16x3' for data of this scale.", stacklevel=3)
            self._range_warned = True

    def _conv_step(self, V_user, V, converted, H, W, p, beta, l1, l2, ortho):
        """trainer.py:72-112 for parameter ``p`` of ONE convolutive layer (see the module docstring)."""
        from . import _capi
        from .engine import mu_gamma
        key = ('conv', id(V_user), tuple(V.shape), W.data_ptr(), H.data_ptr(), float(beta))
        hit = self._engines.get(key)
        versions = [V_user._version, W._version, H._version]
        if hit is not None and hit[2] is V_user and not converted and hit[1][0] == versions[0]:
            b, seen, _ = hit
            if seen[1:] != versions[1:]:               # someone else edited W / H since our last update
                b.eng.refresh_images()
        else:
            for k in [k for k, v in self._engines.items() if v[2] is not V_user or k[0] != 'conv' or k[3:5] != key[3:5]]:
                del self._engines[k]
            b = _ConvBinding(V, W.data, H.data, beta, self._precision)
            bad, _ = b.eng.target_flags()
            assert not bad, "Target should be non-negative."
            self.last_precision = b.precision_name
        neg, pos = b.grads_w() if p is W else b.grads_h()
        grad = p.grad
        if ortho > 0:
            # the orthogonality term sums over dim 1 -- the rank axis of W (C, R, *T) and of H (B, R, *L) -- which the flat
            # [rows][cols] update kernel cannot see: p.grad = relu(pos) - relu(neg) and the term are formed here (both terms
            # are sums of non-negative products, so the kernel's relu of the augmented pos changes nothing)
            torch.sub(pos.relu(), neg.relu(), out=p.grad)
            pos = pos.relu().add_(p.data.sum(1, keepdim=True) - p.data, alpha=float(ortho))
            grad = None
        lib = _capi.load()
        _capi.check(lib.nmfmu_trainer_update(p.data.data_ptr(), p.shape[0], p.data[0].numel(), neg.data_ptr(), pos.data_ptr(),
                                             float(l1), float(l2), 0.0, mu_gamma(float(beta)),
                                             grad.data_ptr() if grad is not None else None,
                                             torch.cuda.current_stream().cuda_stream), 'nmfmu_trainer_update')
        # the operand images of the parameter that changed (its rank sums ride along)
        if p is W:
            b.eng._pack_w()
        else:
            b.eng._pack_h()
        self._engines[key] = (b, [V_user._version, W._version, H._version], V_user)

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
                    X0, Ws, conv = self._chain(pred)
                    names = [X0] + Ws
                    if not any(p is q for q in names):    # p does not feed this prediction (trainer.py:73-75)
                        p.requires_grad = False
                        continue
                    for t, what in [(V, 'BetaMu target')] + [(q, 'BetaMu factor') for q in names]:
                        _nmf._require_device(t, what)
                    V_user = V
                    V = V.detach()
                    converted = V.dtype != torch.float32 or not V.is_contiguous()
                    if converted:
                        V = V.float().contiguous()
                    for q in names:
                        if isinstance(q, torch.nn.Parameter) and not q.data.is_contiguous():
                            q.data = q.data.contiguous()
                    if p.grad is None or p.grad.shape != p.shape or not p.grad.is_contiguous():
                        p.grad = torch.empty_like(p.data)
                    # One layer runs on the fused kernels -- unless 'auto' has no parity-grade mode there: rank 129..256
                    # where the fp16 modes are not admissible (split bf16 stops at rank 128; _engine then answers None)
                    # and anything wider than the kernels' 256 take the exact chain path below, which has no rank limit
                    # (the reference's BetaMu has none either, trainer.py:72-112).
                    if conv is not None:
                        assert V.dim() == Ws[0].dim() and V.shape[0] == X0.shape[0] and V.shape[1] == Ws[0].shape[0] and \
                            all(l == lh + t - 1 for l, lh, t in zip(V.shape[2:], X0.shape[2:], Ws[0].shape[2:])), \
                            f'target shape {tuple(V.shape)} does not match the layer'
                        self._conv_step(V_user, V, converted, X0, Ws[0], p, beta, l1, l2, ortho)
                        p.requires_grad = False
                        continue
                    rank1 = Ws[0].shape[1] if len(Ws) == 1 else 0
                    fused_ok = len(Ws) == 1 and (rank1 <= 128 or (rank1 <= 256 and self._precision != 'bf16x3'))
                    eng = None
                    if fused_ok:
                        H, W = X0, Ws[0]
                        assert V.dim() == 2 and V.shape == (H.shape[0], W.shape[0]), \
                            f'target must be {(H.shape[0], W.shape[0])}, got {tuple(V.shape)}'
                        eng = self._engine(V_user, V, converted, H, W, beta, l1, l2)
                    if eng is not None:
                        eng.trainer_step('W' if p is W else 'H', ortho, p.grad)
                        self._watch_range(eng)
                    else:
                        self._chain_step(V, X0, Ws, p, beta, l1, l2, ortho)
                    p.requires_grad = False
        finally:
            for group in self.param_groups:
                for p in group['params']:
                    p.requires_grad = status_cache[id(p)]
        return None
