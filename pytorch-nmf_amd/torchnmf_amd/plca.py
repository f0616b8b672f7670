"""``PLCA`` -- probabilistic latent component analysis ``V / V.sum() ~ H diag(Z) W^T`` fitted by EM on the MI355X
(reference: plca.py:22-373).

One EM iteration needs ``G = Vn / (H diag(Z) W^T + eps)``, ``G^T H`` and ``G W`` -- the same fused reconstruction ->
ratio -> contraction the NMF beta = 1 half-step performs, with the latent weights folded into the panel of the first
GEMM only.  Both contractions run on ``nmfmu::fused_kernel`` (through ``nmfmu_mu_partial`` with a split panel: image of
the Z-scaled factor for the reconstruction, image of the unscaled factor for the second GEMM); the O((N + C) R)
remainder of plca.py:248-290 (multiply by relu(grad), divide by the latent prior, Dirichlet prior, renormalise) runs in
the small ``nmfmu_plca_*`` kernels, and only the R-element latent vector is updated with torch scalar ops.
``SIPLCA`` / ``SIPLCA2`` / ``SIPLCA3`` (plca.py:376-606) are to PLCA what NMFD / NMF2D / NMF3D are to NMF: the same EM
update on the NMFD GEMM engine (``_ConvPlcaEM``), with W * Z as the reconstruction operand.
"""
from collections.abc import Iterable
from typing import Optional

import torch
from torch import Tensor, nn

from . import _capi
from .constants import eps as _EPS
from .engine import DEFAULT_BACKEND_FACTORY, FactorBuf, StepBuf, _ptr
from .nmf import _require_device, _sqrt2

__all__ = ['PLCA', 'SIPLCA', 'SIPLCA2', 'SIPLCA3', 'BaseComponent']


def get_norm(x: Tensor) -> Tensor:
    """plca.py:27-35: sum over every axis but 1 (a vector: its total)."""
    if x.ndim > 1:
        return x.sum([d for d in range(x.dim()) if d != 1], keepdim=True)
    return x.sum()


def _new(spec, trainable, label):
    if isinstance(spec, Tensor):
        assert bool(torch.all(spec >= 0.)), f"Tensor {label} should be non-negative."
        p = nn.Parameter(torch.empty(*spec.size()), requires_grad=trainable)
        p.data.copy_(spec)
    elif isinstance(spec, Iterable):
        p = nn.Parameter(torch.randn(*tuple(spec)).abs())
    else:
        return None
    p.data.div_(get_norm(p.data))          # plca.py:91, 105
    return p


def _scalar_alpha(a, name: str, factor: Tensor) -> float:
    """A Dirichlet hyper-parameter as the reference's EM loop can use it (plca.py:257-259, 271-273, 285-287): a number,
    or a tensor with exactly one element and no more dimensions than the factor it is added to in place."""
    if isinstance(a, Tensor):
        if a.numel() != 1:
            raise RuntimeError(f'Boolean value of Tensor with more than one value is ambiguous ({name} has '
                               f'{a.numel()} elements; the reference evaluates `if {name} != 1`, plca.py:257-285)')
        if a.dim() > factor.dim():
            raise RuntimeError(f"output with shape {list(factor.shape)} doesn't match the broadcast shape "
                               f"{[1] * (a.dim() - factor.dim()) + list(factor.shape)} ({name} has more dimensions than the "
                               f"factor the reference adds it to in place)")
        return float(a.detach().reshape(()).item())
    return float(a)


class BaseComponent(nn.Module):
    """Base of the PLCA modules (plca.py:38-191): W, H normalised over everything but the rank axis, Z a distribution."""

    def __init__(self, rank=None, W=None, H=None, Z=None, trainable_W=True, trainable_H=True, trainable_Z=True):
        super().__init__()
        self.register_parameter('W', _new(W, trainable_W, 'W'))
        self.register_parameter('H', _new(H, trainable_H, 'H'))
        infer = None
        if self.W is not None:
            infer = self.W.shape[1]
        if self.H is not None:
            infer = self.H.shape[1]
        if isinstance(Z, Tensor):
            assert Z.ndim == 1, "Z should be one dimensional."
            assert bool(torch.all(Z >= 0.)), "Tensor Z should be non-negative."
            z = nn.Parameter(torch.empty(Z.numel()), requires_grad=trainable_Z)
            z.data.copy_(Z)
        elif isinstance(rank, int):
            z = nn.Parameter(torch.ones(rank) / rank)
        else:
            z = None
        self.register_parameter('Z', z)
        if z is not None:
            z.data.div_(get_norm(z.data))   # plca.py:121
            infer = z.shape[0]
        if infer is None:
            assert rank, "A rank should be given when W, H and Z are not available!"
        else:
            if self.Z is not None:
                assert self.Z.shape[0] == infer, "Latent size of Z does not match with others!"
            if self.H is not None:
                assert self.H.shape[1] == infer, "Latent size of H does not match with others!"
            if self.W is not None:
                assert self.W.shape[1] == infer, "Latent size of W does not match with others!"
                self.out_channels = self.W.shape[0]
                if self.W.ndim > 2:
                    self.kernel_size = tuple(self.W.shape[2:])
            rank = infer
        self.rank = rank

    def extra_repr(self) -> str:
        s = f'{self.rank}'
        if self.W is not None:
            s += f', out_channels={self.out_channels}'
            if hasattr(self, 'kernel_size'):
                s += f', kernel_size={self.kernel_size}'
        return s

    def forward(self, H: Tensor = None, W: Tensor = None, Z: Tensor = None, norm: Optional[float] = None) -> Tensor:
        """plca.py:155-183: reconstruction with the module's own tensors substituted for missing arguments."""
        H = self.H if H is None else H
        W = self.W if W is None else W
        Z = self.Z if Z is None else Z
        out = self.reconstruct(H, W, Z)
        return out if norm is None else out * norm

    @staticmethod
    def reconstruct(H: Tensor, W: Tensor, Z: Tensor) -> Tensor:
        raise NotImplementedError

    def _make_em(self, Vn, precision):
        raise NotImplementedError

    @torch.no_grad()
    def fit(self, V, tol=1e-4, max_iter=200, verbose=False, W_alpha=1., H_alpha=1., Z_alpha=1., *, precision=None):
        """EM fit (plca.py:193-304).  Returns ``(n_iter, norm)`` like the reference: the index of the last iteration and
        ``V.sum()``.  ``W_alpha`` / ``H_alpha`` / ``Z_alpha``: floats or one-element tensors (what the reference's
        ``if alpha != 1`` admits); a tensor with more elements raises the same RuntimeError as there."""
        W, H, Z = self.W, self.H, self.Z
        assert W is not None and H is not None and Z is not None
        # Dirichlet hyper-parameters (plca.py:197-199 types them Union[float, Tensor]).  The reference tests them with
        # ``if W_alpha != 1:`` (plca.py:257, 271, 285), so a tensor works there exactly when it has ONE element -- a 0-dim
        # or 1-element tensor, possibly on another device -- and raises "Boolean value of Tensor with more than one value
        # is ambiguous" otherwise.  Same contract here: one-element tensors are taken by value, anything larger raises
        # the RuntimeError the reference's ``if`` raises.
        #  the reference evaluates ``if X_alpha != 1`` only inside ``if X.requires_grad`` (plca.py:255-287): the
        # hyper-parameter of a FROZEN factor is never looked at there, so it is neither validated nor converted here
        W_alpha, H_alpha, Z_alpha = (_scalar_alpha(a, n, f) if f.requires_grad else 1.0
                                     for a, n, f in ((W_alpha, 'W_alpha', W), (H_alpha, 'H_alpha', H), (Z_alpha, 'Z_alpha', Z)))
        for t_, what in ((V, 'fit'), (W, 'fit'), (H, 'fit'), (Z, 'fit')):
            _require_device(t_, what)
        V = V.detach().float()
        assert bool(torch.all(V >= 0.)), "Target should be non-negative."
        norm = V.sum()
        Vn = (V.contiguous() / norm).contiguous()
        for p in (W, H, Z):
            if not p.data.is_contiguous():
                p.data = p.data.contiguous()
        em = self._make_em(Vn, precision)
        nrm = float(norm.item())
        loss_init = previous = _sqrt2(nrm * em.divergence())      # kl_div(WZH * norm, V), plca.py:245-246
        pbar = None
        if verbose:
            from tqdm import tqdm
            pbar = tqdm(total=max_iter)
        n_iter = -1
        try:
            for n_iter in range(max_iter):
                em.em_step(W.requires_grad, H.requires_grad, Z.requires_grad, W_alpha, H_alpha, Z_alpha)
                if n_iter % 10 == 9:
                    loss = _sqrt2(nrm * em.divergence())
                    if pbar is not None:
                        pbar.set_postfix(loss=loss)
                        pbar.update(10)
                    if (previous - loss) / loss_init < tol:
                        break
                    previous = loss
        finally:
            if pbar is not None:
                pbar.close()
        return n_iter, norm


class _PlcaEM:
    """Device state of one PLCA fit: packed Vn (both orientations), factor images (scaled and unscaled), slabs."""

    def __init__(self, Vn, W, H, Z, precision):
        self.be = DEFAULT_BACKEND_FACTORY()
        self.lib = be = self.be.lib
        N, Cc = Vn.shape
        R = W.shape[1]
        self.R, self.r_pad = R, self.be.pad_rank(R)
        if precision in (None, 'auto'):
            # 'auto' = the fp32-grade split mode.  The single-plane fp16 modes NMF.fit's 'auto' takes are not admissible here
            # BY CONSTRUCTION: the factors are probability tables (columns of W sum to one: mean 1 / C, far below
            # DenseMU.F16_MIN_MEAN at any size where the contraction is long enough for fp16), W * Z sits in fp16's
            # subnormals, and with a reconstruction of order 1 / (N C) the eps of plca.py:250 dominates the ratio's
            # denominator -- a power-of-two rescaling of the images would move eps and is not exact.
            precision = 'bf16x3' if self.be.supported(self.r_pad, _capi.PREC_BF16X3) else 'bf16'
        if precision not in _capi.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_capi.PRECISIONS)} or 'auto', got {precision!r}")
        if precision in ('f16x', 'f16'):
            #  the target is normalised to sum 1 here and the factors are probability tables: both sit in
            # fp16's subnormals, so either fp16 mode would flush them (and 'f16x' has no split-panel instance at all)
            raise NotImplementedError(f"precision {precision!r} is not available for PLCA: the normalised target and factors "
                                      "sit below fp16's range -- use 'bf16x3' (the default) or 'bf16'")
        self.prec = _capi.PRECISIONS[precision]
        self.precision_name = precision
        if not self.be.supported(self.r_pad, self.prec):
            raise NotImplementedError(f'precision {precision!r} is not available for rank {R}')
        dev = Vn.device
        self.W, self.H, self.Z = W, H, Z
        mk = lambda t: FactorBuf(t, self.r_pad, self.prec, self.be)
        self.fW, self.fWz, self.fH, self.fHz = mk(W), mk(W), mk(H), mk(H)   # the *z buffers carry images of f * Z
        self.flags = torch.tensor([0, 0x7f800000], dtype=torch.int32, device=dev)
        n_pad, c_pad = self.fH.rows_pad, self.fW.rows_pad
        br = 128
        xp_h = self.be.pack_x(Vn, False, self.prec, br, n_pad, c_pad, self.flags)
        xp_w = self.be.pack_x(Vn, True, self.prec, br, c_pad, n_pad, None)
        ns_h = self.be.choose_nsplit(n_pad, c_pad, br, dev)
        ns_w = self.be.choose_nsplit(c_pad, n_pad, br, dev)

        def step(xp, owner, p1_src, p2_src, ns):
            st = StepBuf(xp, owner, p2_src, R, self.r_pad, ns, self.prec, _capi.STAGE_DMA_SPLIT, br, 1.0, 1.0, 0.0, 0.0,
                         need_den=False)
            # split panel: reconstruction from the Z-scaled images, second GEMM from the unscaled ones
            st.struct.panel.p1_hi, st.struct.panel.p1_lo = _ptr(p1_src.p1_hi), _ptr(p1_src.p1_lo)
            return st
        self.step_w = step(xp_w, self.fW, self.fHz, self.fH, ns_w)   # num'_W = G^T H
        self.step_h = step(xp_h, self.fH, self.fWz, self.fW, ns_h)   # num'_H = G W
        # loss: beta_div(H (W Z)^T, Vn, 1) -- owner H, panel entirely the scaled W
        self.step_l = StepBuf(xp_h, self.fH, self.fWz, R, self.r_pad, ns_h, self.prec, _capi.STAGE_DMA, br, 1.0, 1.0, 0.0,
                              0.0, need_den=False)
        self.loss_part = torch.empty(max((n_pad // br) * ns_h, 1), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=dev)
        nb = max(self.lib.nmfmu_plca_part_bytes(max(N, Cc), self.r_pad), 16)
        self.part = self.be.alloc(nb, dev)
        self.cs = torch.zeros(self.r_pad, dtype=torch.float32, device=dev)
        self.cs2 = torch.zeros(self.r_pad, dtype=torch.float32, device=dev)
        self.zg = torch.zeros(self.r_pad, dtype=torch.float32, device=dev)
        self.zpad = torch.zeros(self.r_pad, dtype=torch.float32, device=dev)   # Z padded to r_pad for the kernels
        self.prior = torch.ones(self.r_pad, dtype=torch.float32, device=dev)   # what the factors' normalisation divides by
        self.ones = torch.ones(self.r_pad, dtype=torch.float32, device=dev)
        self.repack()

    def _s(self):
        return torch.cuda.current_stream().cuda_stream

    def repack(self):
        self.zpad[:self.R] = self.Z
        # (both image sets through the scaled entry -- scale = ones for the plain one: it skips the column-sum finalize
        # launches, whose results the EM iteration never reads)
        for plain, scaled in ((self.fW, self.fWz), (self.fH, self.fHz)):
            for fb, sc in ((plain, self.ones), (scaled, self.zpad)):
                _capi.check(self.lib.nmfmu_pack_factor_scaled(plain_struct(fb), self.R, self.r_pad, self.prec,
                                                              sc.data_ptr(), self._s()), 'nmfmu_pack_factor_scaled')

    def divergence(self) -> float:
        self.be.loss(self.step_l, self.loss_part, self.loss_out)
        return float(self.loss_out.item())

    def _em(self, f, st, z_old, update, want_zgrad):
        """f *= relu(num * z_old) (when ``update``); column sums of the result -> self.cs; Z.grad partials -> self.zg."""
        _capi.check(self.lib.nmfmu_plca_em(f.data_ptr(), f.shape[0], self.R, self.r_pad, st.slab_num.data_ptr(), st.nsplit,
                                           st.owner.rows_pad, z_old.data_ptr(), int(update), self.part.data_ptr(),
                                           self.cs.data_ptr(), self.zg.data_ptr() if want_zgrad else None, self._s()),
                    'nmfmu_plca_em')

    def _normalize(self, f, alpha):
        """plca.py:265-275 / 279-289 after the multiplication: divide by the latent prior held in ``self.prior``."""
        _capi.check(self.lib.nmfmu_plca_normalize(f.data_ptr(), f.shape[0], self.R, self.r_pad, self.prior.data_ptr(),
                                                  float(alpha), self.part.data_ptr(), self.cs2.data_ptr(), self._s()),
                    'nmfmu_plca_normalize')
        if alpha != 1:
            _capi.check(self.lib.nmfmu_plca_scale(f.data_ptr(), f.shape[0], self.R, self.cs2.data_ptr(), self._s()),
                        'nmfmu_plca_scale')

    def em_step(self, tW, tH, tZ, W_alpha, H_alpha, Z_alpha):
        """One EM iteration (plca.py:248-290): every update uses the gradients of ONE reconstruction."""
        self.be.mu_partial(self.step_w)
        self.be.mu_partial(self.step_h)
        z_old = self.zpad                               # (still the old Z: repack() below refreshes it)
        # W's multiplication pass also yields Z.grad = sum W_old * (G^T H); the division by the latent prior, which needs
        # the new Z first (plca.py:253-270), is the separate normalize kernel
        self._em(self.W.data, self.step_w, z_old, tW, True)
        have_prior = False
        if tZ:                                          # plca.py:253-260: one launch; prior = Z_old * relu(Z.grad)
            _capi.check(self.lib.nmfmu_plca_z(self.Z.data.data_ptr(), self.zg.data_ptr(), self.R, float(Z_alpha),
                                              self.prior.data_ptr(), self._s()), 'nmfmu_plca_z')
            have_prior = True
        if tW:
            if not have_prior:                          # frozen Z: get_norm of the multiplied W (plca.py:266-268)
                self.prior[:self.R].copy_(self.cs[:self.R])
                have_prior = True
            self._normalize(self.W.data, W_alpha)
        if tH:
            self._em(self.H.data, self.step_h, z_old, True, False)
            if not have_prior:                          # frozen Z and W: get_norm of the multiplied H
                self.prior[:self.R].copy_(self.cs[:self.R])
            self._normalize(self.H.data, H_alpha)
        self.repack()


def plain_struct(fb: FactorBuf):
    import ctypes
    return ctypes.byref(fb.struct)


class PLCA(BaseComponent):
    """``V ~ H diag(Z) W^T`` with V (N, C), W (C, R), H (N, R), Z (R,) (reference: plca.py:311-373)."""

    def __init__(self, Vshape=None, rank=None, **kwargs):
        if isinstance(Vshape, Iterable):
            M, K = Vshape
            rank = rank if rank else K
            kwargs['W'] = (K, rank)
            kwargs['H'] = (M, rank)
        super().__init__(rank, **kwargs)

    @staticmethod
    def reconstruct(H: Tensor, W: Tensor, Z: Tensor) -> Tensor:
        """``H @ (W * Z).T`` (plca.py:371-373) on the device (exact-fp32 MFMA kernel of NMF.reconstruct)."""
        from .nmf import NMF
        return NMF.reconstruct(H, W.detach() * Z.detach())

    def _make_em(self, Vn, precision):
        assert Vn.dim() == 2 and Vn.shape == (self.H.shape[0], self.W.shape[0])
        return _PlcaEM(Vn, self.W.data, self.H.data, self.Z.data, precision)


class _ConvPlcaEM:
    """EM state of a shift-invariant PLCA fit on the NMFD GEMM engine (``ConvMU`` supplies the packed targets, the
    Toeplitz operands of H and the GEMM plumbing).  Per iteration: W * Z planes -> reconstruction + ratio GEMMs ->
    unscaled numerators (G2; G4 + fold) -> the ``nmfmu_plca3`` update kernels (plca.py:248-290)."""

    def __init__(self, Vn, W, H, Z, precision):
        from .nmfd_engine import ConvMU, _Planes
        import ctypes as C
        self.C = C
        self.eng = e = ConvMU(Vn, W, H, 1.0, precision=precision, own_loop=False)
        self.lib = e.lib
        self.W, self.H, self.Z = W, H, Z
        x3 = e.precision == _capi.PREC_BF16X3
        dev = Vn.device
        self.wm_s = _Planes(e.c_pad, e.rp_pad, x3, dev)      # W * Z, [c][(r,t)]: the reconstruction operand
        self.wmt_s = _Planes(e.rp_pad, e.c_pad, x3, dev)     # (written alongside, unused)
        self.numh = torch.empty(e.B * e.R * e.Lh, dtype=torch.float32, device=dev)
        self.part = torch.empty(max(self.lib.nmfmu_plca3_part_bytes(e.R) // 4, 4), dtype=torch.float32, device=dev)
        self.cs = torch.zeros(e.R, dtype=torch.float32, device=dev)
        self.cs2 = torch.zeros(e.R, dtype=torch.float32, device=dev)
        self.zg = torch.zeros(e.R, dtype=torch.float32, device=dev)
        self.div = torch.ones(e.R, dtype=torch.float32, device=dev)
        self.repack()

    @staticmethod
    def _s():
        return torch.cuda.current_stream().cuda_stream

    def repack(self):
        e = self.eng
        e.refresh_images()                                   # unscaled Wm / WmT planes, H operands, rank sums
        _capi.check(self.lib.nmfmu_conv_pack_w_scaled(self.W.data_ptr(), e.C, e.R, e.T, self.Z.data_ptr(), e.c_pad,
                                                      e.rp_pad, _ptr(self.wm_s.hi), _ptr(self.wm_s.lo),
                                                      _ptr(self.wmt_s.hi), _ptr(self.wmt_s.lo), self._s()),
                    'nmfmu_conv_pack_w_scaled')

    def divergence(self) -> float:
        e = self.eng
        e._gemm(self.wm_s, e.hu, _capi.EPI_LOSS, x=e.x_w, out=e.loss_part, m_valid=e.C, n_valid=e.B * e.L, m_rows=e.c_rows)
        return float(e.loss_part.double().sum().item())

    def _plca3(self, mode, f, outer, inner, num, pitch, vec, alpha, update, want_zg):
        e = self.eng
        _capi.check(self.lib.nmfmu_plca3(mode, f.data_ptr(), outer, e.R, inner, _ptr(num), pitch, vec.data_ptr(),
                                         float(alpha), int(update), self.part.data_ptr(),
                                         (self.cs if mode == 0 else self.cs2).data_ptr(),
                                         self.zg.data_ptr() if want_zg else None, self._s()), 'nmfmu_plca3')

    def _factor_update(self, f, outer, inner, num, pitch, z_old, trainable, alpha, z_prior):
        if not trainable:
            return z_prior
        self._plca3(0, f, outer, inner, num, pitch, z_old, 1.0, True, False)
        if z_prior is None:
            z_prior = self.cs.clone()
        self.div.copy_(z_prior)
        self._plca3(1, f, outer, inner, None, 0, self.div, alpha, True, False)
        if alpha != 1:
            self._plca3(2, f, outer, inner, None, 0, self.cs2, 1.0, True, False)
        return z_prior

    def em_step(self, tW, tH, tZ, W_alpha, H_alpha, Z_alpha):
        e = self.eng
        # one reconstruction (twice, once per output layout) feeds every update of the iteration
        e._gemm(self.wm_s, e.hu, _capi.EPI_RATIO, x=e.x_w, gn=e.gn, m_rows=e.c_rows)       # Gn [c][(b,l)]
        e._gemm(e.gn, e.hut, _capi.EPI_F32, out=e.num_w, k_split=e.w_ksplit, m_rows=e.c_rows)   # (G^T H) [c][(r,t)], unscaled
        if e.w_ksplit > 1:                                                      # split-K partials -> slab 0
            _capi.check(self.lib.nmfmu_slab_sum(e.num_w.data_ptr(), (e.c_rows or e.c_pad) * e.rp_pad, e.w_ksplit, self._s()),
                        'nmfmu_slab_sum')
        e._gemm(e.hu, self.wm_s, _capi.EPI_RATIO, x=e.x_h, gn=e.gnt, n_rows=e.c_rows)       # Gn [(b,l)][c]
        if e.h_rows:
            # (G W) without the unfolded Y: the window-operand GEMM over shifted rows of Gn^T with the unscaled W
            e._gemm_win(e.gnt, e.hnum)
            _capi.check(self.lib.nmfmu_conv_rows_fold(self.numh.data_ptr(), e.B, e.R, e.Lh // e.lhs[-1], e.lhs[-1], e.wk_fold,
                                                      e.hnum.data_ptr(), e.wk_rows, self._s()), 'nmfmu_conv_rows_fold')
        else:
            e._gemm(e.wmt, e.gnt, _capi.EPI_F32, out=e.y)                       # Y [(r,t)][(b,l)] from the unscaled W
            _capi.check(self.lib.nmfmu_convnd_fold(self.numh.data_ptr(), e.B, e.R, e.nd, e._lh_arr, e._t_arr, e.y.data_ptr(),
                                                   e.bl_pad, self._s()), 'nmfmu_convnd_fold')
        z_old = self.Z.clone()
        # Z.grad = sum W * (G^T H): the em pass over W with update = 0
        self._plca3(0, self.W, e.C, e.T, e.num_w, e.rp_pad, z_old, 1.0, False, True)
        z_prior = None
        if tZ:
            z1 = self.Z * self.zg.relu()
            z_prior = z1.clone()
            if Z_alpha != 1:
                z1 = z1 + (Z_alpha - 1)
                z1 = torch.where(z1 > _EPS, z1, torch.full_like(z1, _EPS))
            self.Z.copy_(z1 / z1.sum())
        z_prior = self._factor_update(self.W, e.C, e.T, e.num_w, e.rp_pad, z_old, tW, W_alpha, z_prior)
        self._factor_update(self.H, e.B, e.Lh, self.numh, e.R * e.Lh, z_old, tH, H_alpha, z_prior)
        self.repack()


def _conv_reconstruct(H: Tensor, W: Tensor, Z: Tensor) -> Tensor:
    from .nmfd_engine import reconstruct
    return reconstruct(H, W.detach() * Z.detach().view(1, -1, *([1] * (W.dim() - 2))))


class _ShiftInvariant(BaseComponent):
    def _make_em(self, Vn, precision):
        assert Vn.dim() == self.W.dim() and Vn.shape[0] == self.H.shape[0] and Vn.shape[1] == self.W.shape[0]
        for q in (self.W, self.H, self.Z):              # the EM engine works on the parameters' storage in place
            if not q.data.is_contiguous():
                q.data = q.data.contiguous()
        return _ConvPlcaEM(Vn, self.W.data, self.H.data, self.Z.data, precision)

    @staticmethod
    def reconstruct(H: Tensor, W: Tensor, Z: Tensor) -> Tensor:
        """``convNd(H, W.flip * Z, padding = T - 1)`` (plca.py:447-449, 522-525, 602-605)."""
        return _conv_reconstruct(H, W, Z)


def _ntuple(x, n):
    return tuple(x) if isinstance(x, Iterable) else (x,) * n


class SIPLCA(_ShiftInvariant):
    """Shift-invariant PLCA, V (B, C, L), W (C, R, T), H (B, R, L - T + 1) (reference: plca.py:376-449)."""

    def __init__(self, Vshape=None, rank=None, T=1, **kwargs):
        if isinstance(Vshape, Iterable):
            T, = _ntuple(T, 1)
            batch, K, M = Vshape
            rank = rank if rank else K
            kwargs['W'] = (K, rank, T)
            kwargs['H'] = (batch, rank, M - T + 1)
        super().__init__(rank, **kwargs)


class SIPLCA2(_ShiftInvariant):
    """2-D shift-invariant PLCA (reference: plca.py:452-525)."""

    def __init__(self, Vshape=None, rank=None, kernel_size=1, **kwargs):
        if isinstance(Vshape, Iterable):
            ks = _ntuple(kernel_size, 2)
            batch, channel, K, M = Vshape
            rank = rank if rank else K
            kwargs['W'] = (channel, rank) + ks
            kwargs['H'] = (batch, rank, K - ks[0] + 1, M - ks[1] + 1)
        super().__init__(rank, **kwargs)


class SIPLCA3(_ShiftInvariant):
    """3-D shift-invariant PLCA (reference: plca.py:528-605)."""

    def __init__(self, Vshape=None, rank=None, kernel_size=1, **kwargs):
        if isinstance(Vshape, Iterable):
            ks = _ntuple(kernel_size, 3)
            batch, channel, N, K, M = Vshape
            rank = rank if rank else K
            kwargs['W'] = (channel, rank) + ks
            kwargs['H'] = (batch, rank, N - ks[0] + 1, K - ks[1] + 1, M - ks[2] + 1)
        super().__init__(rank, **kwargs)
