"""Engine for ``NMF.fit`` on a sparse-COO target (reference: nmf.py:351-398, 602-638), beta > 0.

The reference differentiates two scalars (``pos``, ``neg``) built from the stored entries of V; their gradients are
the dense numerator / denominator terms restricted to those entries, so the factor updates equal the dense ones
(tests/test_nmf_sparse.py:8-37).  Here:

  numerator    one HIP kernel per half-step over a CSR copy of V (H half-step) or V^T (W half-step): one wave per
               owner row, lanes across the rank, every panel row one coalesced read (nmfmu_sp_partial)
  denominator  beta 1: the closed-form column sums the dense path keeps (nmf.py:122-131)
               beta 2: owner @ (panel^T panel) -- the gradient of pos = 1/2 <H W^T W, H> (nmf.py:616-617)
               other : the reference's positive term sum (W H^T + eps)^beta / beta runs over EVERY entry (nmf.py:628-636),
                       so its gradient is a dense pass too: the fused MFMA kernel in its target-less denominator mode
                       (nmfmu_den_partial), fp32-grade split-bf16 operands up to rank 128
  apply        the dense path's nmfmu_mu_apply (nmf.py:78-92), fed with one numerator "slab"
  loss         V_norm + pos - neg exactly as nmf.py:172-181, 357, 397: the O(nnz) term in HIP (nmfmu_sp_loss_neg),
               the O(R^2) terms from the column sums / Gram matrices

beta <= 0 is rejected like in the reference (nmf.py:332-336: a sparse target always contains zeros).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi
from .engine import DEFAULT_BACKEND_FACTORY, FactorBuf, StepBuf, _ptr, mu_gamma


def _csr(rows: torch.Tensor, cols: torch.Tensor, vals: torch.Tensor, n_rows: int):
    """(rowptr int32, colidx int32, vals) sorted by (row, col).  One-time set-up on the device."""
    order = torch.argsort(rows * (int(cols.max()) + 1 if cols.numel() else 1) + cols)
    rows, cols, vals = rows[order], cols[order], vals[order]
    counts = torch.bincount(rows, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=rows.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32).contiguous(), cols.to(torch.int32).contiguous(), vals.contiguous()


class SparseMU:
    """Same interface as ``DenseMU`` (target_flags / w_step / h_step / divergence) for a sparse target."""


    def __init__(self, V, W, H, beta, l1=0.0, l2=0.0, update_W=True, update_H=True):
        self.be = DEFAULT_BACKEND_FACTORY()
        self.lib = self.be.lib
        self.beta = float(beta)
        if not self.beta > 0:
            raise ValueError('When beta <= 0 and V contains zeros, the training process may diverge. '
                             'Please add small values to V, or use a positive beta value.')
        assert V.is_sparse and V.dim() == 2
        V = V.coalesce()
        N, Cc = V.shape
        R = W.shape[1]
        assert W.shape == (Cc, R) and H.shape == (N, R)
        assert V._nnz() < 2 ** 31 and max(N, Cc) < 2 ** 31
        self.kl = self.beta == 1.0
        self.generic = self.beta not in (1.0, 2.0)
        self.rank, self.r_pad = R, self.be.pad_rank(R)
        dev = V.device
        idx, vals = V.indices(), V.values().float()
        self.bad = bool((~(vals >= 0)).any().item()) if vals.numel() else False    # nmf.py:329-330
        self.has_zero = bool(V._nnz() < N * Cc or (vals == 0).any().item())
        self.csr_h = _csr(idx[0], idx[1], vals, N)       # owner = rows of V  (H half-step, loss)
        self.csr_w = _csr(idx[1], idx[0], vals, Cc)      # owner = rows of V^T (W half-step)
        self.vals = vals
        # beta in {1, 2}: the images are never read (bf16 keeps them small).  Generic beta: the dense denominator pass
        # reads them -- fp32-grade split-bf16 where the fused kernel has it (padded rank <= 128), else bf16.
        prec = _capi.PREC_BF16
        if self.generic and self.be.supported(self.r_pad, _capi.PREC_BF16X3):
            prec = _capi.PREC_BF16X3
        self.prec = prec
        self.fW = FactorBuf(W, self.r_pad, prec, self.be)
        self.fH = FactorBuf(H, self.r_pad, prec, self.be)
        gamma = mu_gamma(self.beta)
        ns = {'h': 1, 'w': 1}
        if self.generic:
            ns = {'h': self.be.choose_nsplit(self.fH.rows_pad, self.fW.rows_pad, 128, dev),
                  'w': self.be.choose_nsplit(self.fW.rows_pad, self.fH.rows_pad, 128, dev)}

        def mk(own, pan, nsplit):
            st = StepBuf(None, own, pan, R, self.r_pad, nsplit, prec, _capi.STAGE_DMA, 128, self.beta, gamma, l1, l2,
                         need_den=not self.kl)
            st.num1 = torch.empty(st.plane, dtype=torch.float32, device=dev)   # the sparse numerator: one slab
            st.den1 = torch.empty(st.plane, dtype=torch.float32, device=dev) if not self.kl else None
            return st
        self.step_h = mk(self.fH, self.fW, ns['h'])
        self.step_w = mk(self.fW, self.fH, ns['w']) if update_W else None
        self.gram = torch.empty(R * R, dtype=torch.float32, device=dev)
        self.gram2 = torch.empty(R * R, dtype=torch.float32, device=dev)
        self.gram_part = self.be.alloc(self.lib.nmfmu_gram_part_bytes(R), dev)
        self.loss_part = torch.empty((N + 3) // 4, dtype=torch.float64, device=dev)
        self.dloss_part = torch.empty(max((self.fH.rows_pad // 128) * ns['h'], 1), dtype=torch.float32, device=dev)
        self.dloss_out = torch.zeros(1, dtype=torch.float64, device=dev)
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=dev)
        self.be.pack_factor(self.fW, R, self.r_pad, prec)   # column sums (beta == 1 denominators), operand images
        self.be.pack_factor(self.fH, R, self.r_pad, prec)
        vd = vals.double()                                  # nmf.py:172-181
        if self.kl:
            self.v_norm = float((vd @ vd.log() - vd.sum()).item())
        elif self.beta == 2.0:
            self.v_norm = float((vd @ vd).item() * 0.5)
        else:
            self.v_norm = float(vd.pow(self.beta).sum().item() / self.beta / (self.beta - 1))

    @staticmethod
    def _s() -> int:
        return torch.cuda.current_stream().cuda_stream

    def target_flags(self):
        return self.bad, self.has_zero

    def _half_step(self, st: StepBuf, csr):
        rowptr, colidx, vals = csr
        own, pan = st.owner, st.panel
        _capi.check(self.lib.nmfmu_sp_partial(rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), own.rows,
                                              own.f.data_ptr(), pan.f.data_ptr(), self.rank, self.beta,
                                              st.num1.data_ptr(), self.r_pad, self._s()), 'nmfmu_sp_partial')
        if self.kl:
            self.be.mu_apply(st, st.num1, None, 1, pan.colsum)
            return
        if self.generic:     # dense positive term on the fused kernel (no target), contraction-split slabs summed in order
            _capi.check(self.lib.nmfmu_den_partial(C.byref(st.struct), self._s()), 'nmfmu_den_partial')
            self.be.slab_reduce(st, st.den1, None)
        else:
            _capi.check(self.lib.nmfmu_gram(pan.f.data_ptr(), pan.rows, self.rank, self.gram_part.data_ptr(),
                                            self.gram.data_ptr(), self._s()), 'nmfmu_gram')
            _capi.check(self.lib.nmfmu_rowmat(own.f.data_ptr(), own.rows, self.rank, self.gram.data_ptr(),
                                              st.den1.data_ptr(), self.r_pad, self._s()), 'nmfmu_rowmat')
        self.be.mu_apply(st, st.num1, st.den1, 1, None)

    def w_step(self):
        self._half_step(self.step_w, self.csr_w)

    def h_step(self):
        self._half_step(self.step_h, self.csr_h)

    def divergence(self) -> float:
        """V_norm + pos - neg (nmf.py:357, 397).  One host sync."""
        rowptr, colidx, vals = self.csr_h
        _capi.check(self.lib.nmfmu_sp_loss_neg(rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), self.fH.rows,
                                               self.fH.f.data_ptr(), self.fW.f.data_ptr(), self.rank, self.beta,
                                               self.loss_part.data_ptr(), self.loss_out.data_ptr(), self._s()),
                    'nmfmu_sp_loss_neg')
        if self.kl:      # pos = W.sum(0) . H.sum(0)
            pos = float((self.fW.colsum[:self.rank].double() @ self.fH.colsum[:self.rank].double()).item())
        elif self.generic:   # pos = sum (H W^T + eps)^beta / beta over every entry: the fused loss mode without a target
            self.be.loss(self.step_h, self.dloss_part, self.dloss_out)
            pos = float(self.dloss_out.item())
        else:            # pos = 1/2 <H W^T W, H> = 1/2 sum(H^T H * W^T W)
            for f, g in ((self.fH, self.gram), (self.fW, self.gram2)):
                _capi.check(self.lib.nmfmu_gram(f.f.data_ptr(), f.rows, self.rank, self.gram_part.data_ptr(), g.data_ptr(),
                                                self._s()), 'nmfmu_gram')
            pos = 0.5 * float((self.gram.double() @ self.gram2.double()).item())
        return self.v_norm + pos - float(self.loss_out.item())
