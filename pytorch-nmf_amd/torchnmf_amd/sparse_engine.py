"""Engine for ``NMF.fit`` on a sparse-COO target (reference: nmf.py:351-398, 602-638), beta in {1, 2}.

The reference differentiates two scalars (``pos``, ``neg``) built from the stored entries of V; their gradients are
the dense numerator / denominator terms restricted to those entries, so the factor updates equal the dense ones
(tests/test_nmf_sparse.py:8-37).  Here:

  numerator    one HIP kernel per half-step over a CSR copy of V (H half-step) or V^T (W half-step): one wave per
               owner row, lanes across the rank, every panel row one coalesced read (nmfmu_sp_partial)
  denominator  beta 1: the closed-form column sums the dense path keeps (nmf.py:122-131)
               beta 2: owner @ (panel^T panel) -- the gradient of pos = 1/2 <H W^T W, H> (nmf.py:616-617)
  apply        the dense path's nmfmu_mu_apply (nmf.py:78-92), fed with one numerator "slab"
  loss         V_norm + pos - neg exactly as nmf.py:172-181, 357, 397: the O(nnz) term in HIP (nmfmu_sp_loss_neg),
               the O(R^2) terms from the column sums / Gram matrices

Other beta raise NotImplementedError: their positive term is a dense N x C pass in the reference as well.
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

    graphable = False

    def __init__(self, V, W, H, beta, l1=0.0, l2=0.0, update_W=True, update_H=True):
        self.be = DEFAULT_BACKEND_FACTORY()
        self.lib = self.be.lib
        self.beta = float(beta)
        if self.beta not in (1.0, 2.0):
            raise NotImplementedError('sparse targets are implemented for beta in {1, 2}; for other beta the reference '
                                      'itself evaluates a dense N x C term per update (nmf.py:628-636)')
        assert V.is_sparse and V.dim() == 2
        V = V.coalesce()
        N, Cc = V.shape
        R = W.shape[1]
        assert W.shape == (Cc, R) and H.shape == (N, R)
        assert V._nnz() < 2 ** 31 and max(N, Cc) < 2 ** 31
        self.kl = self.beta == 1.0
        self.rank, self.r_pad = R, self.be.pad_rank(R)
        dev = V.device
        idx, vals = V.indices(), V.values().float()
        self.bad = bool((~(vals >= 0)).any().item()) if vals.numel() else False    # nmf.py:329-330
        self.has_zero = bool(V._nnz() < N * Cc or (vals == 0).any().item())
        self.csr_h = _csr(idx[0], idx[1], vals, N)       # owner = rows of V  (H half-step, loss)
        self.csr_w = _csr(idx[1], idx[0], vals, Cc)      # owner = rows of V^T (W half-step)
        self.vals = vals
        prec = _capi.PREC_BF16                            # only selects which images the apply kernel refreshes
        self.fW = FactorBuf(W, self.r_pad, prec, self.be)
        self.fH = FactorBuf(H, self.r_pad, prec, self.be)
        gamma = mu_gamma(self.beta)
        mk = lambda own, pan: StepBuf(None, own, pan, R, self.r_pad, 1, prec, _capi.STAGE_DMA, 128, self.beta, gamma,
                                      l1, l2, need_den=not self.kl)
        self.step_h = mk(self.fH, self.fW)
        self.step_w = mk(self.fW, self.fH) if update_W else None
        self.gram = torch.empty(R * R, dtype=torch.float32, device=dev)
        self.gram2 = torch.empty(R * R, dtype=torch.float32, device=dev)
        self.gram_part = self.be.alloc(self.lib.nmfmu_gram_part_bytes(R), dev)
        self.loss_part = torch.empty((N + 3) // 4, dtype=torch.float64, device=dev)
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=dev)
        self.be.pack_factor(self.fW, R, self.r_pad, prec)   # column sums (beta == 1 denominators)
        self.be.pack_factor(self.fH, R, self.r_pad, prec)
        # nmf.py:172-181
        self.v_norm = float((vals.double() @ vals.double().log() - vals.double().sum()).item()) if self.kl else \
            float((vals.double() @ vals.double()).item() * 0.5)

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
                                              st.slab_num.data_ptr(), self.r_pad, self._s()), 'nmfmu_sp_partial')
        if self.kl:
            self.be.mu_apply(st, st.slab_num, None, 1, pan.colsum)
        else:
            _capi.check(self.lib.nmfmu_gram(pan.f.data_ptr(), pan.rows, self.rank, self.gram_part.data_ptr(),
                                            self.gram.data_ptr(), self._s()),
                        'nmfmu_gram')
            _capi.check(self.lib.nmfmu_rowmat(own.f.data_ptr(), own.rows, self.rank, self.gram.data_ptr(),
                                              st.slab_den.data_ptr(), self.r_pad, self._s()), 'nmfmu_rowmat')
            self.be.mu_apply(st, st.slab_num, st.slab_den, 1, None)

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
        else:            # pos = 1/2 <H W^T W, H> = 1/2 sum(H^T H * W^T W)
            for f, g in ((self.fH, self.gram), (self.fW, self.gram2)):
                _capi.check(self.lib.nmfmu_gram(f.f.data_ptr(), f.rows, self.rank, self.gram_part.data_ptr(), g.data_ptr(),
                                                self._s()), 'nmfmu_gram')
            pos = 0.5 * float((self.gram.double() @ self.gram2.double()).item())
        return self.v_norm + pos - float(self.loss_out.item())
