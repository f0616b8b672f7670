"""TEST-ONLY stand-in for torchnmf_amd.engine.HipBackend, backed by the CPU oracle.

It implements the backend methods the host code calls (same names, same buffer contracts: slabs,
packed all-reduce buffers, column sums, flags) on CPU tensors, so that the sharding / all-reduce /
fit-loop logic can run under pytest without a GPU (gloo, world_size 2).  The product never imports
this file; torchnmf_amd has no CPU path of its own.
"""
import torch

from oracle import mu_oracle as O


class OracleBackend:
    name = 'oracle-cpu (tests only)'

    def pad_rows(self, rows):
        return (rows + 255) // 256 * 256

    def pad_rank(self, rank):
        for r in (32, 64, 128, 256):
            if rank <= r:
                return r
        raise NotImplementedError('rank > 256')

    def supported(self, r_pad, precision):
        return precision in (0, 2, 3, 4) or r_pad <= 128  # like the library: single-plane modes (bf16, f16, f16x, f16r) at every rank pad, bf16x3 <= 128

    def block_rows(self, r_pad, precision, beta):
        return 128

    def choose_nsplit(self, m_pad, k_pad, block_rows, device):
        return 3  # > 1 on purpose: exercises the slab sum

    def alloc(self, nbytes, device):
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)

    # ---- "device" work -----------------------------------------------------------------------
    def pack_x(self, V, transpose, precision, block_rows, m_pad, k_pad, flags, out=None):
        if flags is not None:
            bad = int(not bool(torch.all(V >= 0)))
            mn = int(V.abs().min().view(torch.int32)) if V.numel() else 0x7f800000
            flags[0] = max(int(flags[0]), bad)
            flags[1] = min(int(flags[1]), mn)
        return (V.t() if transpose else V).contiguous().clone()

    def xp_rows(self, xp, r0, n, k_pad, precision):
        return xp[r0:r0 + n]

    def pack_factor(self, fac, rank, r_pad, precision):
        fac.colsum.zero_()
        fac.colsum[:rank] = fac.f.sum(0)

    def _terms(self, st):
        A, B, X = st.owner.f, st.panel.f, st.xp
        return O.mu_terms(X, A @ B.t(), float(st.struct.beta)), B

    def mu_partial(self, st):
        (gn, gp), B = self._terms(st)
        rows, rank = st.owner.f.shape
        num = st.slab_num.view(st.nsplit, st.owner.rows_pad, st.r_pad)
        num.zero_()
        full = gn @ B
        # spread over the slabs so that only their SUM is right
        num[0, :rows, :rank] = 0.25 * full
        num[st.nsplit - 1, :rows, :rank] += 0.75 * full
        if gp is not None:
            den = st.slab_den.view(st.nsplit, st.owner.rows_pad, st.r_pad)
            den.zero_()
            den[0, :rows, :rank] = gp @ B

    def slab_reduce(self, st, num_out, den_out):
        num_out.copy_(st.slab_num.view(st.nsplit, -1).sum(0))
        if den_out is not None:
            den_out.copy_(st.slab_den.view(st.nsplit, -1).sum(0))

    def mu_apply(self, st, num, den, nslab, kl_den):
        rows, rank = st.owner.f.shape
        if num is None:
            num, den, nslab = st.slab_num, st.slab_den, st.nsplit
        neg = num.view(nslab, st.owner.rows_pad, st.r_pad).sum(0)[:rows, :rank]
        if kl_den is not None:
            pos, closed = kl_den[:rank].clone(), True
        else:
            pos, closed = den.view(nslab, st.owner.rows_pad, st.r_pad).sum(0)[:rows, :rank], False
        new = O._apply(st.owner.f, neg, pos, closed, float(st.struct.gamma), float(st.struct.l1), float(st.struct.l2))
        st.owner.f.copy_(new)
        self.pack_factor(st.owner, rank, st.r_pad, 0)

    def trainer_apply(self, st, kl_den, ortho, grad):
        rows, rank = st.owner.f.shape
        neg = st.slab_num.view(st.nsplit, st.owner.rows_pad, st.r_pad).sum(0)[:rows, :rank]
        if kl_den is not None:
            pos = kl_den[:rank].clone().expand(rows, rank)
        else:
            pos = st.slab_den.view(st.nsplit, st.owner.rows_pad, st.r_pad).sum(0)[:rows, :rank]
        new, g = O.betamu_update(st.owner.f, neg, pos, float(st.struct.gamma), float(st.struct.l1),
                                 float(st.struct.l2), float(ortho))
        st.owner.f.copy_(new)
        if grad is not None:
            grad.copy_(g)
        self.pack_factor(st.owner, rank, st.r_pad, 0)

    def loss(self, st, loss_part, out):
        out[0] = float(O.beta_div(st.owner.f @ st.panel.f.t(), st.xp, float(st.struct.beta)))
