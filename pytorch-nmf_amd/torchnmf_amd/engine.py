"""Host-side orchestration of the dense MU iteration on one MI355X (optionally one shard of many).

This is the part of ``torchnmf.nmf.BaseComponent.fit`` (nmf.py:297-409 of the
reference) below the Python loop: buffers, packing, and the two half-steps,
expressed as calls into the C ABI (include/nmfmu.h).  PyTorch is used for
device memory, the current stream and ``torch.distributed`` only.

Column sharding: every rank holds ``V[:, Cg]`` and
``W[Cg]``; ``H`` is replicated.  The W half-step is local; the H half-step
all-reduces one packed fp32 buffer ``[numerator | denominator]`` per iteration.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _capi


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class HipBackend:
    """The real backend: libnmfmu.so on the current ROCm device.  (Tests may substitute an object with
    the same methods to exercise the host logic without a GPU; the product never does.)"""

    name = 'hip'

    def __init__(self):
        self.lib = _capi.load()
        if not torch.cuda.is_available():
            raise _capi.NmfmuError('torchnmf_amd needs a ROCm device (MI355X); none is visible and there is no '
                                   'CPU fallback')

    # -- static queries
    def pad_rows(self, rows: int) -> int:
        return self.lib.nmfmu_pad_rows(rows)

    def pad_rank(self, rank: int) -> int:
        r = self.lib.nmfmu_pad_rank(rank)
        _capi.check(r if r < 0 else 0, f'rank {rank}')
        return r

    def supported(self, r_pad: int, precision: int) -> bool:
        return bool(self.lib.nmfmu_supported(r_pad, precision))

    def block_rows(self, r_pad: int, precision: int, beta: float) -> int:
        return self.lib.nmfmu_block_rows(r_pad, precision, beta)

    def step_block_rows(self, m_pad: int, k_pad: int, r_pad: int, precision: int, beta: float, device) -> int:
        ncu = torch.cuda.get_device_properties(device).multi_processor_count
        br = self.lib.nmfmu_step_block_rows(m_pad, k_pad, r_pad, precision, beta, ncu)
        if br not in (128, 256):
            _capi.check(br if br < 0 else _capi.ERR_ARG, 'nmfmu_step_block_rows')
        return br

    def choose_nsplit(self, m_pad: int, k_pad: int, block_rows: int, device, r_pad=None, precision=None, beta=None) -> int:
        """Contraction split of one half-step.  With (r_pad, precision, beta) the library sizes it for the kernel that will
        run (the software-pipelined rank-256 kernel holds ONE workgroup per CU, the four-wave kernel two)."""
        ncu = torch.cuda.get_device_properties(device).multi_processor_count
        forced = os.environ.get('TORCHNMF_AMD_NSPLIT')        # experiment hook (tools/gpu_r6*.sh)
        if forced:
            return max(1, min(int(forced), k_pad // 64 // 4))
        if r_pad is not None:
            return self.lib.nmfmu_choose_nsplit_for(m_pad, k_pad, r_pad, precision, beta, block_rows, ncu)
        return self.lib.nmfmu_choose_nsplit(m_pad, k_pad, block_rows, ncu)

    def kernel_family(self, r_pad: int, precision: int, beta: float) -> int:
        return self.lib.nmfmu_kernel_family(r_pad, precision, beta)

    @staticmethod
    def stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def alloc(self, nbytes: int, device) -> torch.Tensor:
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)

    # -- device work
    def pack_x(self, V, transpose, precision, block_rows, m_pad, k_pad, flags, out=None):
        xp = out if out is not None else self.alloc(self.lib.nmfmu_xp_bytes(m_pad, k_pad, precision), V.device)
        _capi.check(self.lib.nmfmu_pack_x(V.data_ptr(), V.stride(0), V.shape[0], V.shape[1], int(transpose), precision,
                                          block_rows, xp.data_ptr(), m_pad, k_pad, _ptr(flags), self.stream()),
                    'nmfmu_pack_x')
        return xp

    def xp_rows(self, xp, r0, n, k_pad, precision):
        """Fragment-order X of owner rows [r0, r0 + n) (r0, n multiples of the tile height): row blocks are outermost."""
        lo = self.lib.nmfmu_xp_bytes(r0, k_pad, precision)
        return xp[lo:lo + self.lib.nmfmu_xp_bytes(n, k_pad, precision)]

    def pack_factor(self, fac: 'FactorBuf', rank, r_pad, precision):
        _capi.check(self.lib.nmfmu_pack_factor(C.byref(fac.struct), rank, r_pad, precision, self.stream()),
                    'nmfmu_pack_factor')
        fac.nparts = self.lib.nmfmu_pack_nparts(fac.rows_pad)

    def mu_partial(self, st: 'StepBuf'):
        _capi.check(self.lib.nmfmu_mu_partial(C.byref(st.struct), self.stream()), 'nmfmu_mu_partial')

    def mu_step(self, st: 'StepBuf', kl_den, phase=0):
        _capi.check(self.lib.nmfmu_mu_step(C.byref(st.struct), _ptr(kl_den), phase, self.stream()), 'nmfmu_mu_step')
        if phase != 1:
            st.owner.nparts = self.lib.nmfmu_colsum_nparts(C.byref(st.struct))

    # -- beta == 2 without the reconstruction (nmfmu_gram_panel / nmfmu_xb_step)
    def xb_supported(self, r_pad: int, precision: int, beta: float) -> bool:
        return bool(self.lib.nmfmu_xb_supported(r_pad, precision, beta))

    def gram_alloc(self, r_pad: int, device):
        """(workspace, fp32 matrix, hi image, lo image, per-row scales) of one Gram matrix."""
        return (self.alloc(self.lib.nmfmu_gram_ws_bytes(r_pad), device),
                torch.zeros(r_pad * r_pad, dtype=torch.float32, device=device),
                self.alloc(r_pad * r_pad * 2, device), self.alloc(r_pad * r_pad * 2, device),
                torch.zeros(r_pad, dtype=torch.float32, device=device))

    def gram_panel(self, fac: 'FactorBuf', r_pad, precision, gm):
        ws, gram, hi, lo, scale = gm
        _capi.check(self.lib.nmfmu_gram_panel(C.byref(fac.struct), r_pad, precision, ws.data_ptr(), gram.data_ptr(),
                                              hi.data_ptr(), lo.data_ptr(), scale.data_ptr(), self.stream()),
                    'nmfmu_gram_panel')

    def xb_step(self, st: 'StepBuf', gm, phase=0):
        _, gram, hi, lo, scale = gm
        _capi.check(self.lib.nmfmu_xb_step(C.byref(st.struct), hi.data_ptr(), lo.data_ptr(), scale.data_ptr(), phase,
                                           self.stream()), 'nmfmu_xb_step')
        if phase != 1:
            st.owner.nparts = self.lib.nmfmu_colsum_nparts(C.byref(st.struct))

    # -- the library's own RCCL communicator (nmfmu_comm_*), bootstrapped over a torch.distributed group
    def comm_init(self, group, device):
        import torch.distributed as dist
        if not self.lib.nmfmu_comm_available():
            raise _capi.NmfmuError('librccl could not be loaded by libnmfmu (nmfmu_comm_available() == 0)')
        uid = torch.zeros(128, dtype=torch.uint8)
        if dist.get_rank(group) == 0:
            _capi.check(self.lib.nmfmu_comm_unique_id(uid.data_ptr()), 'nmfmu_comm_unique_id')
        uid_d = uid.to(device)
        dist.broadcast(uid_d, src=dist.get_global_rank(group, 0) if hasattr(dist, 'get_global_rank') else 0, group=group)
        uid = uid_d.cpu()
        comm = C.c_void_p()
        _capi.check(self.lib.nmfmu_comm_init_rank(C.byref(comm), dist.get_world_size(group), uid.data_ptr(),
                                                  dist.get_rank(group)), 'nmfmu_comm_init_rank')
        return _Comm(self.lib, comm)

    def mu_step_allreduce(self, st: 'StepBuf', comm: '_Comm', xbuf):
        _capi.check(self.lib.nmfmu_mu_step_allreduce(C.byref(st.struct), comm.h, xbuf.data_ptr(), self.stream()),
                    'nmfmu_mu_step_allreduce')
        st.owner.nparts = self.lib.nmfmu_pack_nparts(st.owner.rows_pad)

    def colsum_finalize(self, fac: 'FactorBuf', r_pad):
        _capi.check(self.lib.nmfmu_colsum_finalize(C.byref(fac.struct), fac.nparts, r_pad, self.stream()),
                    'nmfmu_colsum_finalize')

    def slab_reduce(self, st, num_out, den_out):
        _capi.check(self.lib.nmfmu_slab_reduce(C.byref(st.struct), _ptr(num_out), _ptr(den_out), self.stream()),
                    'nmfmu_slab_reduce')

    def mu_apply(self, st, num, den, nslab, kl_den):
        _capi.check(self.lib.nmfmu_mu_apply(C.byref(st.struct), _ptr(num), _ptr(den), nslab, _ptr(kl_den),
                                            self.stream()), 'nmfmu_mu_apply')
        st.owner.nparts = self.lib.nmfmu_pack_nparts(st.owner.rows_pad)

    def trainer_apply(self, st, kl_den, ortho, grad):
        _capi.check(self.lib.nmfmu_trainer_apply(C.byref(st.struct), None, None, 0, _ptr(kl_den), float(ortho),
                                                 _ptr(grad), self.stream()), 'nmfmu_trainer_apply')
        st.owner.nparts = self.lib.nmfmu_pack_nparts(st.owner.rows_pad)

    def loss(self, st, loss_part, out):
        _capi.check(self.lib.nmfmu_loss(C.byref(st.struct), _ptr(loss_part), _ptr(out), self.stream()), 'nmfmu_loss')

    # -- "riding loss" (include/nmfmu.h): fit()'s periodic KL loss carried by the next W half-step's kernel
    def riding_supported(self, st) -> bool:
        return self.lib.nmfmu_riding_loss_supported(C.byref(st.struct)) == 1

    def riding_alloc(self, st, device):
        n = self.lib.nmfmu_riding_loss_part_count(C.byref(st.struct))
        return torch.empty(n, dtype=torch.float32, device=device)

    def target_sums(self, V, part, out4):
        """out4 (device float64[4]) = {sum x ln(x + eps), sum x, max x, any(x != fp16(x))} of the fp32 target, one pass."""
        _capi.check(self.lib.nmfmu_target_sums(_ptr(V), V.stride(0), V.shape[0], V.shape[1], _ptr(part), _ptr(out4), self.stream()),
                    'nmfmu_target_sums')

    def mu_step_with_loss(self, st, kl_den, xlogs_part, target_sums, out2):
        _capi.check(self.lib.nmfmu_mu_step_with_loss(C.byref(st.struct), _ptr(kl_den), _ptr(xlogs_part), _ptr(target_sums),
                                                     _ptr(out2), self.stream()), 'nmfmu_mu_step_with_loss')

    def loss_checkpoint(self, st, loss_part, out2, fa, fa_snap, fb, fb_snap):
        """nmfmu_loss + the rest of a fit() loss checkpoint in two launches: out2 = (loss, fp16-range flag), factor snapshots."""
        _capi.check(self.lib.nmfmu_loss_checkpoint(C.byref(st.struct), _ptr(loss_part), _ptr(out2), _ptr(fa), _ptr(fa_snap),
                                                   fa.numel(), _ptr(fb), _ptr(fb_snap), fb.numel(), self.stream()),
                    'nmfmu_loss_checkpoint')


class _Comm:
    """Owner of an nmfmu_comm handle (destroyed with the engine)."""

    def __init__(self, lib, h):
        self.lib, self.h = lib, h

    def __del__(self):
        try:
            if self.h:
                self.lib.nmfmu_comm_destroy(self.h)
                self.h = None
        except Exception:
            pass


class KernelTimer:
    """hipEvent pairs around chosen launches, recorded on the launching stream (nmfmu_timer_* of the C ABI)."""

    def __init__(self, n_events: int):
        self.lib = _capi.load()
        self.h = C.c_void_p()
        _capi.check(self.lib.nmfmu_timer_create(n_events, C.byref(self.h)), 'nmfmu_timer_create')
        self.n = n_events
        self.next = 0
        self.tags = []

    def mark(self, tag: str):
        if self.next >= self.n:
            return
        _capi.check(self.lib.nmfmu_timer_record(self.h, self.next, torch.cuda.current_stream().cuda_stream),
                    'nmfmu_timer_record')
        self.tags.append(tag)
        self.next += 1

    def spans(self):
        """{tag: [ms, ...]} for consecutive (tag+'<', tag+'>') marks."""
        out = {}
        ms = C.c_float()
        for i in range(self.next - 1):
            a, b = self.tags[i], self.tags[i + 1]
            if a.endswith('<') and b == a[:-1] + '>':
                _capi.check(self.lib.nmfmu_timer_elapsed_ms(self.h, i, i + 1, C.byref(ms)), 'nmfmu_timer_elapsed_ms')
                out.setdefault(a[:-1], []).append(ms.value)
        return out

    def close(self):
        if self.h:
            self.lib.nmfmu_timer_destroy(self.h)
            self.h = C.c_void_p()


class FactorBuf:
    """Device state of one factor: the fp32 master (the nn.Parameter's storage) plus its bf16 images."""

    def __init__(self, data: torch.Tensor, r_pad: int, precision: int, backend):
        assert data.dim() == 2 and data.dtype == torch.float32 and data.is_contiguous()
        self.f = data
        self.rows, self.rank = data.shape
        self.rows_pad = backend.pad_rows(self.rows)
        dev = data.device
        nimg = self.rows_pad * r_pad * 2
        x3 = precision == _capi.PREC_BF16X3
        self.p1_hi = backend.alloc(nimg, dev)
        self.p2_hi = backend.alloc(nimg, dev)
        self.p1_lo = backend.alloc(nimg, dev) if x3 else None
        self.p2_lo = backend.alloc(nimg, dev) if x3 else None
        self.colsum = torch.zeros(r_pad, dtype=torch.float32, device=dev)
        self.colsum_part = torch.zeros((self.rows_pad // 16) * r_pad, dtype=torch.float32, device=dev)
        self.nparts = 0               # valid partial column sums in colsum_part (set by whoever wrote them last)
        self.struct = _capi.Factor(_ptr(self.f), _ptr(self.p1_hi), _ptr(self.p1_lo), _ptr(self.p2_hi),
                                   _ptr(self.p2_lo), _ptr(self.colsum), _ptr(self.colsum_part), self.rows,
                                   self.rows_pad)


class StepBuf:
    """One half-step: X in fragment order, owner/panel factors, partial-sum slabs."""

    def __init__(self, xp, owner: FactorBuf, panel: FactorBuf, rank, r_pad, nsplit, precision, stage, block_rows, beta,
                 gamma, l1, l2, need_den, status=None):
        self.xp, self.owner, self.panel = xp, owner, panel
        self.nsplit, self.r_pad, self.block_rows = nsplit, r_pad, block_rows
        dev = owner.f.device
        self.plane = owner.rows_pad * r_pad
        self.slab_num = torch.empty(nsplit * self.plane, dtype=torch.float32, device=dev)
        # (need_den == 'one': the path without reconstruction leaves ONE denominator slab, whatever the numerator's split)
        nden = 0 if not need_den else (1 if need_den == 'one' else nsplit)
        self.slab_den = torch.empty(nden * self.plane, dtype=torch.float32, device=dev) if nden else None
        self.struct = _capi.Step(_ptr(xp), owner.struct, panel.struct, _ptr(self.slab_num), _ptr(self.slab_den), rank,
                                 r_pad, nsplit, precision, stage, block_rows, beta, gamma, l1, l2, _ptr(status))


class _FactorRows:
    """Rows [r0, r0 + n) of a FactorBuf (n, r0 multiples of 256): same storage, offset pointers."""

    def __init__(self, fac: FactorBuf, r0: int, n: int, r_pad: int):
        self.f = fac.f[r0:r0 + n]
        self.rows, self.rank, self.rows_pad = self.f.shape[0], fac.rank, n
        img = lambda t, per_row: None if t is None else t[r0 * per_row:(r0 + n) * per_row]
        self.p1_hi, self.p1_lo = img(fac.p1_hi, r_pad * 2), img(fac.p1_lo, r_pad * 2)     # row-major image
        self.p2_hi, self.p2_lo = img(fac.p2_hi, r_pad * 2), img(fac.p2_lo, r_pad * 2)     # 64-row tiles: 64 | r0
        self.colsum, self.colsum_part = fac.colsum, fac.colsum_part
        self.struct = _capi.Factor(_ptr(self.f), _ptr(self.p1_hi), _ptr(self.p1_lo), _ptr(self.p2_hi), _ptr(self.p2_lo),
                                   _ptr(self.colsum), _ptr(self.colsum_part), self.rows, n)


class StepRows:
    """The partial-sum part of a half-step restricted to owner rows [r0, r0 + n): X, owner images and slabs of that row
    range, the whole panel.  Lets the sharded H half-step run as two launches so that the first half's all-reduce
    travels while the second half computes.  (The apply stays whole: it owns the column-sum finalize.)
    ``slab_num`` / ``slab_den``: storage for nsplit * n * r_pad floats each, carved by the caller out of one buffer that
    the whole half-step shares (the row halves and the single-launch form are never live together)."""

    def __init__(self, st: StepBuf, r0: int, n: int, backend, k_pad: int, nsplit: int, slab_num, slab_den):
        s0 = st.struct
        self.owner, self.panel = _FactorRows(st.owner, r0, n, st.r_pad), st.panel
        self.xp = backend.xp_rows(st.xp, r0, n, k_pad, s0.precision)
        self.nsplit = nsplit
        self.r_pad, self.block_rows = st.r_pad, st.block_rows
        self.r0, self.plane = r0, n * st.r_pad
        assert slab_num.numel() == nsplit * self.plane and (slab_den is None or slab_den.numel() == slab_num.numel())
        self.slab_num, self.slab_den = slab_num, slab_den
        self.struct = _capi.Step(_ptr(self.xp), self.owner.struct, st.panel.struct, _ptr(self.slab_num), _ptr(self.slab_den),
                                 s0.rank, s0.r_pad, self.nsplit, s0.precision, s0.stage, s0.block_rows, s0.beta, s0.gamma,
                                 s0.l1, s0.l2, s0.status)

    @staticmethod
    def nsplit_for(st: StepBuf, n: int, backend, k_pad: int, dev) -> int:
        """Its own contraction split: half the row blocks want twice the workgroups per block to fill the chip."""
        s0 = st.struct
        try:
            ns = backend.choose_nsplit(n, k_pad, st.block_rows, dev, s0.r_pad, s0.precision, s0.beta)
        except TypeError:           # (the oracle-backed stand-in of the CPU tests: the plain rule)
            ns = backend.choose_nsplit(n, k_pad, st.block_rows, dev)
        return max(st.nsplit, ns)


# Factory of the compute backend.  It is HipBackend in the product; the CPU test-suite swaps in an oracle-backed
# stand-in (tests/cpu_backend.py) to exercise the host logic (sharding, all-reduce packing, fit loop) without a GPU.
DEFAULT_BACKEND_FACTORY = HipBackend


def mu_gamma(beta: float) -> float:
    """MU exponent of nmf.py:341-346."""
    if beta < 1:
        return 1.0 / (2.0 - beta)
    if beta > 2:
        return 1.0 / (beta - 1.0)
    return 1.0


# One-entry memo of nmfmu_target_sums: the admission test of 'auto' and the riding loss of the engine built right after it ask
# about the same target.  Keyed by the tensor OBJECT (a weak reference: a new tensor that the caching allocator places at the
# same address is another object) and its version counter (an in-place edit invalidates it).
TARGET_STATS = {'ref': None, 'version': None, 'out4': None}


def target_stats(V, be):
    """Device float64[4] = {sum x ln(x + eps), sum x, max x, any(x != fp16(x))} of the fp32 target V (one pass)."""
    import weakref
    ref = TARGET_STATS['ref']
    if ref is None or ref() is not V or TARGET_STATS['version'] != V._version:
        part = torch.empty(4 * be.lib.nmfmu_target_sums_nparts(), dtype=torch.float64, device=V.device)
        out4 = torch.zeros(4, dtype=torch.float64, device=V.device)
        be.target_sums(V, part, out4)
        TARGET_STATS['ref'], TARGET_STATS['version'], TARGET_STATS['out4'] = weakref.ref(V), V._version, out4
    return TARGET_STATS['out4']


class AsyncLossMixin:
    """Loss checkpoints of ``fit`` without a host sync (reference loop: nmf.py:393-407).

    ``checkpoint_begin`` enqueues the loss kernel, an asynchronous copy of its result (and of the fp16-range flag) into
    pinned host memory, a device-side snapshot of the two factors, and an event -- then returns; the host keeps launching
    iterations.  ``checkpoint_result`` is asked one checkpoint later (the event completed ten iterations ago: no bubble);
    if the stop rule fired for the checkpoint, ``rollback`` restores the snapshot, so the caller gets exactly the factors
    the synchronous loop would have returned.  An engine provides ``_loss_device()`` (enqueue; float64[1] device tensor),
    ``_range_flag_device()`` (float64[1] device tensor or None), ``_ckpt_tensors()`` and ``refresh_images()``."""

    _ck = None

    def checkpoint_begin(self):
        tens = self._ckpt_tensors()
        if self._ck is None:
            on_gpu = tens[0].is_cuda       # (the CPU test-suite drives this logic through the stand-in backend)
            host = torch.zeros(2, dtype=torch.float64)
            self._ck = {'snap': [torch.empty_like(t) for t in tens],
                        'host': host.pin_memory() if on_gpu else host,
                        'stage': torch.zeros(2, dtype=torch.float64, device=tens[0].device),
                        'ev': torch.cuda.Event() if on_gpu else None}
        ck = self._ck
        if self._checkpoint_riding(ck, tens):      # the loss rides in the next W half-step (round 6): nothing else to launch now
            return
        if self._checkpoint_fused(ck['stage'], tens, ck['snap']):     # loss + flag + snapshots in two launches (round 6)
            ck['host'].copy_(ck['stage'], non_blocking=True)
        else:
            ck['stage'][0:1].copy_(self._loss_device())
            flag = self._range_flag_device()
            if flag is not None:
                ck['stage'][1:2].copy_(flag)
            ck['host'].copy_(ck['stage'], non_blocking=True)
            for s_, t_ in zip(ck['snap'], tens):
                s_.copy_(t_)
        if ck['ev'] is not None:
            ck['ev'].record()

    def _checkpoint_fused(self, stage, tens, snaps) -> bool:
        """Engines whose backend has a one-call checkpoint override this; False = take the generic sequence above."""
        return False

    def _checkpoint_riding(self, ck, tens) -> bool:
        """Engines that can let the NEXT half-step carry the loss override this: True = snapshots taken, the loss value, the
        host copy and the event follow with that half-step."""
        return False

    def checkpoint_result(self):
        """(divergence, left_f16_range) of the last ``checkpoint_begin``."""
        assert not getattr(self, '_riding_pending', False), 'the half-step that carries the loss has not run'
        if self._ck['ev'] is not None:
            self._ck['ev'].synchronize()
        return float(self._ck['host'][0]), bool(self._ck['host'][1] != 0)

    def rollback(self):
        for s_, t_ in zip(self._ck['snap'], self._ckpt_tensors()):
            t_.copy_(s_)
        self.refresh_images()


class DenseMU(AsyncLossMixin):
    """Engine for ``NMF.fit``: V (N, C) ~ H (N, R) @ W (C, R)^T on the current device.

    ``W`` / ``H`` are the parameters' ``.data`` tensors and are updated in place
    (nmf.py:92).  With ``group`` set, ``V`` / ``W`` are this rank's column shard.
    """

    # 'auto' may pick the single-plane fp16 mode only where it meets the 1e-4 parity bar:
    #  * both contraction lengths long enough for the per-step operand rounding errors to average down -- emulated and
    #    measured factor errors after 200 iterations: 9e-5 at 2048 x 2048 rank 64, 7e-5 at 4096 x 4096 rank 128;
    #  * V EXACTLY representable in fp16 (integer counts below 2048, 8-bit images, bf16 / fp16-sourced data): the mode
    #    stores the target in fp16, and rounding V perturbs the problem itself -- that error does not average down, it
    #    grows with the iteration count towards the perturbed fixed point (3e-4 after 200 iterations at any size);
    #  * data inside fp16's range with room for the ratios.
    F16_MIN_DIM = 4096
    _F16_MODES = (_capi.PREC_F16, _capi.PREC_F16X, _capi.PREC_F16R)     # fp16 operand images (range watch applies)
    F16_MAX_ABS = 3.0e4
    F16_MIN_MEAN = 2.0 ** -10

    @classmethod
    def f16_stats(cls, V, W, H, be=None):
        """(in_range, exact): the data sit inside fp16's range with room for the ratios / V is exactly representable in
        fp16.  With a device backend: ONE pass over V (nmfmu_target_sums: max, fp16-exactness and the two sums the riding
        loss needs later -- kept in TARGET_STATS for the engine built next), one over W and H, one host sync.  Otherwise
        (the CPU test backend): two passes over V in row chunks, so that the temporaries stay small next to V."""
        if (be is not None and hasattr(be, 'target_sums') and V.is_cuda and V.dtype == torch.float32 and V.dim() == 2
                and V.stride(1) == 1):
            out4 = target_stats(V, be)
            st = torch.stack([out4[3].float(), out4[2].float(), (out4[1] / V.numel()).float(), W.max(), H.max(), W.mean(),
                              H.mean()]).tolist()
            inexact, vmax, vmean, wmax, hmax, wmean, hmean = st
            in_range = max(vmax, wmax, hmax) <= cls.F16_MAX_ABS and min(vmean, wmean, hmean) >= cls.F16_MIN_MEAN
            return in_range, not inexact
        bad = torch.zeros((), dtype=torch.bool, device=V.device)
        vmax = torch.zeros((), dtype=torch.float32, device=V.device)
        vsum = torch.zeros((), dtype=torch.float64, device=V.device)
        step = max(1, (64 << 20) // max(1, V.shape[1]))
        for r0 in range(0, V.shape[0], step):
            v = V[r0:r0 + step]
            bad |= (v.half().float() != v).any()
            vmax = torch.maximum(vmax, v.max())
            vsum += v.sum(dtype=torch.float64)
        stats = torch.stack([bad.float(), vmax, (vsum / V.numel()).float(), W.max(), H.max(), W.mean(), H.mean()]).tolist()
        inexact, vmax, vmean, wmax, hmax, wmean, hmean = stats
        in_range = max(vmax, wmax, hmax) <= cls.F16_MAX_ABS and min(vmean, wmean, hmean) >= cls.F16_MIN_MEAN
        return in_range, not inexact

    @classmethod
    def f16_in_range(cls, V, W, H) -> bool:
        """Admission test of the 'f16' mode (fp16 operands AND target): in range and V exact in fp16."""
        in_range, exact = cls.f16_stats(V, W, H)
        return in_range and exact

    @classmethod
    def auto_single_plane(cls, V, W, H, r_pad, be, group=None) -> Optional[str]:
        """What precision='auto' may take at 1x MFMA work, or None: 'f16' (fp16 operands and target) when V is exact in
        fp16, 'f16x' (fp16 operands, fp32 target) when it is not -- both only where the contraction lengths average the
        operand rounding down (both dimensions >= F16_MIN_DIM) and the data sit inside fp16's range.  On a sharded fit
        EVERY rank enters the same all-reduce, whatever its own shard looks like (shards of 4096, 4096, 4095
        columns must not disagree about entering a collective), and all take the weakest rank's answer."""
        if (os.environ.get('TORCHNMF_AMD_AUTO_F16', '1') == '0' or not hasattr(_capi, 'PREC_F16')
                or not be.supported(r_pad, _capi.PREC_F16)):
            return None                                   # rank-invariant: same library, same environment on every rank
        dims_ok = min(V.shape) >= cls.F16_MIN_DIM
        if group is None and not dims_ok:
            return None
        in_range, exact = cls.f16_stats(V, W, H, be) if dims_ok else (False, False)
        level = 2 if (in_range and exact) else (1 if in_range else 0)      # 2: f16, 1: f16x, 0: neither
        if level == 1 and not (hasattr(_capi, 'PREC_F16X') and be.supported(r_pad, _capi.PREC_F16X)
                               and os.environ.get('TORCHNMF_AMD_AUTO_F16X', '1') != '0'):
            level = 0
        if group is not None:
            import torch.distributed as dist
            flag = torch.tensor([level], dtype=torch.int32, device=V.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            level = int(flag.item())
            if level == 1 and not (hasattr(_capi, 'PREC_F16X') and be.supported(r_pad, _capi.PREC_F16X)):
                level = 0
        return {2: 'f16', 1: 'f16x', 0: None}[level]

    @classmethod
    def auto_mode(cls, V, W, H, r_pad, be, beta, group=None) -> Optional[str]:
        """``auto_single_plane`` with the 3-byte target (round 6): a target fp16 does not hold exactly runs as 'f16r' -- the
        fp32 rounded to its top 24 bits (16 significant bits, fp32's range), 25 % fewer bytes of V per iteration than 'f16x',
        nothing rounded that the parity bar can see -- except for beta == 2, where the target is an MFMA operand itself and stays fp32
        ('f16x').  Rank-invariant like its input (same library and environment on every rank)."""
        mode = cls.auto_single_plane(V, W, H, r_pad, be, group)
        if (mode == 'f16x' and float(beta) != 2.0 and hasattr(_capi, 'PREC_F16R') and be.supported(r_pad, _capi.PREC_F16R)
                and os.environ.get('TORCHNMF_AMD_AUTO_F16R', '1') != '0'):
            mode = 'f16r'
        return mode

    def __init__(self, V, W, H, beta, l1=0.0, l2=0.0, precision='auto', stage=None, group=None, backend=None,
                 update_W=True, update_H=True, block_rows=None, allow_f16=False, ar_overlap=None, allow_gram=False,
                 ar_direct=None):
        self.be = backend if backend is not None else DEFAULT_BACKEND_FACTORY()
        self.group = group
        self.beta = float(beta)
        self.kl = self.beta == 1.0
        N, Cc = V.shape
        R = W.shape[1]
        assert W.shape == (Cc, R) and H.shape == (N, R)
        if V.stride(1) != 1 or V.dtype != torch.float32:     # nmfmu_pack_x reads rows with a row pitch only
            V = V.float().contiguous()
        self.rank = R
        self.r_pad = self.be.pad_rank(R)
        if precision in (None, 'auto'):
            # 'auto' = the fastest mode that meets the reference's 1e-4 bar -- never the plain bf16 mode:
            # fp16 operands (1x MFMA work) where the contraction lengths average the rounding errors down and the data
            # fit fp16's range (sharded: every rank must decide alike, so the range test is all-reduced), else split
            # bf16 (3x MFMA work, fp32-grade, padded rank <= 128), else an error.
            precision = self.auto_mode(V, W, H, self.r_pad, self.be, self.beta, group) if allow_f16 else None
            if precision is None:
                if not self.be.supported(self.r_pad, _capi.PREC_BF16X3):
                    raise NotImplementedError(
                        f"precision='auto' found no mode for rank {R} that meets the 1e-4 parity bar on the fused kernels "
                        f"(fp16 operands -- 'f16', or 'f16x' with an fp32 target -- need both dimensions >= {self.F16_MIN_DIM} "
                        f"and data within fp16's range; split bf16 stops at rank 128); pass precision='bf16' (factors "
                        f"~1e-3), 'f16' or 'f16x' explicitly")
                precision = 'bf16x3'
        if precision not in _capi.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_capi.PRECISIONS)} or 'auto', got {precision!r}")
        self.precision_name = precision
        self.precision = _capi.PRECISIONS[precision]
        if not self.be.supported(self.r_pad, self.precision):
            raise NotImplementedError(f'precision {precision!r} is not available for rank {R} (padded {self.r_pad})')
        if stage is None:
            # (round 6) beta == 1 at padded rank 256 with fp16 operands runs the software-pipelined kernel in BOTH half-steps and
            # for the loss: ONE image per factor, GEMM2's operand gathered from it -- nothing reads the transposed images, so
            # nobody has to keep them up to date (NMFMU_STAGE_DMA_NOP2: 17 % less epilogue traffic)
            nop2 = (hasattr(self.be, 'kernel_family') and hasattr(_capi, 'STAGE_DMA_NOP2') and float(beta) == 1.0
                    and self.r_pad == 256 and self.be.kernel_family(self.r_pad, self.precision, float(beta)) == _capi.KERNEL_SP
                    and os.environ.get('TORCHNMF_AMD_NO_P2', '1') != '0')
            stage = _capi.STAGE_DMA_NOP2 if nop2 else _capi.STAGE_DMA
        # beta == 2 without the reconstruction (nmf.py:61-63 has no eps inside its grad_outputs): numerator = one streaming
        # GEMM over X, denominator through the panel's rank x rank Gram matrix -- a third of the MFMA work, HBM-bound.
        # fit()'s engines only (allow_gram): BetaMu reads numerator AND denominator slabs (p.grad = pos - neg).
        self.gram_path = (allow_gram and self.beta == 2.0 and group is None and block_rows in (None, 128)
                          and hasattr(self.be, 'xb_supported') and self.be.xb_supported(self.r_pad, self.precision, self.beta)
                          and os.environ.get('TORCHNMF_AMD_BETA2_GRAM', '1') != '0')
        gamma = mu_gamma(self.beta)
        dev = V.device

        self.fW = FactorBuf(W, self.r_pad, self.precision, self.be)
        self.fH = FactorBuf(H, self.r_pad, self.precision, self.be)
        # validation flags of nmf.py:329-336: [any(!(v >= 0)), min bit pattern]
        self.flags = torch.tensor([0, 0x7f800000], dtype=torch.int32, device=dev)
        # fp16 mode: bit 0 is set by any update that had to clamp a factor value at 65504 for its fp16 image
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        n_pad, c_pad = self.fH.rows_pad, self.fW.rows_pad
        # tile height per half-step (the two packed copies of V are independent): the library's choice, or forced
        def tile_rows(m_pad, k_pad):
            if block_rows is not None:
                return block_rows
            if hasattr(self.be, 'step_block_rows'):
                return self.be.step_block_rows(m_pad, k_pad, self.r_pad, self.precision, self.beta, dev)
            return self.be.block_rows(self.r_pad, self.precision, self.beta)
        br = tile_rows(n_pad, c_pad)
        self.block_rows = br
        # H half-step and loss: owner axis N, contraction over C
        xp_h = self.be.pack_x(V, False, self.precision, br, n_pad, c_pad, self.flags)
        ns_h = self._nsplit(n_pad, c_pad, br, dev)
        need_den = False if self.kl else ('one' if self.gram_path else True)   # no slab | one slab | nsplit slabs
        self.step_h = StepBuf(xp_h, self.fH, self.fW, R, self.r_pad, ns_h, self.precision, stage, br, self.beta, gamma,
                              l1, l2, need_den=need_den, status=self.status)
        self.step_w = None
        if update_W:
            brw = tile_rows(c_pad, n_pad)
            xp_w = self.be.pack_x(V, True, self.precision, brw, c_pad, n_pad, None)
            ns_w = self._nsplit(c_pad, n_pad, brw, dev)
            self.step_w = StepBuf(xp_w, self.fW, self.fH, R, self.r_pad, ns_w, self.precision, stage, brw, self.beta,
                                  gamma, l1, l2, need_den=need_den, status=self.status)
        self.gm = self.be.gram_alloc(self.r_pad, dev) if self.gram_path else None
        self.timer: Optional[KernelTimer] = None   # bench.py: times the fused launches live
        self.refresh_images()
        self.loss_part = torch.empty(max((n_pad // br) * ns_h, 1), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=dev)
        # "riding loss" (round 6): unsharded beta == 1 fits on the ping-pong kernel let the W half-step that follows a loss
        # checkpoint carry the checkpoint's loss (its reconstruction IS the one nmf.py:400-401 evaluates): no pass over V
        self._riding = None
        self._riding_pending = False
        if (group is None and self.kl and self.step_w is not None and update_H and hasattr(self.be, 'riding_supported')
                and os.environ.get('TORCHNMF_AMD_RIDING_LOSS', '1') != '0' and V.dtype == torch.float32 and V.stride(1) == 1
                and self.be.riding_supported(self.step_w)):
            self._riding = {'V': V, 'bufs': None}       # (target sums: computed at the first checkpoint, once)
        if group is not None:
            # [numerator | denominator (N x R for beta != 1, else R column sums)] -> ONE all-reduce per iteration
            tail = self.r_pad if self.kl else self.step_h.plane
            self.xbuf = torch.empty(self.step_h.plane + tail, dtype=torch.float32, device=dev)
            # default (fit(..., allreduce='single') / TORCHNMF_AMD_AR_OVERLAP=0): one launch, ONE all-reduce of the packed
            # buffer per iteration -- the form north_star names.  'overlap' (TORCHNMF_AMD_AR_OVERLAP=1): the H half-step as
            # two row halves -- the first half's numerators are on the wire while the second half's kernel runs.  The split point
            # defaults to the middle row block; TORCHNMF_AMD_AR_SPLIT=<fraction of the rows in the first part> moves it
            # (to be swept on a multi-GPU node: the first part's all-reduce should just fit behind the second part's kernel)
            # 'direct' (fit(..., allreduce='direct') / TORCHNMF_AMD_COMM=c): the whole sharded H half-step is ONE C call --
            # partial sums, slab reduction, a single RCCL all-reduce on the compute stream (the library's own communicator,
            # bootstrapped over the torch group) and the apply; no return to Python between kernel and collective
            explicit_direct = bool(ar_direct)
            if ar_direct is None:
                ar_direct = os.environ.get('TORCHNMF_AMD_COMM', 'torch') == 'c'
            self._comm = None
            if ar_direct and hasattr(self.be, 'comm_init'):
                self._comm = self.be.comm_init(group, dev)
                ar_overlap = False
            elif explicit_direct:
                # (this used to fall back to the torch.distributed route without a word)
                raise _capi.NmfmuError("allreduce='direct' needs the library's own RCCL communicator (nmfmu_comm_*), which "
                                       f"the {getattr(self.be, 'name', type(self.be).__name__)} backend does not provide")
            st = self.step_h
            nblk = st.owner.rows_pad // 256
            frac = float(os.environ.get('TORCHNMF_AMD_AR_SPLIT', '0.5'))
            r0 = min(max(int(frac * nblk + 1e-9), 1), max(nblk - 1, 1)) * 256
            self._h_rows = None
            if ar_overlap is None:
                # default: ONE all-reduce of the packed buffer per iteration -- what north_star specifies; the two-collective
                # overlap stays opt-in until an N > 1 run shows it wins
                ar_overlap = os.environ.get('TORCHNMF_AMD_AR_OVERLAP', '0') != '0'
            if (ar_overlap and nblk >= 2 and st.owner.rows > r0
                    and hasattr(self.be, 'xp_rows')):
                k_pad = st.panel.rows_pad
                parts = [(0, r0), (r0, st.owner.rows_pad - r0)]
                ns = [StepRows.nsplit_for(st, n, self.be, k_pad, dev) for _, n in parts]
                need = sum(k * n * self.r_pad for k, (_, n) in zip(ns, parts))
                # one slab buffer for both forms of the half-step: re-point the whole-step slabs into it as well
                big_n = torch.empty(max(need, st.nsplit * st.plane), dtype=torch.float32, device=dev)
                big_d = None if st.slab_den is None else torch.empty_like(big_n)
                st.slab_num, st.slab_den = big_n[:st.nsplit * st.plane], (None if big_d is None else big_d[:st.nsplit * st.plane])
                st.struct.slab_num, st.struct.slab_den = _ptr(st.slab_num), _ptr(st.slab_den)
                self._h_rows, off = [], 0
                for k, (a0, n) in zip(ns, parts):
                    cnt = k * n * self.r_pad
                    self._h_rows.append(StepRows(st, a0, n, self.be, k_pad, k, big_n[off:off + cnt],
                                                 None if big_d is None else big_d[off:off + cnt]))
                    off += cnt

    # ------------------------------------------------------------------
    def repack_target(self, V):
        """Pack a new target of the same shape into the existing buffers (both orientations) and re-run the validation
        of nmf.py:329-336; the factors, images and slabs stay.  Used by trainer.BetaMu for targets it has to convert on
        every step (non-fp32 / strided tensors whose source may have changed in place)."""
        st = self.step_h
        assert tuple(V.shape) == (st.owner.rows, st.panel.rows), 'repack_target: shape differs from the bound target'
        if V.stride(1) != 1 or V.dtype != torch.float32:
            V = V.float().contiguous()
        self.flags.copy_(torch.tensor([0, 0x7f800000], dtype=torch.int32))
        self.be.pack_x(V, False, self.precision, st.block_rows, st.owner.rows_pad, st.panel.rows_pad, self.flags, out=st.xp)
        if self.step_w is not None:
            sw = self.step_w
            self.be.pack_x(V, True, self.precision, sw.block_rows, sw.owner.rows_pad, sw.panel.rows_pad, None, out=sw.xp)

    def _nsplit(self, m_pad, k_pad, block_rows, dev):
        """Contraction split of a half-step, sized for the kernel it runs on where the backend can tell (the oracle-backed
        stand-in of the CPU tests has the plain rule only)."""
        try:
            return self.be.choose_nsplit(m_pad, k_pad, block_rows, dev, self.r_pad, self.precision, self.beta)
        except TypeError:
            return self.be.choose_nsplit(m_pad, k_pad, block_rows, dev)

    def refresh_images(self):
        """Re-derive bf16 images / column sums from the fp32 masters (after external edits of W / H)."""
        self.be.pack_factor(self.fW, self.rank, self.r_pad, self.precision)
        self.be.pack_factor(self.fH, self.rank, self.r_pad, self.precision)

    def target_flags(self):
        """(has_negative_or_nan, has_zero) over the whole (possibly sharded) target.  One host sync."""
        fl = self.flags.clone()
        if self.group is not None:
            import torch.distributed as dist
            bad = fl[0:1].clone()
            mn = fl[1:2].clone()
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=self.group)
            fl = torch.cat([bad, mn])
        bad, mn = (int(x) for x in fl.tolist())
        return bool(bad), mn == 0

    def _local_step(self, st, kl_den, tag):
        """A complete single-device half-step (nmfmu_mu_step); with a timer attached the fused kernel is bracketed."""
        if not hasattr(self.be, 'mu_step'):          # stand-in test backend
            self._partial(st, tag)
            self.be.mu_apply(st, None, None, 0, kl_den)
        elif self.timer is None:
            self.be.mu_step(st, kl_den, 0)
        else:
            self.timer.mark(tag + '<')
            self.be.mu_step(st, kl_den, 1)
            self.timer.mark(tag + '>')
            self.be.mu_step(st, kl_den, 2)

    def _partial(self, st, tag):
        if self.timer is None:
            self.be.mu_partial(st)
        else:
            self.timer.mark(tag + '<')
            self.be.mu_partial(st)
            self.timer.mark(tag + '>')

    def _gram_step(self, st, tag):
        """beta == 2 half-step without the reconstruction: Gram matrix of the panel, then X @ panel with the update in the
        kernel's epilogue (unsplit contraction) or in the apply kernel."""
        self.be.gram_panel(st.panel, self.r_pad, self.precision, self.gm)
        if self.timer is None:
            self.be.xb_step(st, self.gm, 0)
        else:
            self.timer.mark(tag + '<')
            self.be.xb_step(st, self.gm, 1)
            self.timer.mark(tag + '>')
            self.be.xb_step(st, self.gm, 2)

    def w_step(self):
        """nmf.py:367-378.  Local even when sharded: W rows belong to this rank's columns."""
        if self.gram_path:
            return self._gram_step(self.step_w, 'w')
        if self._riding_pending:      # this half-step carries the loss of the checkpoint just begun (AsyncLossMixin)
            self._riding_pending = False
            part, sums = self._riding['bufs']
            ck = self._ck
            self.be.mu_step_with_loss(self.step_w, self.fH.colsum, part, sums, ck['stage'])
            ck['host'].copy_(ck['stage'], non_blocking=True)
            if ck['ev'] is not None:
                ck['ev'].record()
            return
        self._local_step(self.step_w, self.fH.colsum if self.kl else None, 'w')

    def h_step(self):
        """nmf.py:380-391, with the freshly updated W."""
        st = self.step_h
        if self.gram_path:
            return self._gram_step(st, 'h')
        if self.group is None:
            self._local_step(st, self.fW.colsum if self.kl else None, 'h')
            return
        import torch.distributed as dist
        if getattr(self, '_comm', None) is not None:
            if self.timer is not None:
                self.timer.mark('h<')
            self.be.mu_step_allreduce(st, self._comm, self.xbuf)
            if self.timer is not None:
                self.timer.mark('h>')
            return
        num = self.xbuf[:st.plane]
        tail = self.xbuf[st.plane:]
        if self._h_rows is not None:
            # rows [0, r0): partial sums -> reduced slab -> all-reduce in flight; rows [r0, N): the same, its all-reduce
            # (with the denominators) is the exposed one
            v0, v1 = self._h_rows
            self._partial(v0, 'h0')
            self.be.slab_reduce(v0, num[:v0.plane], None if self.kl else tail[:v0.plane])
            w0 = dist.all_reduce(self.xbuf[:v0.plane], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._partial(v1, 'h1')
            self.be.slab_reduce(v1, num[v0.plane:], None if self.kl else tail[v0.plane:])
            if self.kl:
                tail.copy_(self.fW.colsum)
            if self.timer is not None:
                self.timer.mark('ar<')
            w1 = dist.all_reduce(self.xbuf[v0.plane:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w0.wait()
            w1.wait()
            if self.timer is not None:
                self.timer.mark('ar>')
            if self.kl:
                self.be.mu_apply(st, num, None, 1, tail)
            else:
                self.be.mu_apply(st, num, tail, 1, None)
            return
        self._partial(st, 'h')
        if self.kl:
            self.be.slab_reduce(st, num, None)
            tail.copy_(self.fW.colsum)
        else:
            self.be.slab_reduce(st, num, tail)
        if self.timer is not None:
            self.timer.mark('ar<')
        dist.all_reduce(self.xbuf, op=dist.ReduceOp.SUM, group=self.group)
        if self.timer is not None:
            self.timer.mark('ar>')
        if self.kl:
            self.be.mu_apply(st, num, None, 1, tail)
        else:
            self.be.mu_apply(st, num, tail, 1, None)

    def trainer_step(self, which: str, ortho: float = 0.0, grad: Optional[torch.Tensor] = None):
        """One parameter update of ``trainer.BetaMu.step`` (trainer.py:72-112): the two backward contractions are the
        fused kernel's numerator / denominator slabs; ``grad`` (same shape as the factor) receives ``p.grad``."""
        if self.group is not None:
            raise NotImplementedError('BetaMu on a column-sharded layer is not implemented')
        assert not self.gram_path, 'trainer_step needs the denominator slabs (construct the engine without allow_gram)'
        st = self.step_w if which == 'W' else self.step_h
        assert st is not None
        other = self.fH if which == 'W' else self.fW
        self._partial(st, which.lower())
        self.be.trainer_apply(st, other.colsum if self.kl else None, ortho, grad)

    def left_f16_range(self) -> bool:
        """fp16 mode: has any update so far clamped a factor value at 65504 for its image?  (One small device read; fit()
        asks at its loss checkpoints, where it synchronises anyway.)"""
        return self.precision in self._F16_MODES and bool(int(self.status.item()) & 1)

    # -- AsyncLossMixin (unsharded fits)
    def _loss_device(self):
        self.be.loss(self.step_h, self.loss_part, self.loss_out)
        return self.loss_out

    def _range_flag_device(self):
        if self.precision not in self._F16_MODES:
            return None
        return (self.status & 1).double()

    def _checkpoint_riding(self, ck, tens) -> bool:
        rd = self._riding
        if rd is None:
            return False
        if rd['bufs'] is None:
            rd['bufs'] = (self.be.riding_alloc(self.step_w, tens[0].device), target_stats(rd['V'], self.be).clone())
            rd['V'] = None                                  # (the engine does not keep the caller's target alive)
        # snapshots of (W, H) as they are now; the W half-step that follows does the rest
        for s_, t_ in zip(ck['snap'], tens):
            s_.copy_(t_)
        self._riding_pending = True
        return True

    def _checkpoint_fused(self, stage, tens, snaps) -> bool:
        if (not hasattr(self.be, 'loss_checkpoint') or len(tens) != 2 or os.environ.get('TORCHNMF_AMD_FUSED_CHECKPOINT', '1') == '0'
                or any(t.numel() % 4 or t.data_ptr() % 16 or not t.is_contiguous() for t in list(tens) + list(snaps))):
            return False
        # (status is only wired into the step structs of the fp16 modes' engines; other precisions never set bit 0)
        self.be.loss_checkpoint(self.step_h, self.loss_part, stage, tens[0], snaps[0], tens[1], snaps[1])
        return True

    def _ckpt_tensors(self):
        return [self.fW.f, self.fH.f]

    def divergence(self) -> float:
        """beta_div(H W^T, V) (nmf.py:360-361 / 400-401), summed over shards.  One host sync."""
        self.be.loss(self.step_h, self.loss_part, self.loss_out)
        if self.group is not None:
            import torch.distributed as dist
            dist.all_reduce(self.loss_out, op=dist.ReduceOp.SUM, group=self.group)
        return float(self.loss_out.item())
