"""Host-side orchestration of NMFD (1-D convolutive NMF) on one MI355X.

Reference: ``NMFD.reconstruct`` = ``F.conv1d(H, W.flip(2), padding=T-1)`` (nmf.py:776-779) driven by the same
``fit`` loop (nmf.py:297-409).  With ``Wm = W.view(C, R*T)`` and the Toeplitz unfold
``Hu[(b,l)][(r,t)] = H[b][r][l-t]`` the reconstruction is ``Wm @ Hu.T`` and both conv-backward passes are GEMMs,
so one MU iteration is four NT GEMMs (C ABI ``nmfmu_gemm``) with fused epilogues plus small unfold / fold /
apply kernels:

    W half-step   Gn[c][(b,l)]   = ratio( Wm Hu^T , V )              GEMM + EPI_RATIO
                  num[c][(r,t)]  = Gn  HuT^T                         GEMM + EPI_F32   -> nmfmu_conv_apply_w
    H half-step   GnT[(b,l)][c]  = ratio( Hu Wm^T , V^T )            GEMM + EPI_RATIO (transposed problem)
                  Y[(r,t)][(b,l)] = WmT GnT^T                        GEMM + EPI_F32   -> nmfmu_conv_fold_apply_h

NMFD is not sharded across GPUs ("replicas only").
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _capi
from .engine import AsyncLossMixin, _ptr, mu_gamma


def _pad128(n: int) -> int:
    return (n + 127) // 128 * 128


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Planes:
    """bf16 (hi[, lo]) planes of a zero-padded row-major matrix."""

    def __init__(self, rows_pad, cols_pad, x3, dev, zero=False):
        self.rows_pad, self.cols_pad = rows_pad, cols_pad
        alloc = torch.zeros if zero else torch.empty     # zero: a writer that does not cover the padding every time
        self.hi = alloc(rows_pad * cols_pad, dtype=torch.int16, device=dev)
        self.lo = alloc(rows_pad * cols_pad, dtype=torch.int16, device=dev) if x3 else None


class _Table:
    """Window table of H standing in for an (implicit) Toeplitz operand: same interface as _Planes."""

    def __init__(self, rows_pad, cols_pad, nbytes, x3, dev):
        self.rows_pad, self.cols_pad = rows_pad, cols_pad
        self.hi = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.lo = torch.empty(nbytes, dtype=torch.uint8, device=dev) if x3 else None


def tail_round_split(mt: int, nt: int, slots: int, k_tiles: int, want: str = '1'):
    """Tail-round split of an EPI_FOLD GEMM with mt x nt tiles on `slots` workgroup slots (two per CU) and k_tiles k-tiles:
    (tail_rows, k_split), (0, 1) = none.  Automatic ('1'): when the tiles make at least one full round plus at most a
    quarter round that is whole tile rows, those rows are contraction-split as far as the free slots, half the k-tiles
    and 8 allow.  'rows,split' forces a choice (tests).  Either way the split is normalised so that every part holds
    ceil(k_tiles / split) k-tiles and none is empty (the kernel's parts are [z * per, (z + 1) * per))."""
    rows, split = 0, 1
    rem = (mt * nt) % slots
    if ',' in want:
        rows, split = (int(v) for v in want.split(','))
        assert 0 < rows <= mt and 2 <= split <= k_tiles
    elif mt * nt > slots and 0 < rem <= slots // 4 and rem % nt == 0 and k_tiles >= 8:
        rows = rem // nt
        split = max(2, min(slots // rem, k_tiles // 2, 8))
    if rows:
        per = -(-k_tiles // split)
        split = -(-k_tiles // per)
        if split < 2:
            rows, split = 0, 1
    return rows, split


def w_contraction_split(tiles: int, k_tiles: int, slots: int) -> int:
    """Split of the W-numerator GEMM's contraction (round 4, the general form): few output tiles and a very long contraction
    (NMF2D: 8 tiles x 2 048 k-tiles) -- as many parts as fill the workgroup slots, at most 64, each at least eight k-tiles,
    dividing the k-tiles evenly (nmfmu_gemm wants k_tiles % k_split == 0)."""
    want = min(64, slots // max(tiles, 1), k_tiles // 8)
    return max([s for s in range(1, max(want, 1) + 1) if k_tiles % s == 0])


def h_tap_fold(rank: int, taps_last: int) -> int:
    """Taps of the last axis folded into the 32-wide N tile of the window-operand GEMM (nmfmu_gemm_desc.win_fold): the
    largest F in {4, 2} with rank * F <= 32 that divides the taps of the last axis, else 1."""
    return max([f for f in (4, 2) if rank * f <= 32 and taps_last % f == 0] + [1])


class ConvMU(AsyncLossMixin):
    """Engine for ``NMFD.fit``: V (B, C, L), W (C, R, T), H (B, R, L-T+1); W / H updated in place."""

    F16_MIN_DIM = 1024       # 'auto' -> 'f16' from this size on (every contraction at least this long)
    F16_MAX_ABS = 3.0e4      # ... and only when V, W, H fit fp16's range with headroom
    F16_MIN_MEAN = 2.0 ** -10

    def __init__(self, V, W, H, beta, l1=0.0, l2=0.0, precision='auto', update_W=True, update_H=True, own_loop=True):
        # own_loop=False: a caller that drives the GEMMs itself (plca._ConvPlcaEM): none of the paths that fuse this engine's
        # own update into a GEMM's neighbours (fold parts, fused sums, ragged channels), split bf16 unless told otherwise
        self.lib = _capi.load()
        self.staged = {}          # tag of a GEMM launch -> nmfmu_gemm_window_staged() of its descriptor (None: the query failed)
        # (resolved once, not per launch) '0' keeps the chunk-major implicit tiles of rounds 1-4 (A/B switch)
        self._stage_mode = 1 if os.environ.get('TORCHNMF_AMD_NMFD_WINSTAGE', '1') == '0' else 0
        if not torch.cuda.is_available():
            raise _capi.NmfmuError('torchnmf_amd needs a ROCm device (MI355X); there is no CPU fallback')
        # NMFD has one shift axis, NMF2D / NMF3D two / three (nmf.py:700-942); flattened they are the same problem
        nd = V.dim() - 2
        assert nd in (1, 2, 3) and W.dim() == V.dim() and H.dim() == V.dim()
        B, Cc = V.shape[:2]
        C_, R = W.shape[:2]
        Bh, Rh = H.shape[:2]
        self.ls, self.ts, self.lhs = tuple(V.shape[2:]), tuple(W.shape[2:]), tuple(H.shape[2:])
        assert (C_, Rh, Bh) == (Cc, R, B) and all(lh == l - t + 1 for lh, l, t in zip(self.lhs, self.ls, self.ts)), \
            'V, W, H shapes are inconsistent'
        L, T, Lh = (int(torch.tensor(x).prod()) for x in (self.ls, self.ts, self.lhs))   # flattened extents
        self.nd = nd
        self._lh_arr = (C.c_int32 * nd)(*self.lhs)
        self._t_arr = (C.c_int32 * nd)(*self.ts)
        for t_ in (W, H):
            assert t_.dtype == torch.float32 and t_.is_contiguous()
        # 'f16' (fp16 operand planes, window tables and ratio planes: 11 significant bits at the bf16 MFMA rate) exists for
        # the beta == 1 iteration on implicit operands, in two forms: one shift axis with >= 128 taps (the fold-parts /
        # fused-sums path), and -- round 4 -- the path whose H numerator is the window-operand GEMM (any number of shift
        # axes, fewer than 128 taps).  'auto' takes it where every contraction is long enough for the rounding errors to
        # average down and the data fit fp16's range; the fp32-grade split mode otherwise.
        aligned = (self.ts[-1] % 8 == 0 and self.ls[-1] % 8 == 0 and os.environ.get('TORCHNMF_AMD_NMFD_EXPLICIT', '0') != '1')
        f16_fold = (nd == 1 and T >= 128 and os.environ.get('TORCHNMF_AMD_NMFD_FOLD_PARTS', '1') != '0' and
                    os.environ.get('TORCHNMF_AMD_NMFD_FUSED_SUMS', '1') != '0')
        f16_rows = ((nd > 1 or 1 < T < 128) and os.environ.get('TORCHNMF_AMD_NMFD_H_ROWS', '1') != '0')
        # ... and only where one of the two H-numerator paths built for fp16 planes will actually be taken ('auto'
        # used to decide from f16_fold / f16_rows alone and could then fail the explicit-'f16' check further down -- ratio
        # planes of 2 GiB or more, or TORCHNMF_AMD_NMFD_H_ROWS=0 where the fold-parts path does not apply -- instead of
        # falling back to split bf16): the same two predicates that set self.fold_parts / self.h_rows below
        pad_ = (lambda n: (n + 127) // 128 * 128)
        fold_parts_will = (own_loop and nd == 1 and bool(self.lib.nmfmu_fold_parts_supported(B, R, Lh, T)) and
                           os.environ.get('TORCHNMF_AMD_NMFD_FOLD_PARTS', '1') != '0')
        h_rows_will = (T > 1 and not fold_parts_will and 2 * pad_(B * L) * pad_(Cc) < 2 ** 31 and
                       os.environ.get('TORCHNMF_AMD_NMFD_H_ROWS', '1') != '0')
        f16_ok = (own_loop and float(beta) == 1.0 and aligned and (f16_fold or f16_rows) and
                  (fold_parts_will or h_rows_will))
        # (fold path: Y elements contract over the channels alone; window-operand path: over channels x taps)
        long_enough = (min(Cc, B * L) >= self.F16_MIN_DIM and R * T >= self.F16_MIN_DIM) if f16_fold else \
            min(Cc * T, B * L, R * T) >= self.F16_MIN_DIM
        if precision in (None, 'auto'):
            precision = 'bf16x3'
            if (own_loop and nd == 1 and float(beta) == 1.0 and T >= 128 and (T % 8 or L % 8) and
                    min(Cc, B * L) >= self.F16_MIN_DIM and R * T >= self.F16_MIN_DIM):
                import warnings
                warnings.warn(f"torchnmf_amd: NMFD with {T} taps over {L} frames: the single-plane fp16 mode needs taps and "
                              "frames that are multiples of 8 (implicit Toeplitz operands); precision='auto' falls back to "
                              "split bf16 at three times the matrix work.  Trim or pad the frame axis to a multiple of 8 "
                              "for the fast mode.", stacklevel=3)
            if f16_ok and os.environ.get('TORCHNMF_AMD_AUTO_F16', '1') != '0' and long_enough:
                stats = torch.stack([V.abs().max(), W.abs().max(), H.abs().max(), V.abs().mean(), W.abs().mean(),
                                     H.abs().mean()]).tolist()          # one host sync at engine set-up
                if max(stats[:3]) <= self.F16_MAX_ABS and min(stats[3:]) >= self.F16_MIN_MEAN:
                    precision = 'f16'
        if precision not in _capi.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_capi.PRECISIONS)} or 'auto', got {precision!r}")
        if precision == 'f16' and not f16_ok:
            raise ValueError("precision 'f16' is built for the convolutive models with beta == 1 and taps / frames (of the "
                             "last shift axis) that are multiples of 8 (implicit Toeplitz operands), on shapes whose H "
                             "numerator takes the fold-parts or the window-operand path (ratio planes below 2 GiB); use "
                             "'bf16x3' or 'bf16'")
        self.precision_name = precision
        self.precision = _capi.PRECISIONS[precision]
        x3 = self.precision == _capi.PREC_BF16X3
        self.beta = float(beta)
        self.kl = self.beta == 1.0
        self.gamma, self.l1, self.l2 = mu_gamma(self.beta), float(l1), float(l2)
        self.W, self.H = W, H
        self.B, self.C, self.L, self.R, self.T, self.Lh = B, Cc, L, R, T, Lh
        dev = V.device
        self.tile = 128                                  # GEMM workgroup tile (rows = columns)
        pad = (lambda n: (n + self.tile - 1) // self.tile * self.tile)
        self.c_pad, self.bl_pad, self.rp_pad = pad(Cc), pad(B * L), pad(R * T)
        cp, blp, rpp = self.c_pad, self.bl_pad, self.rp_pad

        # targets: X_w[c][(b,l)] and X_h[(b,l)][c], fp32, zero padded; validation fused into the first gather
        self.flags = torch.tensor([0, 0x7f800000], dtype=torch.int32, device=dev)
        V = V.contiguous()
        self.x_w = torch.empty(cp * blp, dtype=torch.float32, device=dev)
        self.x_h = torch.empty(blp * cp, dtype=torch.float32, device=dev)
        self._pack2d(V, Cc, B * L, 1, L, 0, L, Cc * L, 1, cp, blp, self.x_w, None, self.flags)
        self._pack2d(V, B * L, Cc, L, Cc * L, 1, 1, L, 0, blp, cp, self.x_h, None, None)

        # operand planes
        self.wm = _Planes(cp, rpp, x3, dev)     # [c][(r,t)]
        self.wmt = _Planes(rpp, cp, x3, dev)    # [(r,t)][c]
        # Hu [(b,l)][(r,t)] = H[b][r][l-t] and its transpose: never materialised when the taps and the frame count are
        # multiples of 8 -- the GEMMs then fetch those operands chunk-wise from two window tables of H that are 8x H
        # (nmfmu_conv_tables) instead of T x H; otherwise explicit planes rebuilt every iteration (nmfmu_conv_unfold).
        # With several shift axes (round 4) the same holds when taps and V extent of the LAST axis are multiples of 8: the tables
        # are those of every last-axis line of the zero-padded H (nmfmu_convnd_tables) and the k-chunk's share of the chunk
        # index comes from a small precomputed array (nmfmu_convnd_koff).
        self.implicit = (T % 8 == 0 and L % 8 == 0 if nd == 1 else self.ts[-1] % 8 == 0 and self.ls[-1] % 8 == 0)
        self.implicit = self.implicit and os.environ.get('TORCHNMF_AMD_NMFD_EXPLICIT', '0') != '1'
        self.koff = {}
        if self.implicit and nd > 1:
            nb = self.lib.nmfmu_convnd_table_bytes(B, R, nd, self._lh_arr, self._t_arr)
            self.implicit = 0 < nb < 2 ** 31          # the GEMM lanes address the table with 32-bit byte offsets
        if self.implicit and nd > 1:
            for ops, kp in ((_capi.OPS_B_HU, rpp), (_capi.OPS_B_HUT, blp)):
                host = torch.empty(kp // 8 + 8, dtype=torch.int32)
                _capi.check(self.lib.nmfmu_convnd_koff(ops, B, R, nd, self._lh_arr, self._t_arr, kp, host.data_ptr()),
                            'nmfmu_convnd_koff')
                self.koff[ops] = host.to(dev)
            self.koff[_capi.OPS_A_HU] = self.koff[_capi.OPS_B_HU]
            self.hu = _Table(blp, rpp, nb, x3, dev)
            self.hut = _Table(rpp, blp, nb, x3, dev)
        elif self.implicit and self.lib.nmfmu_conv_table_bytes(B, R, Lh, T) >= 2 ** 31:
            self.implicit = False                     # (32-bit byte offsets into the table: explicit planes beyond 2 GiB)
            self.hu = _Planes(blp, rpp, x3, dev)
            self.hut = _Planes(rpp, blp, x3, dev)
        elif self.implicit:
            nb = self.lib.nmfmu_conv_table_bytes(B, R, Lh, T)
            self.hu = _Table(blp, rpp, nb, x3, dev)     # reversed windows: rows (b,l), k = (r,t)
            self.hut = _Table(rpp, blp, nb, x3, dev)    # forward windows:  rows (r,t), k = (b,l)
        else:
            self.hu = _Planes(blp, rpp, x3, dev)    # [(b,l)][(r,t)]
            self.hut = _Planes(rpp, blp, x3, dev)   # [(r,t)][(b,l)]
        # Ragged channels: with C = 128 k + (1..8) channels (1025 bins at configs[3]) the reconstruction GEMMs run over the
        # first 128 k only and nmfmu_conv_ragged_rows sums the rest directly -- a whole tile row (64 of 576 workgroups,
        # i.e. a second scheduling round: 30 us per reconstruction) for one channel otherwise.
        self.c_main = (Cc // 128) * 128
        self.ragged = (own_loop and nd == 1 and self.c_main >= 128 and 0 < Cc - self.c_main <= 8 and
                       bool(self.lib.nmfmu_conv_ragged_supported(R, T)) and
                       os.environ.get('TORCHNMF_AMD_NMFD_RAGGED', '1') != '0')
        # ... or ride inside the reconstruction GEMMs' own grids as one extra 16 x 16 MFMA block per workgroup
        # (nmfmu_gemm_desc.rag_c0 / rag_channels: no launch of their own; needs >= 1024 whole channels and frames; '0'
        # keeps the separate nmfmu_conv_ragged_rows launches)
        self.ragged_in_grid = (self.ragged and self.implicit and
                               bool(self.lib.nmfmu_gemm_ragged_supported(_capi.OPS_B_HU, self.c_main, blp, Cc - self.c_main)) and
                               bool(self.lib.nmfmu_gemm_ragged_supported(_capi.OPS_A_HU, blp, self.c_main, Cc - self.c_main)) and
                               os.environ.get('TORCHNMF_AMD_NMFD_RAGGED_IN_GRID', '1') != '0')
        # at most 64 channels with several shift axes: the GEMMs' channel side runs 64-row / 64-column tiles (half the MFMA work
        # and explicit-operand traffic of a half-empty 128 tile); the planes keep their 128 pitch
        self.c_rows = 64 if (nd > 1 and self.implicit and Cc <= 64 and
                             os.environ.get('TORCHNMF_AMD_NMFD_NARROW', '1') != '0') else None
        rz = self.ragged or bool(self.c_rows)   # the GEMM then leaves the padding rows / columns of the ratio planes alone
        self.gn = _Planes(cp, blp, x3, dev, rz)     # W half-step ratio, [c][(b,l)]
        self.gnt = _Planes(blp, cp, x3, dev, rz)    # H half-step ratio, [(b,l)][c]
        self.gp = None if self.kl else _Planes(cp, blp, x3, dev, rz)
        self.gpt = None if self.kl else _Planes(blp, cp, x3, dev, rz)
        self.den_w = None if self.kl else torch.empty(cp * rpp, dtype=torch.float32, device=dev)
        # H numerator Y[(r,t)][(b,l)] before the col2im sum: with >= 128 taps the GEMM hands over per-tile diagonal sums
        # (NMFMU_EPI_FOLD, 4 KiB per tile) instead of storing Y (4 R T B L bytes, 105 MB at configs[3])
        self.fold_parts = fold_parts_will
        # tail-round split of the H-numerator GEMM (fold epilogue): its (rp_pad / 128) x (bl_pad / 128) tiles run two per CU;
        # when they make N full rounds of the chip plus at most a quarter round that is whole tile rows (configs[3]: 1600
        # tiles on 512 slots = 3 rounds + the last row of 64), those rows are contraction-split so that the last round
        # costs a fraction of a tile time.  The parts add up in the gather (fixed order).
        self.h_tail_rows, self.h_tail_split = 0, 1
        want = os.environ.get('TORCHNMF_AMD_NMFD_TAIL_SPLIT', '1')      # '0' off, '1' automatic, 'rows,split' forced (tests)
        if self.fold_parts and want != '0':
            slots = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
            self.h_tail_rows, self.h_tail_split = tail_round_split(rpp // 128, blp // 128, slots, -(-Cc // 64), want)
        # H numerator without Y at all (NMFMU_OPS_A_WIN): out[(b,j)][r] = sum_{t,c} GnT[(b, j + t)][c] W[c][r][t] -- the rows of the
        # A operand are shifted rows of the ratio planes the H half-step has just written, so nothing is unfolded or folded
        # (Y is 4 R T B L bytes: 537 MB for a 256 x 512 frame with 8 x 16 taps).  Any number of shift axes, no alignment
        # rules.  The fold-parts path above stays where it applies (1-D, >= 128 taps: it multiplies no padding of the rank).
        self.h_rows = h_rows_will
        assert not (self.precision == _capi.PREC_F16 and not (self.fold_parts or self.h_rows))   # (f16_ok covers it)
        self.y = self.y_den = None
        if self.h_rows:
            self.wk_rows = 32 if R <= 32 else 64 if R <= 64 else pad(R)
            # a small rank would leave most of the 32-wide N tile as padding: F consecutive last-axis taps share a k position
            # and take F columns each (nmfmu_gemm_desc.win_fold) -- 1 / F of the MFMA work and of the operand traffic
            self.wk_fold = h_tap_fold(R, self.ts[-1])
            if os.environ.get('TORCHNMF_AMD_NMFD_H_FOLD', '1') == '0':
                self.wk_fold = 1
            self.wk_klen = (T // self.wk_fold) * (-(-Cc // 64)) * 64
            self.wk = _Planes(self.wk_rows, pad(self.wk_klen), x3, dev)         # W as [(r, d)][(to, q, c)]
            self.hj_pad = pad(B * (Lh // self.lhs[-1]) * (self.lhs[-1] + self.wk_fold - 1))
            # few positions and a long (t, c) contraction (a short spectrogram with many bins): split the contraction over the
            # idle workgroup slots, nmfmu_slab_sum adds the partials
            self.h_ksplit = 1
            if os.environ.get('TORCHNMF_AMD_NMFD_KSPLIT', '1') != '0':
                slots = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
                self.h_ksplit = w_contraction_split((self.hj_pad // 128) * -(-self.wk_rows // 128), self.wk_klen // 64, slots)
            self.hnum = torch.empty(self.h_ksplit * self.hj_pad * self.wk_rows, dtype=torch.float32, device=dev)
            self.hden = None if self.kl else torch.empty(self.h_ksplit * self.hj_pad * self.wk_rows, dtype=torch.float32,
                                                         device=dev)
        else:
            ny = (self.lib.nmfmu_fold_part_bytes(rpp, blp) // 4) * self.h_tail_split if self.fold_parts else rpp * blp
            self.y = torch.empty(ny, dtype=torch.float32, device=dev)
            self.y_den = None if self.kl else torch.empty(ny, dtype=torch.float32, device=dev)
        self.sum_h = torch.zeros(R, dtype=torch.float32, device=dev)   # sum_{b,j} H[b][r][j]
        self.sum_w = torch.zeros(R, dtype=torch.float32, device=dev)   # sum_{c,t} W[c][r][t]
        self.sum_part = torch.empty(R * 128, dtype=torch.float32, device=dev)
        # beta == 1 on the fold-parts path: the rank sums ride in the kernels that produce / consume them
        self.fused_sums = (self.kl and self.fold_parts and os.environ.get('TORCHNMF_AMD_NMFD_FUSED_SUMS', '1') != '0')
        # ... and the H update rewrites the window tables of the new H itself (no nmfmu_conv_tables launch per iteration).
        # Its blocks recompute a halo of their neighbours' elements, for which they need the old values while the
        # neighbours overwrite theirs: two shadow copies of H alternate as "old, read-only" and "new" (see
        # nmfmu_conv_fold_parts_apply_h_tables); _pack_h -- the path every outside change of H takes -- refreshes them.
        self.fused_tables = (self.fused_sums and self.implicit and
                             os.environ.get('TORCHNMF_AMD_NMFD_FUSED_TABLES', '1') != '0')
        self._h_shadow, self._hs, self._hs_valid = None, 0, False
        if self.fused_tables:
            self._h_shadow = [torch.empty_like(H), torch.empty_like(H)]
        if self.fused_sums:
            self.n_hparts = (self.lib.nmfmu_fold_hsum_parts_tables(B, Lh) if self.fused_tables else
                             self.lib.nmfmu_fold_hsum_parts(B, Lh))
            self.hpart = torch.zeros(R * self.n_hparts, dtype=torch.float32, device=dev)
            self.wcol = torch.zeros((cp // 64) * (rpp // 64) * 2, dtype=torch.float32, device=dev)
        self._h_parts_valid = False
        # Launch diet of the window-operand path (round 5; NMF2D / NMF3D / NMFD below 128 taps, beta == 1, >= 64 taps in total,
        # rank <= 256): the rank sums ride in the two apply kernels (tile sums of W out of / partial sums of H into the W
        # update, the reverse for the H update) and the W update emits the Wk planes too -- four launches less per iteration
        # (rank_sums x 3, conv_pack_wk).  '0' keeps the separate launches.
        self.rows_fused = (own_loop and self.kl and self.h_rows and not self.fused_sums and T >= 64 and R <= 256 and
                           os.environ.get('TORCHNMF_AMD_NMFD_ROWS_FUSED', '1') != '0')
        if self.rows_fused:
            self.n_hparts = self.lib.nmfmu_conv_h_rows_parts(B, R, Lh // self.lhs[-1], self.lhs[-1])
            self.hpart = torch.zeros(R * self.n_hparts, dtype=torch.float32, device=dev)
            self.wcol = torch.zeros((cp // 64) * (rpp // 64) * 2, dtype=torch.float32, device=dev)
        # W numerator GEMM: [C x B L] . [B L x R T] has few tiles and a long contraction (225 tiles x 128 k-steps at
        # configs[3]: one workgroup per CU, the second slot idle): split the contraction in two when that fills the chip
        # better; the apply kernel adds the partials (beta == 1 fused-sums path only)
        tiles = (cp // 128) * (rpp // 128)
        self.w_ksplit = 1
        # (k-tiles of that GEMM's contraction as _gemm runs it: the logical B L rounded up to 64 on implicit operands)
        kt_w = (-(-(B * L) // 64)) if (self.implicit and nd == 1) else blp // 64
        if os.environ.get('TORCHNMF_AMD_NMFD_KSPLIT', '1') != '0':
            if self.fused_sums:
                self.w_ksplit = 2 if (tiles <= 256 and kt_w % 2 == 0 and blp >= 2048) else 1
            else:
                slots = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
                self.w_ksplit = w_contraction_split(tiles, kt_w, slots)
        self.num_w = torch.empty(self.w_ksplit * cp * rpp, dtype=torch.float32, device=dev)
        if self.den_w is not None:
            self.den_w = torch.empty(self.w_ksplit * cp * rpp, dtype=torch.float32, device=dev)
        self._loss_main = (self.c_main // 128) * (blp // 128)      # partials of the GEMM part when the channels are ragged
        nrag = self.lib.nmfmu_conv_ragged_blocks(B, Lh, T) * (Cc - self.c_main) if self.ragged else 0
        self.loss_part = torch.zeros((cp // 128) * (blp // 128) + nrag, dtype=torch.float32, device=dev)  # not all written
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=dev)
        self._wk_ready, self._wcol_valid = False, False
        self.refresh_images()
        if self.rows_fused:                          # second pass: now through the fused kernel (tile sums of W for the first H update)
            self._pack_w()

    # ------------------------------------------------------------------ helpers
    def _pack2d(self, src, rows, cols, rin, ros, ris, cin, cos, cis, rows_pad, cols_pad, dst_f32, planes, flags):
        _capi.check(self.lib.nmfmu_pack2d(src.data_ptr(), rows, cols, rin, ros, ris, cin, cos, cis, rows_pad, cols_pad,
                                          _ptr(dst_f32), _ptr(planes.hi) if planes else None,
                                          _ptr(planes.lo) if planes else None, _ptr(flags), _stream()), 'nmfmu_pack2d')

    def _gemm(self, a: _Planes, b: _Planes, epi, x=None, gn=None, gp=None, out=None, m_valid=0, n_valid=0, m_rows=None,
              n_rows=None, k_len=0, k_split=0, tail_rows=0, ragged=False, tag=None):
        """D = A B^T with the given epilogue.  m_rows / n_rows: only the first rows of A / of B (ragged channels); the
        output planes keep their leading dimension.  tag: name of the launch for an attached KernelTimer (bench.py)."""
        assert a.cols_pad == b.cols_pad
        m_pad, n_pad = m_rows or a.rows_pad, n_rows or b.rows_pad
        n_ld = b.rows_pad if n_rows else 0
        ops = _capi.OPS_PLANES
        if self.implicit:
            ops = (_capi.OPS_A_HU if a is self.hu else _capi.OPS_B_HU if b is self.hu else
                   _capi.OPS_B_HUT if b is self.hut else _capi.OPS_PLANES)
        tile = 128
        if not k_len and ops != _capi.OPS_PLANES and self.nd == 1:
            # the contraction of an implicit operand runs over its logical extent rounded up to whole k-tiles, not over the
            # 128-padded pitch of the explicit planes (whose tail is zero): one k-tile less for half of all shapes, and the form
            # the library's window staging asks for (nmfmu_gemm_window_staged: k_len == the logical extent)
            k_len = -(-(self.B * self.L if ops == _capi.OPS_B_HUT else self.R * self.T) // 64) * 64
        d = _capi.GemmDesc(_ptr(a.hi), _ptr(a.lo), _ptr(b.hi), _ptr(b.lo), m_pad, n_pad, a.cols_pad,
                           self.precision, self.beta, _ptr(x), _ptr(gn.hi) if gn else None,
                           _ptr(gn.lo) if gn else None, _ptr(gp.hi) if gp else None, _ptr(gp.lo) if gp else None,
                           _ptr(out), m_valid, n_valid, ops, self.B, self.R, self.T, self.Lh, tile, n_ld, k_len, k_split,
                           tail_rows, self.c_main if ragged else 0, self.C if ragged else 0)
        if ops != _capi.OPS_PLANES and self.nd > 1:
            d.win_nd, d.win_lh, d.win_taps = self.nd, (C.c_int32 * 3)(*self.lhs), (C.c_int32 * 3)(*self.ts)
            d.t_koff = self.koff[ops].data_ptr()
        # implicit operands are staged as a window of table entries wherever the library's shape test admits it (round 5);
        # TORCHNMF_AMD_NMFD_WINSTAGE=0 keeps the chunk-major tiles of rounds 1-4 (A/B switch; bit-identical results)
        d.stage_mode = self._stage_mode
        timer = getattr(self, 'timer', None) if tag else None
        if tag and tag not in self.staged:       # which launches stage their implicit operand as a window (host-side query, once)
            q = int(self.lib.nmfmu_gemm_window_staged(C.byref(d), epi))
            self.staged[tag] = q if q >= 0 else None        # (a negative answer is an error code, not a flag)
        if timer is not None:
            timer.mark(tag + '<')
        _capi.check(self.lib.nmfmu_gemm(C.byref(d), epi, _stream()), 'nmfmu_gemm')
        if timer is not None:
            timer.mark(tag + '>')

    def _gemm_win(self, planes: _Planes, out, tag=None):
        """H numerator (or denominator) as the window-operand GEMM: out[(b,j)][r] from the ratio planes [(b,l)][c]."""
        d = _capi.GemmDesc(_ptr(planes.hi), _ptr(planes.lo), _ptr(self.wk.hi), _ptr(self.wk.lo), self.hj_pad, self.wk_rows,
                           self.wk.cols_pad, self.precision, self.beta, None, None, None, None, None, _ptr(out), 0, 0,
                           _capi.OPS_A_WIN, self.B, self.R, self.T, self.Lh, 128, 0, self.wk_klen, self.h_ksplit, 0, 0, 0,
                           self.nd, (C.c_int32 * 3)(*self.lhs), (C.c_int32 * 3)(*self.ts), self.C, planes.cols_pad,
                           self.wk_fold)
        timer = getattr(self, 'timer', None) if tag else None
        if timer is not None:
            timer.mark(tag + '<')
        _capi.check(self.lib.nmfmu_gemm(C.byref(d), _capi.EPI_F32, _stream()), 'nmfmu_gemm')
        if timer is not None:
            timer.mark(tag + '>')
        if self.h_ksplit > 1:
            _capi.check(self.lib.nmfmu_slab_sum(out.data_ptr(), self.hj_pad * self.wk_rows, self.h_ksplit, _stream()),
                        'nmfmu_slab_sum')

    def _ragged(self, mode, x, gn=None, gp=None):
        """The channels the reconstruction GEMM left out (mode 0: W half-step planes, 1: H half-step planes, 2: loss)."""
        ld = self.bl_pad if mode != 1 else self.c_pad
        loss = self.loss_part.data_ptr() + 4 * self._loss_main if mode == 2 else None
        _capi.check(self.lib.nmfmu_conv_ragged_rows(
            self.W.data_ptr(), self.C, self.R, self.T, self.H.data_ptr(), self.B, self.Lh, self.c_main, self.precision,
            self.beta, mode, x.data_ptr(), ld, _ptr(gn.hi) if gn else None, _ptr(gn.lo) if gn else None,
            _ptr(gp.hi) if gp else None, _ptr(gp.lo) if gp else None, loss, _stream()), 'nmfmu_conv_ragged_rows')

    def _rank_sums(self, src, outer, inner, out):
        """beta == 1 denominators (nmf.py:122-131): sum over everything but the rank axis.  (Running these two small
        kernels on a side stream beside the next GEMM was measured: no gain -- the event fork / join costs what the
        overlap saves; the same holds for the ragged-channel kernel beside its reconstruction GEMM.)"""
        if self.kl:
            _capi.check(self.lib.nmfmu_rank_sums(src.data_ptr(), outer, self.R, inner, self.sum_part.data_ptr(),
                                                 out.data_ptr(), _stream()), 'nmfmu_rank_sums')

    def _pack_w(self, update: bool = False):
        """W -> Wm / WmT planes (one kernel); with ``update`` the MU apply of nmf.py:78-92 runs in the same pass."""
        if self.fused_sums:
            # beta == 1 denominators ride along: sum_h arrives finished or as the partials of the kernel that updated H,
            # the column sums of the new W leave as per-64-channel-tile partials for the H half-step (no rank_sums launches)
            kl = update and self.kl
            _capi.check(self.lib.nmfmu_conv_apply_pack_w_sums(
                self.W.data_ptr(), self.C, self.R, self.T, _ptr(self.num_w) if update else None, None,
                self.sum_h.data_ptr() if (kl and not self._h_parts_valid) else None,
                self.hpart.data_ptr() if (kl and self._h_parts_valid) else None, self.n_hparts, self.wcol.data_ptr(),
                self.w_ksplit, self.c_pad, self.rp_pad, self.l1, self.l2, self.gamma, int(update), self.precision, _ptr(self.wm.hi),
                _ptr(self.wm.lo),
                _ptr(self.wmt.hi), _ptr(self.wmt.lo), _stream()), 'nmfmu_conv_apply_pack_w_sums')
            return
        if self._pack_w_planes(update):
            return                                   # (rows_fused: the Wk planes came out of the same launch)
        if self.h_rows:
            self._wk_ready = True                    # (the standalone kernel also writes the planes' zero padding: once)
            _capi.check(self.lib.nmfmu_conv_pack_wk(self.W.data_ptr(), self.C, self.R, self.T, self.ts[-1], self.wk_fold,
                                                    self.wk.rows_pad, self.wk.cols_pad, self.precision, _ptr(self.wk.hi),
                                                    _ptr(self.wk.lo), _stream()), 'nmfmu_conv_pack_wk')

    def _pack_w_planes(self, update: bool):
        slabs = self.w_ksplit
        if update and (slabs > 2 or (slabs > 1 and self.c_rows)):
            # many split-K partials: a wide reduction first (the apply kernel has few blocks).  A slab holds the rows the GEMM
            # ran over (c_rows of them when the channel side uses the 64-row tile)
            for buf in (self.num_w, self.den_w):
                if buf is not None:
                    _capi.check(self.lib.nmfmu_slab_sum(buf.data_ptr(), (self.c_rows or self.c_pad) * self.rp_pad, slabs,
                                                        _stream()), 'nmfmu_slab_sum')
            slabs = 1
        if self.rows_fused and self._wk_ready:
            # one launch: the update of nmf.py:78-92 with sum_{b,j} H taken from the H update's partial sums (or from the
            # finished vector after an outside change of H), Wm / WmT / Wk planes, tile sums of the new W for the H update
            parts = update and self._h_parts_valid
            _capi.check(self.lib.nmfmu_conv_apply_pack_w_wk(
                self.W.data_ptr(), self.C, self.R, self.T, _ptr(self.num_w) if update else None, None,
                self.sum_h.data_ptr() if (update and not parts) else None, self.hpart.data_ptr() if parts else None,
                self.n_hparts, self.wcol.data_ptr(), slabs, self.c_pad, self.rp_pad, self.l1, self.l2, self.gamma, int(update),
                self.precision, _ptr(self.wm.hi), _ptr(self.wm.lo), _ptr(self.wmt.hi), _ptr(self.wmt.lo), self.ts[-1],
                self.wk_fold, self.wk.rows_pad, self.wk.cols_pad, _ptr(self.wk.hi), _ptr(self.wk.lo), _stream()),
                'nmfmu_conv_apply_pack_w_wk')
            self._wcol_valid = True
            return True
        # (the _sums entry without its partial-sum operands: the one that adds the split-K slabs of num / den)
        _capi.check(self.lib.nmfmu_conv_apply_pack_w_sums(
            self.W.data_ptr(), self.C, self.R, self.T, _ptr(self.num_w) if update else None,
            _ptr(self.den_w) if update else None, self.sum_h.data_ptr() if (update and self.kl) else None, None, 0, None,
            slabs, self.c_pad, self.rp_pad, self.l1, self.l2, self.gamma, int(update), self.precision,
            _ptr(self.wm.hi), _ptr(self.wm.lo), _ptr(self.wmt.hi), _ptr(self.wmt.lo), _stream()),
            'nmfmu_conv_apply_pack_w_sums')
        self._rank_sums(self.W, self.C, self.T, self.sum_w)
        return False

    def _pack_h(self, sums: bool = True):
        if self.implicit and self.nd > 1:
            _capi.check(self.lib.nmfmu_convnd_tables(self.H.data_ptr(), self.B, self.R, self.nd, self._lh_arr, self._t_arr,
                                                     self.precision, _ptr(self.hu.hi), _ptr(self.hu.lo), _ptr(self.hut.hi),
                                                     _ptr(self.hut.lo), _stream()), 'nmfmu_convnd_tables')
        elif self.implicit:
            if self.precision == _capi.PREC_F16:
                _capi.check(self.lib.nmfmu_conv_tables_f16(self.H.data_ptr(), self.B, self.R, self.Lh, self.T,
                                                           _ptr(self.hu.hi), _ptr(self.hut.hi), _stream()),
                            'nmfmu_conv_tables_f16')
            else:
                _capi.check(self.lib.nmfmu_conv_tables(self.H.data_ptr(), self.B, self.R, self.Lh, self.T,
                                                       _ptr(self.hu.hi), _ptr(self.hu.lo), _ptr(self.hut.hi),
                                                       _ptr(self.hut.lo), _stream()), 'nmfmu_conv_tables')
        else:
            self._unfold()
        if self.fused_tables:
            self._h_shadow[self._hs].copy_(self.H)
            self._hs_valid = True
        if sums:
            self._rank_sums(self.H, self.B, self.Lh, self.sum_h)
            self._h_parts_valid = False

    def _unfold(self):
        if self.nd == 1:
            _capi.check(self.lib.nmfmu_conv_unfold(self.H.data_ptr(), self.B, self.R, self.Lh, self.T, _ptr(self.hu.hi),
                                                   _ptr(self.hu.lo), _ptr(self.hut.hi), _ptr(self.hut.lo), self.bl_pad,
                                                   self.rp_pad, _stream()), 'nmfmu_conv_unfold')
        else:
            _capi.check(self.lib.nmfmu_convnd_unfold(self.H.data_ptr(), self.B, self.R, self.nd, self._lh_arr,
                                                     self._t_arr, _ptr(self.hu.hi), _ptr(self.hu.lo), _ptr(self.hut.hi),
                                                     _ptr(self.hut.lo), self.bl_pad, self.rp_pad, _stream()),
                        'nmfmu_convnd_unfold')

    def refresh_images(self):
        self._pack_w()
        self._pack_h()

    # ------------------------------------------------------------------ the fit-loop interface (see nmf.BaseComponent.fit)
    def target_flags(self):
        bad, mn = (int(x) for x in self.flags.tolist())
        return bool(bad), mn == 0

    def recon_ratio_w(self):
        """Reconstruction + ratio planes of the W half-step (nmf.py:61-74 on Wm Hu^T): the GEMM over the channels that
        fill whole tiles, the ragged ones by direct summation."""
        if self.ragged:
            self._gemm(self.wm, self.hu, _capi.EPI_RATIO, x=self.x_w, gn=self.gn, gp=self.gp, m_rows=self.c_main,
                       ragged=self.ragged_in_grid, tag='recon_w')
            if not self.ragged_in_grid:
                self._ragged(0, self.x_w, self.gn, self.gp)
        else:
            self._gemm(self.wm, self.hu, _capi.EPI_RATIO, x=self.x_w, gn=self.gn, gp=self.gp, m_rows=self.c_rows,
                       tag='recon_w')

    def w_step(self):
        """nmf.py:367-378 for the conv1d model."""
        self.recon_ratio_w()
        self._gemm(self.gn, self.hut, _capi.EPI_F32, out=self.num_w, k_split=self.w_ksplit, m_rows=self.c_rows, tag='num_w')
        if not self.kl:
            self._gemm(self.gp, self.hut, _capi.EPI_F32, out=self.den_w, k_split=self.w_ksplit, m_rows=self.c_rows)
        self._pack_w(update=True)

    def h_step(self):
        """nmf.py:380-391 for the conv1d model (uses the freshly updated W)."""
        if self.ragged:
            self._gemm(self.hu, self.wm, _capi.EPI_RATIO, x=self.x_h, gn=self.gnt, gp=self.gpt, n_rows=self.c_main,
                       ragged=self.ragged_in_grid, tag='recon_h')
            if not self.ragged_in_grid:
                self._ragged(1, self.x_h, self.gnt, self.gpt)
        else:
            self._gemm(self.hu, self.wm, _capi.EPI_RATIO, x=self.x_h, gn=self.gnt, gp=self.gpt, n_rows=self.c_rows,
                       tag='recon_h')
        if self.h_rows:
            self._gemm_win(self.gnt, self.hnum, tag='num_h')
            if not self.kl:
                self._gemm_win(self.gpt, self.hden)
            if self.rows_fused and self._wcol_valid:
                _capi.check(self.lib.nmfmu_conv_apply_h_rows_sums(
                    self.H.data_ptr(), self.B, self.R, self.Lh // self.lhs[-1], self.lhs[-1], self.wk_fold, self.hnum.data_ptr(),
                    self.wcol.data_ptr(), self.c_pad // 64, self.rp_pad, self.T, self.wk_rows, self.l1, self.l2, self.gamma,
                    self.hpart.data_ptr(), _stream()), 'nmfmu_conv_apply_h_rows_sums')
                self._pack_h(sums=False)
                self._h_parts_valid = True
                return
            #  with rows_fused the rank sums ride in the fused kernels and self.sum_w is only as fresh as the last
            # unfused pass: this fallback must not be reached once the fused form has run
            assert not (self.rows_fused and self._h_parts_valid), 'stale sum_w: the unfused H update after a fused one'
            _capi.check(self.lib.nmfmu_conv_apply_h_rows(
                self.H.data_ptr(), self.B, self.R, self.Lh // self.lhs[-1], self.lhs[-1], self.wk_fold, self.hnum.data_ptr(),
                _ptr(self.hden),
                self.sum_w.data_ptr() if self.kl else None, self.wk_rows, self.l1, self.l2, self.gamma, _stream()),
                'nmfmu_conv_apply_h_rows')
            self._pack_h()
            return
        epi = _capi.EPI_FOLD if self.fold_parts else _capi.EPI_F32
        kc = -(-self.C // 64) * 64             # the contraction runs over the channels: skip the zero tail of the padding
        tail = dict(k_split=self.h_tail_split, tail_rows=self.h_tail_rows) if self.h_tail_rows else {}
        self._gemm(self.wmt, self.gnt, epi, out=self.y, k_len=kc, tag='num_h', **tail)
        if not self.kl:
            self._gemm(self.wmt, self.gpt, epi, out=self.y_den, k_len=kc, **tail)
        kl_den = self.sum_w.data_ptr() if self.kl else None
        if self.fused_tables and self._hs_valid:
            old, new = self._h_shadow[self._hs], self._h_shadow[self._hs ^ 1]
            _capi.check(self.lib.nmfmu_conv_fold_parts_apply_h_tables(
                self.H.data_ptr(), old.data_ptr(), new.data_ptr(), self.B, self.R, self.Lh, self.T, self.y.data_ptr(), None,
                None, self.wcol.data_ptr(), self.c_pad // 64, self.rp_pad, self.hpart.data_ptr(), self.bl_pad, self.l1,
                self.l2, self.gamma, self.rp_pad, self.h_tail_rows, self.h_tail_split, self.precision, _ptr(self.hu.hi),
                _ptr(self.hu.lo), _ptr(self.hut.hi), _ptr(self.hut.lo), _stream()), 'nmfmu_conv_fold_parts_apply_h_tables')
            self._hs ^= 1
            self._h_parts_valid = True
            return
        if self.fused_sums:
            _capi.check(self.lib.nmfmu_conv_fold_parts_apply_h_tail(
                self.H.data_ptr(), self.B, self.R, self.Lh, self.T, self.y.data_ptr(), None, None, self.wcol.data_ptr(),
                self.c_pad // 64, self.rp_pad, self.hpart.data_ptr(), self.bl_pad, self.l1, self.l2, self.gamma,
                self.rp_pad, self.h_tail_rows, self.h_tail_split, _stream()), 'nmfmu_conv_fold_parts_apply_h_tail')
            self._h_parts_valid = True
            self._pack_h(sums=False)
            return
        if self.fold_parts:
            _capi.check(self.lib.nmfmu_conv_fold_parts_apply_h_tail(
                self.H.data_ptr(), self.B, self.R, self.Lh, self.T, self.y.data_ptr(), _ptr(self.y_den), kl_den, None, 0, 0,
                None, self.bl_pad, self.l1, self.l2, self.gamma, self.rp_pad, self.h_tail_rows, self.h_tail_split,
                _stream()), 'nmfmu_conv_fold_parts_apply_h_tail')
        elif self.nd == 1:
            _capi.check(self.lib.nmfmu_conv_fold_apply_h(self.H.data_ptr(), self.B, self.R, self.Lh, self.T,
                                                         self.y.data_ptr(), _ptr(self.y_den), kl_den, self.bl_pad,
                                                         self.l1, self.l2, self.gamma, _stream()),
                        'nmfmu_conv_fold_apply_h')
        else:
            _capi.check(self.lib.nmfmu_convnd_fold_apply_h(self.H.data_ptr(), self.B, self.R, self.nd, self._lh_arr,
                                                           self._t_arr, self.y.data_ptr(), _ptr(self.y_den), kl_den,
                                                           self.bl_pad, self.l1, self.l2, self.gamma, _stream()),
                        'nmfmu_convnd_fold_apply_h')
        self._pack_h()

    def left_f16_range(self) -> bool:
        """'f16' only: the range gate of 'auto' / the constructor looks at the initial data.  A factor value
        that grows beyond fp16's largest finite number during the fit is clamped to 65504 in the operand planes / window
        tables, and the updates stop following the reference.  fit() asks at its loss checkpoints (one more host sync
        there, two small reductions) and warns."""
        if self.precision != _capi.PREC_F16:
            return False
        return bool(torch.maximum(self.W.max(), self.H.max()).item() > 65504.0)

    def _loss_device(self):
        """Enqueue beta_div(conv1d reconstruction, V) (nmf.py:360-361 / 400-401); the value as a float64[1] device tensor."""
        self._gemm(self.wm, self.hu, _capi.EPI_LOSS, x=self.x_w, out=self.loss_part, m_valid=self.C,
                   n_valid=self.B * self.L, m_rows=self.c_main if self.ragged else self.c_rows)
        if self.ragged:
            self._ragged(2, self.x_w)
        return self.loss_part.double().sum().reshape(1)

    def _range_flag_device(self):
        if self.precision != _capi.PREC_F16:
            return None
        return (torch.maximum(self.W.max(), self.H.max()) > 65504.0).double().reshape(1)

    def _ckpt_tensors(self):
        return [self.W, self.H]

    def divergence(self) -> float:
        """One host sync."""
        return float(self._loss_device().item())


class WideRankMU:
    """``NMF.fit`` for ranks above 256.  The fused kernel keeps a rank-wide accumulator tile in registers, which
    stops at a padded rank of 256; a wider NMF is the T = 1 member of the NMFD family -- V^T as (1, C, N), W as
    (C, R, 1), H^T as (1, R, N) -- and runs on the GEMM engine, whose effective rank R*T is unbounded.  W shares
    storage with the parameter; H is kept transposed and copied back after every H half-step (N x R floats)."""


    def __init__(self, V, W, H, beta, l1=0.0, l2=0.0, precision='auto', update_W=True, update_H=True):
        assert V.dim() == 2 and W.dim() == 2 and H.dim() == 2
        self.H_user = H
        self.Ht = H.t().contiguous().unsqueeze(0)
        self.eng = ConvMU(V.t().contiguous().unsqueeze(0), W.unsqueeze(2), self.Ht, beta, l1, l2, precision=precision,
                          update_W=update_W, update_H=update_H)
        self.precision_name = self.eng.precision_name

    def target_flags(self):
        return self.eng.target_flags()

    def w_step(self):
        self.eng.w_step()

    def h_step(self):
        self.eng.h_step()
        self.H_user.copy_(self.Ht[0].t())

    def divergence(self) -> float:
        return self.eng.divergence()


def reconstruct(H: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """``NMFD / NMF2D / NMF3D.reconstruct`` (nmf.py:776-779, 857-860, 937-940) on the device: Wm @ Hu^T through the
    split-bf16 GEMM, fp32 out."""
    if H.device.type != 'cuda' or W.device.type != 'cuda':
        raise _capi.NmfmuError('reconstruct: tensors must live on the ROCm device (no CPU fallback)')
    lib = _capi.load()
    Hc, Wc = H.detach().float().contiguous(), W.detach().float().contiguous()
    nd = Hc.dim() - 2
    assert nd in (1, 2, 3) and Wc.dim() == Hc.dim()
    B, R = Hc.shape[:2]
    Cc, R2 = Wc.shape[:2]
    assert R == R2
    lhs, ts = tuple(Hc.shape[2:]), tuple(Wc.shape[2:])
    ls = tuple(lh + t - 1 for lh, t in zip(lhs, ts))
    L, T = int(torch.tensor(ls).prod()), int(torch.tensor(ts).prod())
    dev = H.device
    cp, blp, rpp = _pad128(Cc), _pad128(B * L), _pad128(R * T)
    wm, hu, hut = _Planes(cp, rpp, True, dev), _Planes(blp, rpp, True, dev), _Planes(rpp, blp, True, dev)
    RT = R * T
    _capi.check(lib.nmfmu_pack2d(Wc.data_ptr(), Cc, RT, 1, RT, 0, 1, 1, 0, cp, rpp, None, _ptr(wm.hi), _ptr(wm.lo), None,
                                 _stream()), 'nmfmu_pack2d')
    if nd == 1:
        _capi.check(lib.nmfmu_conv_unfold(Hc.data_ptr(), B, R, lhs[0], T, _ptr(hu.hi), _ptr(hu.lo), _ptr(hut.hi),
                                          _ptr(hut.lo), blp, rpp, _stream()), 'nmfmu_conv_unfold')
    else:
        _capi.check(lib.nmfmu_convnd_unfold(Hc.data_ptr(), B, R, nd, (C.c_int32 * nd)(*lhs), (C.c_int32 * nd)(*ts),
                                            _ptr(hu.hi), _ptr(hu.lo), _ptr(hut.hi), _ptr(hut.lo), blp, rpp, _stream()),
                    'nmfmu_convnd_unfold')
    out = torch.empty(cp, blp, dtype=torch.float32, device=dev)
    d = _capi.GemmDesc(_ptr(wm.hi), _ptr(wm.lo), _ptr(hu.hi), _ptr(hu.lo), cp, blp, rpp, _capi.PREC_BF16X3, 2.0, None,
                       None, None, None, None, out.data_ptr(), 0, 0, _capi.OPS_PLANES, 0, 0, 0, 0)
    _capi.check(lib.nmfmu_gemm(C.byref(d), _capi.EPI_F32, _stream()), 'nmfmu_gemm')
    return out[:Cc, :B * L].reshape(Cc, B, *ls).transpose(0, 1).contiguous()
