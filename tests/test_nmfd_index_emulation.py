"""CPU restatements (numpy) of the index algebra of the round-2 NMFD kernels, checked against direct formulas.

* EPI_FOLD epilogue of nt_gemm_kernel + conv_fold_parts_apply_h_kernel (csrc/nmfmu_gemm.h, csrc/nmfmu_nmfd.hip): the H
  numerator neg[b][r][j] = sum_t Y[(r,t)][(b, j+t)] (col2im of conv1d's backward pass, nmf.py:77/82 of the reference)
  from per-tile diagonal sums part[(tm, tn)][seg][dd] -- never from Y itself.
* the beta == 1 denominators handed from kernel to kernel as partial sums: per 64 x 64 tile and rank for W
  (conv_apply_pack_w_kernel -> the H update), per block for H (the H update -> conv_apply_pack_w_kernel).
* the implicit Toeplitz operand's chunk index (row + scalar k position) against Hu[(b,l)][(r,t)] = H[b][r][l-t].

If one of these mirrors and the device code drift apart the GPU parity tests fail; if the algebra itself is wrong, these
fail without a GPU.
"""
import numpy as np
import pytest


def fold_parts_of_tile(Y, tm, tn, T, L):
    """What the EPI_FOLD epilogue emits for the 128 x 128 tile (tm, tn) of Y [m_pad][n_pad]: [4][256]."""
    out = np.zeros((4, 256))
    rb = (tm * 128 // T + 1) * T - tm * 128          # first tile row of the second r (>= 128: none)
    nb = (tn * 128 // L + 1) * L - tn * 128          # first tile column of the second b
    tile = Y[tm * 128:(tm + 1) * 128, tn * 128:(tn + 1) * 128]
    for dd in range(255):
        lo, hi1 = max(0, 127 - dd), min(127, 254 - dd) + 1
        rsw = min(max(rb, lo), hi1)
        bsw = min(max(nb - dd + 127, lo), hi1)
        run = lambda r0, r1: sum(tile[ml, ml + dd - 127] for ml in range(r0, r1))
        out[0, dd] = run(lo, min(rsw, bsw))
        out[1, dd] = run(max(lo, bsw), rsw)
        out[2, dd] = run(rsw, max(rsw, bsw))
        out[3, dd] = run(max(rsw, bsw), hi1)
    return out


def gather(parts, tiles_n, b, r, jx, T, L):
    """conv_fold_parts_apply_h_kernel's sum for one (b, r, j)."""
    m_lo, m_hi = r * T, r * T + T
    diag = jx + b * L - m_lo
    neg = 0.0
    for tm in range(m_lo // 128, (m_hi - 1) // 128 + 1):
        ta, tb = max(m_lo, tm * 128) - m_lo, min(m_hi, tm * 128 + 128) - m_lo
        rbit = 2 if r > (tm * 128) // T else 0
        na, nz = b * L + jx + ta, b * L + jx + tb - 1
        for tn in range(na // 128, nz // 128 + 1):
            seg = rbit + (1 if b > (tn * 128) // L else 0)
            dd = diag - 128 * (tn - tm) + 127
            assert 0 <= dd <= 254
            neg += parts[tm * tiles_n + tn][seg, dd]
    return neg


@pytest.mark.parametrize('B,R,T,Lh', [(1, 2, 128, 130), (2, 3, 130, 77), (1, 1, 400, 201), (3, 2, 136, 200), (2, 2, 257, 1)])
def test_fold_from_tile_diagonal_sums(B, R, T, Lh):
    L = Lh + T - 1
    rng = np.random.default_rng(B * 1000 + T)
    m_pad, n_pad = -(-R * T // 128) * 128, -(-B * L // 128) * 128
    Y = np.zeros((m_pad, n_pad))
    Y[:R * T, :B * L] = rng.random((R * T, B * L))     # padding rows / columns are exact zeros on the device too
    tiles_m, tiles_n = m_pad // 128, n_pad // 128
    parts = [fold_parts_of_tile(Y, tm, tn, T, L) for tm in range(tiles_m) for tn in range(tiles_n)]
    for b in range(B):
        for r in range(R):
            for jx in sorted({0, min(1, Lh - 1), Lh // 2, Lh - 1}):
                want = sum(Y[r * T + t, b * L + jx + t] for t in range(T))
                assert gather(parts, tiles_n, b, r, jx, T, L) == pytest.approx(want, rel=1e-12)
    # every element of Y that belongs to some (b, r, j) is counted exactly once: sum over all j of the gathers
    total = sum(gather(parts, tiles_n, b, r, jx, T, L) for b in range(B) for r in range(R) for jx in range(Lh))
    want = sum(Y[r * T + t, b * L + jx + t] for b in range(B) for r in range(R) for jx in range(Lh) for t in range(T))
    assert total == pytest.approx(want, rel=1e-10)


@pytest.mark.parametrize('C,R,T', [(70, 3, 64), (130, 2, 136), (257, 8, 400), (64, 5, 100)])
def test_w_rank_sums_from_tile_sums(C, R, T):
    """conv_apply_pack_w_kernel leaves, per 64 x 64 tile (channel tile ct, k tile kt) of Wm [C][R T], the sums of the first
    and of the second rank in the tile; the H update finishes sum_{c,t} W[c][r][t] from the tiles that hold taps of r."""
    rng = np.random.default_rng(C + T)
    W = rng.random((C, R, T))
    RT = R * T
    c_pad, rp_pad = -(-C // 64) * 64, -(-RT // 64) * 64
    Wm = np.zeros((c_pad, rp_pad))
    Wm[:C, :RT] = W.reshape(C, RT)
    c_tiles, k_tiles = c_pad // 64, rp_pad // 64
    wcol = np.zeros((c_tiles, k_tiles, 2))
    for ct in range(c_tiles):
        for kt in range(k_tiles):
            r_lo = kt * 64 // T
            for kl in range(64):
                second = (kt * 64 + kl) // T != r_lo
                wcol[ct, kt, int(second)] += Wm[ct * 64:(ct + 1) * 64, kt * 64 + kl].sum()
    for r in range(R):
        kt_lo = r * T // 64
        kt_n = (r * T + T - 1) // 64 - kt_lo + 1
        got = sum(wcol[ct, kt, 1 if r > (kt * 64) // T else 0] for ct in range(c_tiles) for kt in range(kt_lo, kt_lo + kt_n))
        assert got == pytest.approx(W[:, r, :].sum(), rel=1e-12)


@pytest.mark.parametrize('B,R,Lh', [(1, 8, 7793), (3, 2, 300), (2, 5, 256)])
def test_h_rank_sums_from_block_sums(B, R, Lh):
    """The H update leaves one partial per (r, b, 256-frame block); conv_apply_pack_w sums hsum_part[r][:] (any order of
    blocks, fixed on the device)."""
    rng = np.random.default_rng(Lh)
    H = rng.random((B, R, Lh))
    jblocks = -(-Lh // 256)
    part = np.zeros((R, B * jblocks))
    for b in range(B):
        for r in range(R):
            for jb in range(jblocks):
                part[r, b * jblocks + jb] = H[b, r, jb * 256:(jb + 1) * 256].sum()
    assert np.allclose(part.sum(1), H.sum((0, 2)), rtol=1e-12)


@pytest.mark.parametrize('B,R,T,Lh', [(1, 2, 8, 25), (2, 3, 16, 41), (1, 1, 24, 9)])
def test_implicit_toeplitz_chunk_index(B, R, T, Lh):
    """Window tables (nmfmu_conv_tables) and the GEMM's chunk index: rows-(b,l) operand, k = (r, 8 tc .. 8 tc + 7) is the
    reversed window at j = l - 8 tc of (b, r): index 1 + (b R + r) JJ + (l - 8 tc) + (T - 1), i.e. the kernel's
    (b R JJ + l + T - 1) + (1 + r JJ - 8 tc) = per-lane row part + wave-uniform k part."""
    L, JJ = Lh + T - 1, Lh + 2 * T - 2
    rng = np.random.default_rng(T)
    H = rng.random((B, R, Lh))
    rev = np.zeros((1 + B * R * JJ, 8))
    fwd = np.zeros((1 + B * R * JJ, 8))
    for br in range(B * R):
        for jj in range(JJ):
            j = jj - (T - 1)
            for e in range(8):
                if 0 <= j - e < Lh:
                    rev[1 + br * JJ + jj, e] = H[br // R, br % R, j - e]
                if 0 <= j + e < Lh:
                    fwd[1 + br * JJ + jj, e] = H[br // R, br % R, j + e]
    Hu = np.zeros((B * L, R * T))
    for b in range(B):
        for l in range(L):
            for r in range(R):
                for t in range(T):
                    if 0 <= l - t < Lh:
                        Hu[b * L + l, r * T + t] = H[b, r, l - t]
    for row in range(B * L):
        b, l = divmod(row, L)
        trow = b * R * JJ + l + T - 1                       # per-lane part
        for r in range(R):
            for tc in range(T // 8):
                soff = 1 + r * JJ - 8 * tc                   # wave-uniform part (kq = r, kr = tc)
                assert np.array_equal(rev[trow + soff], Hu[row, r * T + 8 * tc:r * T + 8 * tc + 8])
    # rows-(r,t) operand (HuT), k = (b, l0 .. l0 + 7): forward window at j = l0 - t
    for r in range(R):
        for t in range(T):
            trow = r * JJ - t + T - 1
            for b in range(B):
                for l0 in range(0, L - 7, 8):
                    soff = 1 + b * R * JJ + l0
                    assert np.array_equal(fwd[trow + soff], Hu[b * L + l0:b * L + l0 + 8, r * T + t])


@pytest.mark.parametrize('B,R,T,Lh,C,tail_rows,k_split', [(1, 2, 136, 200, 600, 1, 3), (2, 2, 130, 77, 520, 2, 2), (1, 1, 400, 201, 1025, 4, 6)])
def test_tail_round_split_grid_mapping_and_gather(B, R, T, Lh, C, tail_rows, k_split):
    """Round 3: the tail-round split of the H-numerator GEMM.  Grid row y of nt_gemm_kernel<EPI_FOLD> -> (tile row, contraction
    part): rows below m_tiles - tail_rows appear once, the last tail_rows rows k_split times (dispatched last); part z
    runs k-tiles [z * per, min((z + 1) * per, ktiles)) and writes slab z; the gather adds the slabs of those rows.  The
    mapping must cover every (tile row, k-tile) exactly once and the gathered sums must equal the unsplit ones."""
    L = Lh + T - 1
    rng = np.random.default_rng(T + C)
    m_pad, n_pad = -(-R * T // 128) * 128, -(-B * L // 128) * 128
    tiles_m, tiles_n = m_pad // 128, n_pad // 128
    tail_rows = min(tail_rows, tiles_m)
    ktiles = -(-C // 64)
    per = -(-ktiles // k_split)
    assert (k_split - 1) * per < ktiles, 'the host normalises the split so that every part is non-empty'
    # operands of Y = Wm^T-planes [m_pad][C] x Gn^T-planes [n_pad][C], zero padded
    A = np.zeros((m_pad, ktiles * 64)); A[:R * T, :C] = rng.random((R * T, C))
    Bm = np.zeros((n_pad, ktiles * 64)); Bm[:B * L, :C] = rng.random((B * L, C))
    covered = np.zeros((tiles_m, ktiles), dtype=int)
    slabs = [[None] * (tiles_m * tiles_n) for _ in range(k_split)]
    row0 = tiles_m - tail_rows
    for y in range(tiles_m + tail_rows * (k_split - 1)):            # gridDim.y of the launch
        bm, zz, nsp = y, 0, 1
        if tail_rows > 0 and y >= row0:
            q = y - row0
            zz, bm, nsp = q % k_split, row0 + q // k_split, k_split
        kt_per = -(-ktiles // nsp)
        kt0 = zz * kt_per
        kts = max(0, min(kt_per, ktiles - kt0))
        covered[bm, kt0:kt0 + kts] += 1
        Ypart = A[bm * 128:(bm + 1) * 128, kt0 * 64:(kt0 + kts) * 64] @ Bm[:, kt0 * 64:(kt0 + kts) * 64].T    # this part's tile row of Y
        Yfull = np.zeros((m_pad, n_pad)); Yfull[bm * 128:(bm + 1) * 128] = Ypart
        for tn in range(tiles_n):
            slabs[zz][bm * tiles_n + tn] = fold_parts_of_tile(Yfull, bm, tn, T, L)
    assert (covered == 1).all()
    Y = A @ Bm.T
    want_parts = [fold_parts_of_tile(Y, tm, tn, T, L) for tm in range(tiles_m) for tn in range(tiles_n)]
    # the gather of conv_fold_parts_apply_h_kernel with (tail_tm0, tail_split): slab 0 everywhere, slabs 1.. for the tail rows
    summed = []
    for tm in range(tiles_m):
        for tn in range(tiles_n):
            nz = k_split if tm >= row0 else 1
            summed.append(sum(slabs[z][tm * tiles_n + tn] for z in range(nz)))
    for i in range(len(summed)):
        np.testing.assert_allclose(summed[i], want_parts[i], rtol=1e-12, atol=1e-9)
    for (b, r, jx) in [(0, 0, 0), (B - 1, R - 1, Lh - 1), (0, R - 1, Lh // 2)]:
        direct = sum(Y[r * T + t, b * L + jx + t] for t in range(T))
        assert gather(summed, tiles_n, b, r, jx, T, L) == pytest.approx(direct, rel=1e-10)


@pytest.mark.parametrize('ops', ['B_HU', 'A_HU'])
def test_ragged_channels_as_an_extra_mfma_block(ops):
    """Round 3: ragged channels inside the reconstruction GEMM's grid (nmfmu_gemm.h, RAGK).  Per k-tile a 16-row tile of
    the explicit operand (rows rag_c0 .. rag_c0 + 15 of the W planes) lands in LDS row-major with the tiles' source-side
    XOR swizzle (slot s of row r holds chunk s ^ ((r >> 1) & 7)); the implicit operand's tile is chunk-major
    [8 chunks][128 rows].  Wave 0 reads 16 x 16 x 32 fragments -- lane = (row lane & 15, chunk 4 ks2 + (lane >> 4)) -- and
    workgroup (bm, bn) takes frames 16 sub .. 16 sub + 15 of its implicit tile (sub = bm for B_HU, bn for A_HU, < 8).
    Emulated with the kernel's address arithmetic and the MFMA's documented lane mapping: the eight workgroups of a tile
    column / row cover its 128 frames exactly once and every output equals the plain contraction."""
    rng = np.random.default_rng(5)
    k_tiles, tiles_other = 3, 9                       # contraction 192; nine tiles along the explicit operand (eight take part)
    K = 64 * k_tiles
    E = rng.integers(-4, 5, size=(16, K)).astype(np.float64)          # explicit rows rag_c0 .. +15 (only some are real channels)
    n_imp_tiles = 2
    I = rng.integers(-4, 5, size=(128 * n_imp_tiles, K)).astype(np.float64)   # implicit operand rows (b,l)
    want = E @ I.T                                                    # [16 channels][(b,l)]
    got = np.full_like(want, np.nan)
    hits = np.zeros_like(want, dtype=int)
    for it in range(n_imp_tiles):                     # tile column (B_HU: bn) resp. tile row (A_HU: bm) of the implicit operand
        for other in range(tiles_other):              # B_HU: bm, A_HU: bn
            sub = other
            if sub >= 8:
                continue
            acc = np.zeros((16, 16))                  # D[row][col] of the 16 x 16 block
            for kt in range(k_tiles):
                # LDS images of this stage
                rag = np.zeros((16, 8, 8))            # [row][slot][8 elements]
                for tid in range(128):                # waves 0, 1: one dwordx4 each, LDS offset = tid * 16 (linear)
                    row, slot = tid >> 3, tid & 7
                    sslot = slot ^ ((row >> 1) & 7)
                    rag[row, slot] = E[row, kt * 64 + sslot * 8: kt * 64 + sslot * 8 + 8]
                imp = np.zeros((8, 128, 8))           # chunk-major [chunk][row][8 elements]
                for q in range(8):
                    imp[q] = I[it * 128:(it + 1) * 128, kt * 64 + q * 8: kt * 64 + q * 8 + 8]
                for ks2 in range(2):
                    Af = np.zeros((16, 32)); Bf = np.zeros((16, 32))
                    for lane in range(64):
                        r16, g4 = lane & 15, lane >> 4
                        q = 4 * ks2 + g4
                        eo = r16 * 128 + ((q ^ ((r16 >> 1) & 7)) << 4)            # byte offset in the ragged tile
                        e = rag[eo // 128, (eo % 128) // 16]
                        io = (q * 128 + 16 * sub + r16) * 16                      # byte offset in the chunk-major tile
                        i = imp[io // (128 * 16), (io // 16) % 128]
                        a, b = (e, i) if ops == 'B_HU' else (i, e)
                        Af[r16, 8 * g4: 8 * g4 + 8] = a                           # A / B lane = (row lane & 15, k = 8 (lane >> 4) + i)
                        Bf[r16, 8 * g4: 8 * g4 + 8] = b
                    acc += Af @ Bf.T
            for lane in range(64):                    # C / D lane = (column lane & 15, rows 4 (lane >> 4) + i)
                r16, g4 = lane & 15, lane >> 4
                for i in range(4):
                    row = 4 * g4 + i
                    if ops == 'B_HU':                 # rows = channels, columns = frames
                        c, frame = row, it * 128 + 16 * sub + r16
                    else:                             # rows = frames, columns = channels
                        c, frame = r16, it * 128 + 16 * sub + row
                    got[c, frame] = acc[row, r16]
                    hits[c, frame] += 1
    assert (hits == 1).all()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('B,R,T,Lh', [(1, 2, 136, 385), (2, 1, 8, 9), (1, 3, 16, 242 - 14), (1, 1, 24, 242 - 13), (2, 2, 128, 700)])
def test_h_update_blocks_rewrite_the_window_tables(B, R, T, Lh):
    """Round 3: conv_fold_parts_apply_h_kernel<TAB>.  A block covers 256 positions jx = jb * 242 + tid - 14 of one (b, r)
    row, owns tid 7 .. 248 and recomputes the rest as halo; positions run over [-7, Lh + 6] (H = 0 outside [0, Lh)).
    Owned threads write the reversed window {H[j] .. H[j-7]} for 0 <= j <= Lh + 6 and the forward window {H[j] .. H[j+7]}
    for -7 <= j <= Lh - 1 at entry 1 + (b R + r) JJ + j + (T - 1).  Emulated with the kernel's index arithmetic: every
    element of H is stored exactly once, and the tables equal what conv_tables_kernel builds (all other entries zero)."""
    OWN = 242
    JJ = Lh + 2 * T - 2
    rng = np.random.default_rng(Lh)
    Hnew = rng.random((B, R, Lh)) + 0.5
    n = 1 + B * R * JJ
    # reference tables (conv_tables_kernel)
    rev_want, fwd_want = np.zeros((n, 8)), np.zeros((n, 8))
    for i in range(1, n):
        k = i - 1
        br, j = k // JJ, k % JJ - (T - 1)
        for e in range(8):
            if 0 <= j - e < Lh:
                rev_want[i, e] = Hnew.reshape(B * R, Lh)[br, j - e]
            if 0 <= j + e < Lh:
                fwd_want[i, e] = Hnew.reshape(B * R, Lh)[br, j + e]
    rev, fwd = np.zeros((n, 8)), np.zeros((n, 8))       # set-up state: every entry once by the standalone kernel (zeros here)
    stored = np.zeros((B, R, Lh), dtype=int)
    jblocks = (Lh + 14 + OWN - 1) // OWN
    for blk in range(B * R * jblocks):
        jb, r, b = blk % jblocks, (blk // jblocks) % R, blk // (jblocks * R)
        hl = np.zeros(256)
        for tid in range(256):
            jx = jb * OWN + tid - 14
            if 0 <= jx < Lh:
                hl[tid] = Hnew[b, r, jx]                 # (the update itself: any function of the OLD shadow and the parts)
                if 7 <= tid < 7 + OWN:
                    stored[b, r, jx] += 1
        for tid in range(7, 7 + OWN):
            jx = jb * OWN + tid - 14
            assert jx >= -7
            e = 1 + (b * R + r) * JJ + jx + T - 1
            if jx <= Lh + 6:
                if jx >= 0:
                    rev[e] = [hl[tid - q] for q in range(8)]
                if jx <= Lh - 1:
                    fwd[e] = [hl[tid + q] for q in range(8)]
    assert (stored == 1).all()
    np.testing.assert_array_equal(rev, rev_want)
    np.testing.assert_array_equal(fwd, fwd_want)


# ----------------------------------------------------------------------------------------------------------------------
# Window staging of the implicit Toeplitz operand (round 5; csrc/nmfmu_gemm.h, template flag WS): a k-tile's implicit
# operand reaches LDS as the window of DISTINCT table entries it touches (two regions of 192 slots) instead of a
# chunk-major 8 x 128 tile.  The mirror below follows the kernel statement by statement -- per-thread slot entries ws_tsl,
# the per-k-tile scalar offsets / region-1 rule of stage_issue, the lane bases ws_base and the k-step immediates of
# load_frags -- and every fragment (row, chunk) of every k-tile must name the table entry whose window IS the operand:
#   rows (b,l), k = (r,t): reversed window at j = l - t of line (b, r)   (Hu[(b,l)][(r,t..t+7)]  = H[b][r][l-t .. l-t-7])
#   rows (r,t), k = (b,l): forward  window at j = l - t of line (b, r)   (HuT[(r,t)][(b,l..l+7)] = H[b][r][l-t .. l-t+7])
# with the table layout of nmfmu_conv_tables: entry 1 + (b R + r) JJ + j + (T - 1), JJ = Lh + 2 T - 2.
# ----------------------------------------------------------------------------------------------------------------------
WIN_REG = 192


def ws_stageable(ops, B, R, T, Lh, rows, k_len):
    """gemm_window_stageable()."""
    L = Lh + T - 1
    if ops == 'B_HUT':
        return rows == R * T and (R * T) % 128 == 0 and T >= 128 and L % 64 == 0 and k_len == B * L
    return rows == B * L and L % 128 == 0 and T >= 64 and k_len == R * T and (R * T) % 64 == 0


def ws_emulate_tile(ops, B, R, T, Lh, tile, kt0, ktiles):
    """For tile `tile` of the implicit operand (128 rows) and k-tiles kt0 .. kt0 + ktiles - 1: yields (kt, row, q, entry) for
    every fragment the MFMAs consume, `entry` = what sits in the LDS slot the lane reads (the kernel's arithmetic)."""
    hu_rows = ops != 'B_HUT'
    L, JJ, T8 = Lh + T - 1, Lh + 2 * T - 2, T // 8
    nent = 1 + B * R * JJ
    row0 = tile * 128
    u = np.arange(256)                                   # tid; waves 0-2 (u < 192) issue
    if hu_rows:
        b, l0 = divmod(row0, L)
        tsl = [b * R * JJ + l0 + T - 1 + (u - 56)] * 2
        q0, r0 = divmod(kt0 * 8, T8)
        two = False
    else:
        rr, t0 = divmod(row0, T)
        n0 = min(128, T - t0)
        n1 = 128 - n0
        two = n1 > 0
        tsl = [rr * JJ - t0 - n0 + T + u, (rr + 1) * JJ - n1 + T + u]
        q0, r0 = divmod(kt0 * 64, L)
    for kt in range(kt0, kt0 + ktiles):
        # ---- stage_issue
        lds = np.full(2 * WIN_REG, -1, dtype=np.int64)   # slot -> table entry (-1: never written in this stage)
        if hu_rows:
            qs = T8 - r0
            need1 = qs < 8
            soff = [1 + q0 * JJ - 8 * r0, 1 + (q0 + 1) * JJ + 8 * qs]
            qs_cur = qs if need1 else 8
        else:
            soff = [1 + q0 * R * JJ + r0] * 2
            need1, qs_cur = two, 8
        lds[:WIN_REG] = np.clip(tsl[0][:WIN_REG] + soff[0], 0, nent - 1)
        if need1:
            lds[WIN_REG:] = np.clip(tsl[1][:WIN_REG] + soff[1], 0, nent - 1)
        # ---- load_frags: every lane (j, hl) of every 32-row block, every k-step
        for row in range(128):
            for ks in range(4):
                for hl in range(2):
                    q = 2 * ks + hl
                    if hu_rows:
                        base = row * 16 + 128 - 128 * hl
                        wo = (3 - ks) * 256 + (WIN_REG * 16 if q >= qs_cur else 0)
                    else:
                        seg = row >= n0
                        rl, ng = (row - n0, n1) if seg else (row, n0)
                        base = ((ng - 1 - rl) + (WIN_REG if seg else 0) + 8 * hl) * 16
                        wo = ks * 256
                    addr = base + wo
                    assert addr % 16 == 0 and 0 <= addr < 2 * WIN_REG * 16
                    yield kt, row, q, int(lds[addr // 16])
        # ---- advance
        if hu_rows:
            r0 += 8
            if r0 >= T8:
                r0 -= T8
                q0 += 1
        else:
            r0 += 64
            if r0 >= L:
                r0 -= L
                q0 += 1


@pytest.mark.parametrize('ops,B,R,T,Lh,kt0', [
    ('B_HU', 1, 8, 400, 7793, 0),        # BASELINE configs[3]: L = 8192, R T = 3200; k-tiles straddle r at 384 + 16
    ('B_HU', 1, 8, 400, 7793, 25),       # the second half of a contraction split in two
    ('A_HU', 2, 3, 64, 193, 0),          # T = 64: every r boundary sits on a k-tile boundary, two batch entries
    ('B_HU', 3, 8, 72, 57, 0),           # T8 = 9 (odd): boundaries inside a k-step (hl halves in different regions); L = 128
    ('A_HU', 1, 16, 88, 297, 2),         # T8 = 11, L = 384, R T = 1408 = 22 k-tiles
    ('B_HU', 1, 5, 104, 281, 3),         # T8 = 13, L = 384, R T = 520 is not a multiple of 64 -> not stageable (checked below)
    ('B_HUT', 1, 8, 400, 7793, 0),       # configs[3] W numerator: rows (r,t) 3200 = 25 tiles, tiles 3, 6, .. span two r
    ('B_HUT', 2, 3, 128, 65, 0),         # T = 128: one r per tile, L = 192: k-tiles stay inside a batch entry
    ('B_HUT', 2, 2, 192, 129, 2),        # T = 192: every second tile spans two r; L = 320
])
def test_window_staging_index_algebra(ops, B, R, T, Lh, kt0):
    L, JJ = Lh + T - 1, Lh + 2 * T - 2
    hu_rows = ops != 'B_HUT'
    rows, k_len = (B * L, R * T) if hu_rows else (R * T, B * L)
    if not ws_stageable(ops, B, R, T, Lh, rows, k_len):
        assert (R * T) % 64 != 0          # the one deliberately inadmissible case of the list
        return
    ktiles_all = k_len // 64
    tiles = sorted({0, 1, rows // 128 // 2, rows // 128 - 1})
    for tile in tiles:
        n = 0
        for kt, row, q, entry in ws_emulate_tile(ops, B, R, T, Lh, tile, kt0, min(ktiles_all - kt0, 9)):
            m, k = tile * 128 + row, kt * 64 + 8 * q
            if hu_rows:                                 # row (b,l), chunk (r, t .. t+7): reversed window at j = l - t
                (b, l), (r, t) = divmod(m, L), divmod(k, T)
            else:                                       # row (r,t), chunk (b, l .. l+7): forward window at j = l - t
                (r, t), (b, l) = divmod(m, T), divmod(k, L)
            assert b < B and r < R
            assert entry == 1 + (b * R + r) * JJ + (l - t) + (T - 1), (ops, tile, kt, row, q)
            n += 1
        assert n == min(ktiles_all - kt0, 9) * 128 * 8


def test_window_staging_matches_the_chunk_major_index():
    """The same entries as toep_index() of the chunk-major path (trow + soff), i.e. the two staging forms feed the MFMAs
    identical operands: bit-identical GEMM results."""
    B, R, T, Lh = 1, 8, 400, 7793
    L, JJ, T8 = Lh + T - 1, Lh + 2 * T - 2, T // 8
    for ops in ('B_HU', 'B_HUT'):
        for kt, row, q, entry in ws_emulate_tile(ops, B, R, T, Lh, 5, 4, 4):
            m = 5 * 128 + row
            kc = kt * 8 + q
            if ops == 'B_HU':
                b, l = divmod(m, L)
                trow = b * R * JJ + l + T - 1
                kq, kr = divmod(kc, T8)
                soff = 1 + kq * JJ - 8 * kr
            else:
                r, t = divmod(m, T)
                trow = r * JJ - t + T - 1
                kq, kr = divmod(kc * 8, L)
                soff = 1 + kq * R * JJ + kr
            assert entry == trow + soff
