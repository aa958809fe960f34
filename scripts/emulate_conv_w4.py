"""Host emulation of the addressing and the pipeline of csrc/conv_fwd_w4.hip (no GPU needed).

The kernel's LDS-DMA pieces, fragment reads, K walk, weight ring and halo double buffer are re-stated here formula by formula
on TAGS instead of data: every 16-byte chunk that a DMA piece writes carries (operand, element offset of its 8 values), every
fragment read checks that the chunk it finds is the one the MFMA operand needs for (row, k) of the implicit GEMM
    C[m, n] = sum_k A[row(m) + tap/chunk offset + k] * B[n * ldb + tap/chunk offset + k].
Timing model: the code between two s_barriers is an "interval"; a DMA piece issued in interval T is in flight until the barrier
behind the first s_waitcnt vmcnt(K) of the issuing wave that covers it (loads retire in order) -- a read of its bytes in any
interval in between is reported as a race, and so is a read that finds a later (or no) tenant.  Run: python scripts/emulate_conv_w4.py
"""
import sys

HROWS, HRUN, HPIECES, BPIECES, BT = 272, 272 * 64, 34, 10, 160 * 64
HALO = 2 * HRUN


class Lds:
    """bytes -> list of (issue interval, visible interval or None, tag) per 16-byte chunk"""

    def __init__(self):
        self.hist = {}

    def write(self, addr, tag, rec):
        assert addr % 16 == 0
        self.hist.setdefault(addr, []).append([rec, tag])

    def read(self, addr, now, what):
        assert addr % 16 == 0, what
        h = self.hist.get(addr)
        assert h, f"read of never-written LDS chunk {addr} ({what})"
        cur = None
        for rec, tag in h:
            t_issue, t_vis = rec["issue"], rec["visible"]
            if t_issue <= now and (t_vis is None or now < t_vis):
                raise AssertionError(f"race: {what} reads chunk {addr} at interval {now} while a piece issued in {t_issue} "
                                     f"(visible {t_vis}) is in flight")
            if t_vis is not None and t_vis <= now:
                cur = tag
        assert cur is not None, what
        return cur


def lin_vw(N, F):
    """RowMap mode 2 (dfold_common.h): rows per window = the F * (N + 4) cells of the frame range in whole 256-row runs"""
    return (F * (N + 4) + 255) // 256 * 256


def lin_rows_check(N, F, W, C=8):
    """The mode-2 row map on the host: every interior cell (w, f, n) is the image of exactly one valid row, its input offset is the
    top-left corner of its 5 x 5 window and its output offset the cell itself; all other rows are invalid (pad columns, the ragged
    end of the last run); rows of one 256-row run are consecutive cells."""
    Wp, Fp, vw = N + 4, F + 4, lin_vw(N, F)
    seen = set()
    for m in range(W * vw):
        w_, v = divmod(m, vw)
        off_in = (w_ * Fp * Wp + v) * C
        off_out = ((2 * Wp + 2) + w_ * Fp * Wp + v) * C
        if m % 256:
            assert off_in == prev + C            # a run is a run of consecutive cells
        prev = off_in
        valid = v < F * Wp and v % Wp < N
        if not valid:
            continue
        f, n = divmod(v, Wp)
        assert off_in == ((w_ * Fp + f) * Wp + n) * C and off_out == ((w_ * Fp + f + 2) * Wp + n + 2) * C
        seen.add((w_, f, n))
    assert len(seen) == W * F * N
    # what a run reads: at most 272 cells + 4 frame rows from its first cell -- behind the last window that is the slack
    # ops.grid_slack() keeps readable
    last_read = (W - 1) * Fp * Wp + (vw - 256) + 4 * Wp + 272
    assert last_read - W * Fp * Wp <= 4 * Wp + 288
    return len(seen)


def run(CI=128, CO=160, N=256, F=2, W=1, m_tile=0, n_tile=0, verbose=False, g_begin=0, g_end=None, lin=False):
    """g_begin / g_end: the K-group range of a stream-K piece (default: the whole tile).  lin: RowMap mode 2 (any N_res)"""
    Wp, Fp = N + 4, F + 4
    ld = CI                      # a_rows.ld
    ldb = 25 * CI
    M = W * lin_vw(N, F) if lin else W * F * N
    a_s0, a_s1, a_s2 = 64, Wp * CI, CI
    b_s0, b_s1, b_s2 = 64, 5 * CI, CI
    nseg = 25 * (CI // 64)
    ngroups = (nseg // 25) * 10
    g_end = ngroups if g_end is None else g_end
    assert 0 <= g_begin < g_end <= ngroups
    m0, n0 = m_tile * 512, n_tile * 160

    def row_off(m):              # rows_in(): top-left corner of the 5x5 window of GEMM row m
        if lin:
            w_, v = divmod(m, lin_vw(N, F))
            return (w_ * Fp * Wp + v) * ld
        wf, n = divmod(m, N)
        w_, f = divmod(wf, F)
        return ((w_ * Fp + f) * Wp + n) * ld

    def grp(g, s0, s1):
        g = min(g, g_end - 1)
        h, t = g & 1, g >> 1
        c, df = divmod(t, 5)
        return c * s0 + df * s1 + h * 32            # elements

    lds = Lds()
    lds0, lds_b = 0, 2 * HALO
    hrow = [row_off(m0), row_off(m0 + 256 if m0 + 256 < M else m0)]
    fifo = {w: [] for w in range(4)}                # per wave: issued pieces, oldest first
    state = {"t": 0}                                # current interval

    def dma(w, src_elem_of_lane, lds_addr, op):
        rec = {"issue": state["t"], "visible": None}
        for lane in range(64):
            lds.write(lds_addr + 16 * lane, (op, src_elem_of_lane(lane)), rec)
        fifo[w].append(rec)

    def lch(lane):
        return (lane & 3) ^ ((lane >> 4) & 3)

    def halo_piece(w, gbase, q, hbuf):
        r = 1 if q >= 17 else 0
        pr = q - 17 * r
        base = gbase + hrow[r] + pr * 16 * ld
        if pr == 16:
            dma(w, lambda l: base + min(l >> 2, 3) * ld + lch(l) * 8, hbuf + q * 1024, "A")
        else:
            dma(w, lambda l: base + (l >> 2) * ld + lch(l) * 8, hbuf + q * 1024, "A")

    def b_piece(w, tbase, pp, stage):
        base = tbase + n0 * ldb + pp * 16 * ldb
        dma(w, lambda l: base + (l >> 2) * ldb + lch(l) * 8, stage + pp * 1024, "B")

    def wait_vmcnt(w, k):
        n = len(fifo[w]) - k
        for rec in fifo[w][:max(n, 0)]:
            rec["landed_for"] = True
        done = fifo[w][:max(n, 0)]
        fifo[w] = fifo[w][max(n, 0):]
        return done

    pending = []                                    # pieces waited for by their wave, visible after the next barrier

    def barrier():
        state["t"] += 1
        for rec in pending:
            rec["visible"] = state["t"]
        pending.clear()

    def a_lane(w, lane, dn, kb):
        frow, fhalf = lane & 31, lane >> 5
        row = (w & 1) * 128 + frow + dn
        return (w >> 1) * HRUN + row * 64 + (((kb * 2 + fhalf) ^ ((row >> 2) & 3)) << 4)

    def bf_lane(lane, kb):
        frow, fhalf = lane & 31, lane >> 5
        return frow * 64 + (((kb * 2 + fhalf) ^ ((frow >> 2) & 3)) << 4)

    nchecked = [0]

    def frag_reads(w, hbuf, stage, dn_addr, kb, tile_u):
        """the nine reads of K16 block kb of tile u (its group / residue tap give the expected operands)"""
        g, dn = divmod(tile_u, 5)
        g += g_begin
        if g >= g_end:
            return                                   # reads behind the last tile: never consumed
        assert dn == dn_addr
        ka = grp(g, a_s0, a_s1) + dn * a_s2
        kbo = grp(g, b_s0, b_s1) + dn * b_s2
        for lane in range(64):
            frow, fhalf = lane & 31, lane >> 5
            k = (kb * 2 + fhalf) * 8
            for i in range(4):
                m = m0 + w * 128 + i * 32 + frow
                mm = m if m < M else m - 256         # a run past M repeats run 0 (never stored)
                want = ("A", row_off(mm) + ka + k)
                got = lds.read(hbuf + a_lane(w, lane, dn, kb) + i * 2048, state["t"], f"A frag w{w} u{tile_u} kb{kb} i{i} lane{lane}")
                assert got == want, (got, want, w, tile_u, kb, i, lane)
            for j in range(5):
                n = n0 + j * 32 + frow
                want = ("B", n * ldb + kbo + k)
                got = lds.read(stage + bf_lane(lane, kb) + j * 2048, state["t"], f"B frag w{w} u{tile_u} kb{kb} j{j} lane{lane}")
                assert got == want, (got, want, w, tile_u, kb, j, lane)
            nchecked[0] += 9

    # ---------------- prologue ----------------
    pa_n = {w: grp(g_begin, a_s0, a_s1) for w in range(4)}
    pb_c, pb_n = grp(g_begin, b_s0, b_s1), grp(g_begin + 1, b_s0, b_s1)
    for w in range(4):
        for t in range(9):
            q = w + 4 * t
            if q < HPIECES:
                halo_piece(w, pa_n[w], q, lds0)
        for v in range(3):
            for t in range(3):
                pp = w + 4 * t
                if pp < BPIECES:
                    b_piece(w, pb_c + v * b_s2, pp, lds_b + v * BT)
    pa_next = grp(g_begin + 1, a_s0, a_s1)
    hb_c, hb_n = lds0, lds0 + HALO
    st = [lds_b, lds_b + BT, lds_b + 2 * BT]          # stages of tiles u, u+1, u+2
    for w in range(4):
        pending.extend(wait_vmcnt(w, 0))
    barrier()
    for w in range(4):
        frag_reads(w, hb_c, st[0], 0, 0, 0)

    HB0 = [0, 9, 18, 26, 34]
    u = 0
    for g in range(g_begin, g_end):
        for DN in range(5):
            # first half: reads of block 1 of tile u, halo pieces of group g + 1
            for w in range(4):
                frag_reads(w, hb_c, st[0], DN, 1, u)
                if DN < 4:
                    for t in range(2):
                        halo_piece(w, pa_next, HB0[DN] + w + 4 * t, hb_n)
                if DN < 2 and w == 0:
                    halo_piece(w, pa_next, HB0[DN] + 8, hb_n)
            for w in range(4):
                pending.extend(wait_vmcnt(w, 4 if DN < 4 else 2))
            barrier()
            # second half: reads of block 0 of tile u + 1, weight tile u + 3
            DN1 = 0 if DN == 4 else DN + 1
            for w in range(4):
                frag_reads(w, hb_n if DN == 4 else hb_c, st[1], DN1, 0, u + 1)
                tb = pb_c + (DN + 3) * b_s2 if DN + 3 < 5 else pb_n + (DN - 2) * b_s2
                rot = (w + DN) & 3
                for t in range(2):
                    b_piece(w, tb, rot + 4 * t, st[0])
                if rot < 2:
                    b_piece(w, tb, rot + 8, st[0])
            st = [st[1], st[2], st[0]]
            u += 1
        hb_c, hb_n = hb_n, hb_c
        pb_c, pb_n = pb_n, grp(g + 2, b_s0, b_s1)
        pa_next = grp(g + 2, a_s0, a_s1)
    if verbose:
        print(f"conv_w4 emulation: CI {CI}, N {N}, F {F}, tile ({m_tile}, {n_tile}): {u} steps, {nchecked[0]} fragment reads checked")
    return nchecked[0]


def streamk_plan(tiles, ngroups, n_wg):
    """The stream-K work split of dfold_conv_w4_kernel<true>, replayed: workgroup s takes units [s per, (s + 1) per) of the
    tile-major (tile, group) sequence.  Returns (per, pieces) with pieces[s] = [(tile, g_begin, g_end, slot or None)], and checks
    what the kernel relies on: every (tile, group) covered exactly once; a piece is parked iff it does not cover its whole tile;
    no slot is written twice; the reducer's formulas (s_lo, s_hi, slot of workgroup sw's piece of the tile) name exactly the
    parked pieces of that tile, in workgroup order; at most two parked pieces per workgroup."""
    units = tiles * ngroups
    per = -(-units // n_wg)
    pieces, parked, covered = {}, {}, {}
    for s in range(n_wg):
        u, u_end = s * per, min(units, s * per + per)
        first = True
        pieces[s] = []
        while u < u_end:
            tile = u // ngroups
            gb = u - tile * ngroups
            ge = min(ngroups, gb + (u_end - u))
            slot = None
            if gb != 0 or ge != ngroups:
                slot = 2 * s + (0 if first else 1)
                assert slot not in parked, (s, slot)
                parked[slot] = (tile, s, gb, ge)
            for g in range(gb, ge):
                assert (tile, g) not in covered
                covered[(tile, g)] = s
            pieces[s].append((tile, gb, ge, slot))
            u += ge - gb
            first = False
        assert sum(1 for pc in pieces[s] if pc[3] is not None) <= 2
    assert len(covered) == units
    for tile in range(tiles):
        mine = sorted((v[1], k, v[2], v[3]) for k, v in parked.items() if v[0] == tile)       # (workgroup, slot, gb, ge)
        s_lo, s_hi = (tile * ngroups) // per, ((tile + 1) * ngroups - 1) // per
        if s_lo == s_hi:
            assert not mine, tile                       # one workgroup covers the tile: straight to the epilogue
            continue
        want = [(sw, 2 * sw + (0 if (sw * per) // ngroups == tile else 1)) for sw in range(s_lo, s_hi + 1)]
        assert [(a, b) for a, b, _, _ in mine] == want, (tile, mine, want)
        assert mine[0][2] == 0 and mine[-1][3] == ngroups and all(x[3] == y[2] for x, y in zip(mine, mine[1:])), (tile, mine)
    return per, pieces


if __name__ == "__main__":
    run(verbose=True)
    run(CI=192, F=3, m_tile=1, n_tile=0, verbose=True)       # odd number of 256-row runs: the last tile's second run repeats run 0
    run(CI=64, N=512, F=1, W=2, m_tile=1, verbose=True)       # one frame row = one 512-row tile
    print("ok")
