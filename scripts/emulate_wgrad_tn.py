"""Host emulation of the data movement of csrc/conv_wgrad_tn.hip (LDS-DMA image, XOR keys, ds_read_b64_tr_b16 gather,
v_mfma_f32_32x32x16_bf16 operand / accumulator layouts, epilogue addressing) on small integer grids, checked against the
plain definition of the 5x5 conv weight gradient.  Index arithmetic only -- the formulas below are transcribed from the
kernel; run it after touching the kernel's addressing:   python scripts/emulate_wgrad_tn.py"""
import numpy as np

TBK, TBM, TBC, TNJ = 64, 256, 64, 5
TA_PITCH, TB_PITCH = TBM * 2, TBC * 2
TA_BYTES, TB_ROWS, TB_BYTES = TBK * TA_PITCH, 68, 9 * 1024


def tr_read(lds16, addr):
    """ds_read_b64_tr_b16: lane q supplies a byte address of 4 contiguous b16; within each 16-lane group
    out[p][j] = in[4 j + (p >> 2)][p & 3]"""
    out = np.zeros((64, 4), dtype=lds16.dtype)
    for l in range(64):
        g, p = l >> 4, l & 15
        for j in range(4):
            q = g * 16 + 4 * j + (p >> 2)
            assert addr[q] % 8 == 0
            out[l, j] = lds16[addr[q] // 2 + (p & 3)]
    return out


def run(CA=512, CB=128, Wn=2, F=3, N=64, flip=0, f0=0, nf=None, seed=0, check_wgs=6):
    rng = np.random.default_rng(seed)
    Fp, Wp = F + 4, N + 4
    nf = F - f0 if nf is None else nf
    A = np.zeros((Wn, Fp, Wp, CA), dtype=np.int64)
    B = np.zeros((Wn, Fp, Wp, CB), dtype=np.int64)
    A[:, 2:-2, 2:-2] = rng.integers(-3, 4, size=(Wn, F, N, CA))
    B[:, 2:-2, 2:-2] = rng.integers(-3, 4, size=(Wn, F, N, CB))
    # reference: out[a][tap][b] = sum_cells A[w, 2+f, 2+n, a] * B[w, f+z0, n+z1, b], tap = z0*5+z1 (flipped: 24 - ..)
    ref = np.zeros((CA, 25, CB), dtype=np.int64)
    Ac = A[:, 2 + f0:2 + f0 + nf, 2:2 + N].reshape(-1, CA)
    for z0 in range(5):
        for z1 in range(5):
            Bs = B[:, f0 + z0:f0 + z0 + nf, z1:z1 + N].reshape(-1, CB)
            tap = 24 - (z0 * 5 + z1) if flip else z0 * 5 + z1
            ref[:, tap, :] = Ac.T @ Bs
    Ab, Bb = A.reshape(-1), B.reshape(-1)          # element arrays (2 bytes per element)
    pitchA, pitchB = CA * 2, CB * 2
    rowA, rowB = Wp * pitchA, Wp * pitchB
    winA, winB = Fp * rowA, Fp * rowB
    pA = (((2 + f0) * Wp + 2) * CA) * 2            # byte offsets of p.A / p.B inside the grids
    pB = (f0 * Wp * CB) * 2
    nchunk, nF, nW = N // TBK, nf, Wn
    out = np.zeros((CA, 25, CB), dtype=np.int64)
    tiles_m = CA // TBM
    nwg = tiles_m * 5 * (CB // TBC)
    wgs = rng.choice(nwg, size=min(check_wgs, nwg), replace=False)
    done = []
    for lid in wgs:
        r = int(lid)
        m0 = (r % tiles_m) * TBM
        r //= tiles_m
        z0 = r % 5
        n0 = (r // 5) * TBC
        pa = pA + m0 * 2
        pb = pB + z0 * rowB + n0 * 2
        acc = np.zeros((8, 2, TNJ, 64, 16), dtype=np.int64)     # [wave][i][j][lane][e]
        lane = np.arange(64)
        for wi in range(nW):
            for fi in range(nF):
                for ci in range(nchunk):
                    sa = pa + wi * winA + fi * rowA + ci * TBK * pitchA
                    sb = pb + wi * winB + fi * rowB + ci * TBK * pitchB
                    lds = np.full((TA_BYTES + TB_BYTES) // 2, 10 ** 6, dtype=np.int64)   # poison: unwritten LDS must not be read
                    for w in range(8):                         # the DMA image
                        cell = 2 * w + (lane >> 5)
                        lc = (lane & 31) ^ ((cell & 3) << 2)
                        aoff0 = cell * pitchA + lc * 16
                        cb = 8 * w + (lane >> 3)
                        boff0 = cb * pitchB + (((lane & 7) ^ (((cb >> 1) & 1) << 2)) << 4)
                        ch = 64 + (lane >> 3)
                        chs = np.minimum(ch, TB_ROWS - 1)
                        boff8 = chs * pitchB + (((lane & 7) ^ (((ch >> 1) & 1) << 2)) << 4)
                        for t in range(4):
                            src = sa + t * 16 * pitchA + aoff0
                            dst = (t * 8 + w) * 1024 + lane * 16
                            for l in range(64):
                                lds[dst[l] // 2:dst[l] // 2 + 8] = Ab[src[l] // 2:src[l] // 2 + 8]
                        for (off, piece) in ((boff0, w), (boff8, 8)):
                            src = sb + off
                            dst = TA_BYTES + piece * 1024 + lane * 16
                            for l in range(64):
                                lds[dst[l] // 2:dst[l] // 2 + 8] = Bb[src[l] // 2:src[l] // 2 + 8]
                    for w in range(8):
                        wm, wn = w >> 1, w & 1
                        p16, g = lane & 15, lane >> 4
                        cell_l = (g >> 1) * 8 + (p16 >> 2)
                        c0 = (g & 1) * 2 + ((p16 >> 1) & 1)
                        fa = cell_l * TA_PITCH + (((wm * 4 + c0) ^ ((p16 >> 2) << 2)) << 4) + (p16 & 1) * 8
                        fb = []
                        for j in range(TNJ):
                            cell = cell_l + j
                            fb.append(TA_BYTES + cell * TB_PITCH + (((wn * 4 + c0) ^ (((cell >> 1) & 1) << 2)) << 4) + (p16 & 1) * 8)
                        for kb in range(4):
                            af, bf = [], []
                            for i in range(2):
                                a = fa + i * 256 + kb * 16 * TA_PITCH
                                af.append(np.concatenate([tr_read(lds, a), tr_read(lds, a + 4 * TA_PITCH)], 1))
                            for j in range(TNJ):
                                a = fb[j] + kb * 16 * TB_PITCH
                                bf.append(np.concatenate([tr_read(lds, a), tr_read(lds, a + 4 * TB_PITCH)], 1))
                            for i in range(2):
                                for j in range(TNJ):
                                    # v_mfma_f32_32x32x16: A lane l = row l & 31, k = (l >> 5) * 8 + e; B likewise (column);
                                    # D lane l, element e: column l & 31, row (e & 3) + 8 (e >> 2) + 4 (l >> 5)
                                    Am = np.zeros((32, 16), dtype=np.int64)
                                    Bm = np.zeros((32, 16), dtype=np.int64)
                                    for l in range(64):
                                        Am[l & 31, (l >> 5) * 8:(l >> 5) * 8 + 8] = af[i][l]
                                        Bm[l & 31, (l >> 5) * 8:(l >> 5) * 8 + 8] = bf[j][l]
                                    D = Am @ Bm.T
                                    for l in range(64):
                                        for e in range(16):
                                            acc[w, i, j, l, e] += D[(e & 3) + 8 * (e >> 2) + 4 * (l >> 5), l & 31]
        for w in range(8):
            wm, wn = w >> 1, w & 1
            for i in range(2):
                for j in range(TNJ):
                    tap = 24 - (z0 * 5 + j) if flip else z0 * 5 + j
                    for l in range(64):
                        frow, fhalf = l & 31, l >> 5
                        for e in range(16):
                            m = m0 + wm * 32 + i * 128 + (e & 3) + 8 * (e >> 2) + 4 * fhalf
                            n = n0 + wn * 32 + frow
                            out[m, tap, n] += acc[w, i, j, l, e]
        done.append((m0, z0, n0))
    bad = 0
    for (m0, z0, n0) in done:
        taps = [24 - (z0 * 5 + j) if flip else z0 * 5 + j for j in range(5)]
        bad += int(np.abs(out[m0:m0 + TBM, taps, n0:n0 + TBC] - ref[m0:m0 + TBM, taps, n0:n0 + TBC]).max() != 0)
    return bad, len(done)


if __name__ == "__main__":
    for kw in (dict(), dict(flip=1, seed=1), dict(N=128, F=2, Wn=1, seed=2, check_wgs=3), dict(F=4, f0=1, nf=2, seed=3, check_wgs=3)):
        bad, n = run(**kw)
        print(kw, "workgroups wrong:", bad, "of", n)
        assert bad == 0
    print("ok")
