"""Host emulation of the data movement of csrc/tn_gemm.hip (LDS-DMA image, XOR keys, ds_read_b64_tr_b16 gather, MFMA operand /
accumulator layouts, both epilogues' addressing) for one 256 x 256 tile and one K step, against A^T B.  Index arithmetic
only; formulas transcribed from the kernel:   python scripts/emulate_tn_gemm.py"""
import numpy as np

from emulate_wgrad_tn import tr_read

GBK, GBT, GNJ = 64, 256, 4
G_PITCH = GBT * 2
G_TILE = GBK * G_PITCH


def run(lda=320, ldb=512, seed=0, M=None, N=None):
    """M, N (multiples of 8, <= 256): ragged output extents -- the tile hangs over the edge of the output: columns past M (N) are
    fetched from column chunk 0 instead and neither epilogue stores there."""
    rng = np.random.default_rng(seed)
    A = rng.integers(-3, 4, size=(GBK, lda)).astype(np.int64)
    B = rng.integers(-3, 4, size=(GBK, ldb)).astype(np.int64)
    m0, n0 = 0, 256 if ldb >= 512 else 0
    ragged = M is not None
    if ragged:
        n0 = 0
        assert M <= lda and N <= ldb and M % 8 == 0 and N % 8 == 0
        # a poisoned element right behind each operand: a fetch past the last valid chunk of the LAST row would pick it up
        A = np.concatenate([A.reshape(-1), np.full(4096, 10 ** 9, dtype=np.int64)])[:GBK * lda + 4096]
        B = np.concatenate([B.reshape(-1), np.full(4096, 10 ** 9, dtype=np.int64)])[:GBK * ldb + 4096]
        ref = np.zeros((GBT, GBT), dtype=np.int64)
        ref[:M, :N] = A[:GBK * lda].reshape(GBK, lda)[:, :M].T @ B[:GBK * ldb].reshape(GBK, ldb)[:, :N]
    else:
        M, N = m0 + GBT, n0 + GBT
        ref = A[:, m0:m0 + GBT].T @ B[:, n0:n0 + GBT]
    Ab, Bb = A.reshape(-1), B.reshape(-1)
    pitchA, pitchB = lda * 2, ldb * 2
    pa, pb = m0 * 2, n0 * 2
    lane = np.arange(64)
    lds = np.full(2 * G_TILE // 2, 10 ** 6, dtype=np.int64)
    for w in range(8):
        row = 2 * w + (lane >> 5)
        lc = (lane & 31) ^ ((row & 3) << 2)
        aoff0 = row * pitchA + np.where(m0 + lc * 8 < M, lc, 0) * 16
        boff0 = row * pitchB + np.where(n0 + lc * 8 < N, lc, 0) * 16
        for t in range(4):
            for (src0, base, flat) in ((pa + t * 16 * pitchA + aoff0, 0, Ab), (pb + t * 16 * pitchB + boff0, G_TILE, Bb)):
                dst = base + (t * 8 + w) * 1024 + lane * 16
                for l in range(64):
                    lds[dst[l] // 2:dst[l] // 2 + 8] = flat[src0[l] // 2:src0[l] // 2 + 8]
    out = np.zeros((GBT, GBT), dtype=np.int64)
    stored = np.zeros((GBT, GBT), dtype=np.int64)
    for w in range(8):
        wm, wn = w >> 1, w & 1
        p16, g = lane & 15, lane >> 4
        row_l = (g >> 1) * 8 + (p16 >> 2)
        c0 = (g & 1) * 2 + ((p16 >> 1) & 1)
        key = (p16 >> 2) << 2
        fa0 = row_l * G_PITCH + (((wm * 4 + c0) ^ key) << 4) + (p16 & 1) * 8
        fb00 = row_l * G_PITCH + (((wn * 4 + c0) ^ key) << 4) + (p16 & 1) * 8
        fb10 = row_l * G_PITCH + (((wn * 4 + 8 + c0) ^ key) << 4) + (p16 & 1) * 8
        acc = np.zeros((2, GNJ, 64, 16), dtype=np.int64)
        for kb in range(4):
            def frag(base, off):
                return np.concatenate([tr_read(lds, base + off), tr_read(lds, base + off + 4 * G_PITCH)], 1)
            af = [frag(fa0, kb * 16 * G_PITCH), frag(fa0, kb * 16 * G_PITCH + 256)]
            bf = [frag(fb00, G_TILE + kb * 16 * G_PITCH), frag(fb10, G_TILE + kb * 16 * G_PITCH),
                  frag(fb00, G_TILE + kb * 16 * G_PITCH + 256), frag(fb10, G_TILE + kb * 16 * G_PITCH + 256)]
            for i in range(2):
                for j in range(GNJ):
                    Am = np.zeros((32, 16), dtype=np.int64)
                    Bm = np.zeros((32, 16), dtype=np.int64)
                    for l in range(64):
                        Am[l & 31, (l >> 5) * 8:(l >> 5) * 8 + 8] = af[i][l]
                        Bm[l & 31, (l >> 5) * 8:(l >> 5) * 8 + 8] = bf[j][l]
                    D = Am @ Bm.T
                    for l in range(64):
                        for e in range(16):
                            acc[i, j, l, e] += D[(e & 3) + 8 * (e >> 2) + 4 * (l >> 5), l & 31]
        # fp32 epilogue
        for i in range(2):
            for j in range(GNJ):
                for l in range(64):
                    frow, fhalf = l & 31, l >> 5
                    for e in range(16):
                        m = (wm + 4 * i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf
                        if m0 + m < M and n0 + wn * 32 + frow + j * 64 < N:
                            out[m, wn * 32 + frow + j * 64] += acc[i, j, l, e]
        # bf16 epilogue through the staging rows (272-byte pitch, 16-byte chunks)
        for i in range(2):
            stg = np.zeros(32 * 136, dtype=np.int64)
            for l in range(64):
                frow, fhalf = l & 31, l >> 5
                for e in range(16):
                    r = (e & 3) + 8 * (e >> 2) + 4 * fhalf
                    for j in range(GNJ):
                        stg[(r * 272 + (j * 32 + frow) * 2) // 2] = acc[i, j, l, e]
            for k in range(8):
                for l in range(64):
                    c = l + 64 * k
                    r, ch = c >> 4, c & 15
                    j, qq = ch >> 2, ch & 3
                    v = stg[(r * 272 + ch * 16) // 2:(r * 272 + ch * 16) // 2 + 8]
                    m = (wm + 4 * i) * 32 + r
                    n = (wn + 2 * j) * 32 + qq * 8
                    if m0 + m < M and n0 + n < N:
                        stored[m, n:n + 8] += v
    return int(np.abs(out - ref).max()), int(np.abs(stored - ref).max())


if __name__ == "__main__":
    for kw in (dict(), dict(lda=256, ldb=256, seed=1), dict(lda=128, ldb=136, seed=2, M=128, N=136), dict(lda=8, ldb=328, seed=3, M=8, N=256)):
        e = run(**kw)
        print(kw, "max |diff| fp32 / bf16 epilogue:", e)
        assert e == (0, 0)
    print("ok")
