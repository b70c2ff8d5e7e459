"""Index arithmetic of the InfoNCE tile kernels (csrc/infonce.hip: TileImg, fwd_tiles_lds_kernel, bwd_tiles_lds_kernel)
restated in numpy and checked against plain matrix products - runs anywhere (python tools/tile_image_emulation.py):
  * the LDS-DMA staging (1 KB pieces, lane-linear destination, XOR swizzle on the SOURCE address) leaves every element of
    the 32 x D tile where `frag` and `at` look for it;
  * fragment reads (ds_read_b128, lane = row) touch 16 distinct 16-byte slots in each of the instruction's four 16-lane
    groups, second-stage operand reads (ds_read_b32, one row per half-wave) 32 distinct banks: no bank conflicts
    (measured: SQ_LDS_BANK_CONFLICT = 0, profiles/r03_infonce_pmc.txt);
  * a similarity tile computed TRANSPOSED (A operand = tile s, B operand = tile t) puts entry [i = lane & 31][j(r, h)] in
    accumulator register r, j(r, h) = (r & 3) + 8 (r >> 2) + 4 h, and feeding register r straight back as the A operand of
    second-stage step r - with rows j(r, 0), j(r, 1) of tile s as that step's B operand - yields C . n_s in accumulator
    layout. MFMA 32x32x2 model: A[lane & 31][lane >> 5], B[lane >> 5][lane & 31], D[j(r, h)][lane & 31] in register r."""
import numpy as np

T = 32


def jmap(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma(a_op, b_op, acc):
    A, B = np.zeros((32, 2)), np.zeros((2, 32))
    for lane in range(64):
        A[lane % 32][lane // 32] = a_op[lane]
        B[lane // 32][lane % 32] = b_op[lane]
    D = A @ B
    for lane in range(64):
        for r in range(16):
            acc[lane][r] += D[jmap(r, lane // 32)][lane % 32]


def check(D):
    SL, RB, RPP = D // 4, 64 // D, 256 // D
    key = lambda row: (row // RB) & (SL - 1)          # noqa: E731
    rng = np.random.default_rng(0)

    def stage(tile):
        img, flat = np.full(T * D, np.nan), tile.reshape(-1)
        for p in range(T // RPP):
            for lane in range(64):
                row = p * RPP + lane // SL
                src = row * D + ((lane % SL) ^ key(row)) * 4
                img[p * 256 + lane * 4:p * 256 + lane * 4 + 4] = flat[src:src + 4]
        assert not np.isnan(img).any()
        return img

    def frag(img, lr, h):
        f = np.zeros(D // 2)
        for q in range(D // 8):
            off = lr * D + (((h * (SL // 2) + q) ^ key(lr)) << 2)
            f[4 * q:4 * q + 4] = img[off:off + 4]
        return f

    def at(img, row, col):
        return img[row * D + (((col >> 2) ^ key(row)) << 2) + (col & 3)]

    n1t, n2s = rng.standard_normal((T, D)), rng.standard_normal((T, D))
    It, Is = stage(n1t), stage(n2s)
    for lane in range(64):
        assert np.array_equal(frag(It, lane % 32, lane // 32), n1t[lane % 32, (lane // 32) * D // 2:(lane // 32 + 1) * D // 2])
    assert all(at(Is, r, c) == n2s[r, c] for r in range(T) for c in range(D))
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[x + 32 for x in g] for g in groups]                        # ds_read_b128 lane groups (MI355X_MICROARCH.md)
    for q in range(D // 8):
        for g in groups:
            slots = {((lane % 32) * D * 4 + (((lane // 32 * (SL // 2) + q) ^ key(lane % 32)) << 4)) // 16 % 16 for lane in g}
            assert len(slots) == 16, (D, q, len(slots))
    for row in range(T):
        for f in range(D // 32):
            banks = {(row * D + ((((32 * f + lr) >> 2) ^ key(row)) << 2) + ((32 * f + lr) & 3)) % 32 for lr in range(32)}
            assert len(banks) == 32
    acc = np.zeros((64, 16))
    fs = [frag(Is, lane % 32, lane // 32) for lane in range(64)]
    ft = [frag(It, lane % 32, lane // 32) for lane in range(64)]
    for k in range(D // 2):
        mfma([fs[lane][k] for lane in range(64)], [ft[lane][k] for lane in range(64)], acc)
    S = n1t @ n2s.T
    assert all(np.isclose(acc[lane][r], S[lane % 32][jmap(r, lane // 32)]) for lane in range(64) for r in range(16))
    g = [np.zeros((64, 16)) for _ in range(D // 32)]
    for r in range(16):
        for f in range(D // 32):
            mfma([acc[lane][r] for lane in range(64)], [at(Is, jmap(r, lane // 32), 32 * f + lane % 32) for lane in range(64)], g[f])
    G = S @ n2s
    assert all(np.isclose(g[f][lane][r], G[jmap(r, lane // 32)][32 * f + lane % 32])
               for f in range(D // 32) for lane in range(64) for r in range(16))
    print("D = %d: staging, fragments, bank conflicts, transposed-tile chaining: ok" % D)


if __name__ == "__main__":
    check(64)
    check(32)
