"""Lane-level emulation of k_slam's sweep16_loop (one tile row per wave, ONE barrier per block step, operand-image panels,
transposed products, MFMA block-pivot inversion): checks the index algebra and the absence of order dependence between
the waves inside a barrier interval (waves are run in a random order per interval; results must not depend on it).
Compares with the dense solution / inverse from numpy."""
import sys
import numpy as np
from inv16_blk_emul import mfma, inv16_blk, LR, LC

LANE = np.arange(64)


def mfma4(a, b, c):
    for s in range(4):
        c = mfma(a[s], b[s], c)
    return c


def ld_op(tile):  # tile: 256 doubles, image [h][lane][2] -> o[s][lane]
    o = np.zeros((4, 64))
    for s in range(4):
        o[s] = tile[(s >> 1) * 128 + 2 * LANE + (s & 1)]
    return o


def st_op(tile, v):
    for s in range(4):
        tile[(s >> 1) * 128 + 2 * LANE + (s & 1)] = v[s]


def acc_addr(r):  # address in an operand image of the element (row = lr + 4 r, col = lc) held in accumulator layout
    row = LR + 4 * r
    lanep = 16 * (LC & 3) + row
    s = LC >> 2
    return (s >> 1) * 128 + 2 * lanep + (s & 1)


def sweep(Afull, np_, FT, rng):
    N = Afull.shape[0]
    Tn = N // 16
    nK = (np_ + 15) // 16
    # wave -> tile row (as sweep_packed_fast)
    trow = []
    for wv in range(8):
        t = wv if wv < FT // 2 else (FT - 1) - (wv - FT // 2)
        if Tn == FT:
            t = 0 if wv == 0 else 1 if wv == FT // 2 else wv + 1 if wv < FT // 2 else FT + FT // 2 - wv
        trow.append(t)
    e_row = FT - 1 if Tn < FT else 0
    acc = {}
    for wv in range(8):
        I = trow[wv]
        a = np.zeros((FT, 4, 64))
        for u in range(FT):
            for r in range(4):
                i = 16 * I + LR + 4 * r
                j = 16 * u + LC
                ok = (i < N) & (j < N)
                ii = np.minimum(np.maximum(i, j), N - 1)
                jj = np.minimum(np.minimum(i, j), N - 1)
                a[u][r] = np.where(ok, Afull[ii, jj], 0.0)
        acc[wv] = a
    pan = np.full((2, FT, 256), np.nan)
    wt = np.full((2, FT, 256), np.nan)
    einv = np.full((2, 256), np.nan)
    dscr = np.full((2, 256), np.nan)
    wave_of = {trow[w]: w for w in range(8)}
    # prologue: E_0
    st_op(dscr[0], acc[wave_of[0]][0])
    d = ld_op(dscr[0])
    d = inv16_blk_masked(d, min(16, np_))
    st_op(einv[0], d)
    for K in range(nK):
        kb = 16 * K
        b = K & 1
        has_mask = np_ < kb + 16
        have_next = kb + 16 < np_
        # ---- P(K) ----
        for wv in rng.permutation(8):
            I = trow[wv]
            if I >= Tn:
                continue
            a = acc[wv]
            if I >= K:
                colact = kb + LC < np_
                for r in range(4):
                    pan[b][I][acc_addr(r)] = np.where(colact, a[K][r], 0.0)
            if I == K:
                for u in range(K):
                    v = np.array([np.where(kb + LR + 4 * r < np_, a[u][r], 0.0) for r in range(4)])
                    st_op(pan[b][u], v)
            if have_next and I == K + 1:
                st_op(dscr[(K + 1) & 1], a[K + 1])
        # ---- barrier B1(K) ----
        for wv in rng.permutation(8):
            I = trow[wv]
            a = acc[wv]
            live = I < Tn
            if live and K >= 1 and I == K - 1:  # deferred pivot-row replacement of step K - 1 (all rows active)
                for u in range(K - 1):
                    t = ld_op(wt[(K - 1) & 1][u])
                    a[u] = -t
            if I == e_row and have_next:
                aP1 = ld_op(pan[b][K + 1])
                eB = ld_op(einv[b])
                w1T = mfma4(eB, aP1, np.zeros((4, 64)))
                dn = ld_op(dscr[(K + 1) & 1])
                dn = mfma4(w1T, aP1, dn)
                dn = inv16_blk_masked(dn, min(16, np_ - kb - 16))
                st_op(einv[(K + 1) & 1], dn)
            if not live:
                continue
            aP = ld_op(pan[b][I])
            eB = ld_op(einv[b])
            wT = mfma4(eB, aP, np.zeros((4, 64)))
            st_op(wt[b][I], wT)
            if I != K or has_mask:
                for u in range(FT):
                    if u <= I and (u != K or I == K):
                        bP = ld_op(pan[b][u])
                        a[u] = mfma4(wT, bP, a[u])
            w = np.array([wt[b][I][acc_addr(r)] for r in range(4)])  # own W in accumulator layout
            if I > K:
                a[K] = -w
            if I == K:
                colact = kb + LC < np_
                for r in range(4):
                    rowact = kb + LR + 4 * r < np_
                    a[K][r] = np.where(rowact, np.where(colact, eB[r], -wT[r]), np.where(colact, -w[r], a[K][r]))
    # final barrier, deferred replacement of the last pivot row
    K = nK - 1
    wv = wave_of[K]
    for u in range(K):
        t = ld_op(wt[K & 1][u])
        for r in range(4):
            rowact = 16 * K + LR + 4 * r < np_
            acc[wv][u][r] = np.where(rowact, -t[r], acc[wv][u][r])
    out = np.zeros((N, N))
    for wv in range(8):
        I = trow[wv]
        if I >= Tn:
            continue
        for u in range(I + 1):
            for r in range(4):
                i = 16 * I + LR + 4 * r
                j = 16 * u + LC
                out[i, j] = acc[wv][u][r]
    return out


def inv16_blk_masked(d, nact):
    d = d.copy()
    if nact < 16:
        for r in range(4):
            row = LR + 4 * r
            m = (row >= nact) | (LC >= nact)
            d[r] = np.where(m, np.where(row == LC, 1.0, 0.0), d[r])
    return inv16_blk(d)


def run(P, FT=8, seed=0):
    rng = np.random.default_rng(seed)
    np_ = 3 * P
    na = np_ + 1
    Tn = (na + 15) // 16
    N = 16 * Tn
    B = rng.normal(size=(np_, np_))
    S = B @ B.T + np_ * np.eye(np_)
    rhs = rng.normal(size=np_)
    A = np.zeros((N, N))
    A[:np_, :np_] = S
    A[np_, :np_] = rhs
    A[:np_, np_] = rhs
    outs = [sweep(A, np_, FT, np.random.default_rng(s)) for s in (1, 2, 3)]
    assert all(np.array_equal(outs[0], o, equal_nan=True) for o in outs[1:]), "wave-order dependence"
    out = outs[0]
    low = np.tril(out[:np_, :np_])
    full = low + np.tril(low, -1).T
    ref = -np.linalg.inv(S)
    sol = np.linalg.solve(S, rhs)
    e1 = np.abs(full - ref).max() / np.abs(ref).max()
    e2 = np.abs(out[np_, :np_] - sol).max() / np.abs(sol).max()
    print("P=%d np=%d Tn=%d: inverse rel err %.2e, solution rel err %.2e" % (P, np_, Tn, e1, e2))
    assert e1 < 1e-10 and e2 < 1e-10


if __name__ == "__main__":
    for P in (37, 42, 36, 21, 5, 16, 2, 1, 32):
        run(P)
