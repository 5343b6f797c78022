"""Lane-level emulation (numpy) of the MFMA block-pivot 16 x 16 inversion used by k_slam's sweep (inv16_blk):
checks the index algebra of the HIP code against numpy.linalg.inv.  Conventions = those of k_slam.hip:
lane = 16 lr + lc; A operand lane -> A[i = lc][k = lr]; B operand lane -> B[k = lr][j = lc];
accumulator reg r, lane -> C[lr + 4 r][lc]."""
import numpy as np

LR = np.arange(64) >> 4
LC = np.arange(64) & 15


def mfma(a, b, c):
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    for l in range(64):
        A[LC[l], LR[l]] = a[l]
        B[LR[l], LC[l]] = b[l]
    C = np.zeros((16, 16))
    for r in range(4):
        for l in range(64):
            C[LR[l] + 4 * r, LC[l]] = c[r][l]
    C = C + A @ B
    out = np.zeros((4, 64))
    for r in range(4):
        for l in range(64):
            out[r][l] = C[LR[l] + 4 * r, LC[l]]
    return out


def to_acc(D):
    d = np.zeros((4, 64))
    for r in range(4):
        for l in range(64):
            d[r][l] = D[LR[l] + 4 * r, LC[l]]
    return d


def from_acc(d):
    D = np.zeros((16, 16))
    for r in range(4):
        for l in range(64):
            D[LR[l] + 4 * r, LC[l]] = d[r][l]
    return D


def e4_closed(p):
    """-inv of the SPD 4x4 p (lower entries used), 2x2 block form (pivot_inverse_from)"""
    a00, a10, a11 = p[0, 0], p[1, 0], p[1, 1]
    a20, a21, a22 = p[2, 0], p[2, 1], p[2, 2]
    a30, a31, a32, a33 = p[3, 0], p[3, 1], p[3, 2], p[3, 3]
    ip = 1.0 / (a00 * a11 - a10 * a10)
    p00, p10, p11 = a11 * ip, -a10 * ip, a00 * ip
    t00, t01 = a20 * p00 + a21 * p10, a20 * p10 + a21 * p11
    t10, t11 = a30 * p00 + a31 * p10, a30 * p10 + a31 * p11
    s00 = a22 - (t00 * a20 + t01 * a21)
    s10 = a32 - (t10 * a20 + t11 * a21)
    s11 = a33 - (t10 * a30 + t11 * a31)
    is_ = 1.0 / (s00 * s11 - s10 * s10)
    r00, r10, r11 = s11 * is_, -s10 * is_, s00 * is_
    u00, u01 = r00 * t00 + r10 * t10, r00 * t01 + r10 * t11
    u10, u11 = r10 * t00 + r11 * t10, r10 * t01 + r11 * t11
    e00 = -(p00 + t00 * u00 + t10 * u10)
    e10 = -(p10 + t01 * u00 + t11 * u10)
    e11 = -(p11 + t01 * u01 + t11 * u11)
    E = np.array([[e00, e10, u00, u10], [e10, e11, u01, u11], [u00, u01, -r00, -r10], [u10, u11, -r10, -r11]])
    return E


def inv16_blk(d):
    d = d.copy()
    for Kb in range(4):
        # uniform pivot block through readlane: P[i][j] = d[Kb] at lane 16 i + 4 Kb + j
        P = np.array([[d[Kb][16 * i + 4 * Kb + j] for j in range(4)] for i in range(4)])
        E = e4_closed(P)
        e_lane = np.array([E[LR[l], LC[l] & 3] for l in range(64)])
        eA = np.where(LC < 4, e_lane, 0.0)
        wt = mfma(eA, d[Kb], np.zeros((4, 64)))[0]          # lane (lr, lc): W[lc][lr]
        inK = (LC >> 2) == Kb
        bop = np.where(inK, np.where((LC & 3) == LR, -1.0, 0.0), d[Kb])
        cin = np.array([np.where(inK, 0.0, d[r]) for r in range(4)])
        d = mfma(wt, bop, cin)
        d[Kb] = np.where(inK, e_lane, -wt)
    return d


rng = np.random.default_rng(0)
for trial in range(5):
    B = rng.normal(size=(16, 16))
    D = B @ B.T + np.eye(16)
    s = np.array([1.0, 1.0, 60.0] * 5 + [1.0])
    D = D * s[:, None] * s[None, :]
    out = from_acc(inv16_blk(to_acc(D)))
    ref = -np.linalg.inv(D)
    print("trial", trial, "max rel err", np.abs(out - ref).max() / np.abs(ref).max(), "asym", np.abs(out - out.T).max())
