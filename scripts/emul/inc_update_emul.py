#!/usr/bin/env python
"""Numerical experiment behind the incremental (covariance-form) belief update — DEV TOOL, runs on the CPU.

Between relinearisations the iSAM2 restatement (oracle/drlgx_oracle.cpp: Isam::update) keeps the
linearisation point of every old variable, so one update only ADDS the new pose's odometry factor and this step's
bearing-range factors to the normal equations.  In covariance form (the arithmetic of FastMarginals2::propagate / update,
src/em_exploration/FastMarginals.cpp:188-321) that is

    new pose       d' = F d_p + c                 Sigma' = [[Sigma, Sigma_p^T F^T], [F Sigma_p, F Sigma_pp F^T + Q]]
    re-observed    T = R + A Sigma A^T            Sigma' = Sigma - Sigma A^T T^-1 A Sigma,   d' = d + K (-e - A d)
    new landmark   d_l = G d_x + c                as the pose, through the (square) landmark Jacobian

This script drives the CPU oracle through the bench's motion script (and a random walk), applies the incremental update in
numpy starting from the oracle's joint covariance after every update that relinearised, and prints the largest deviation
of the estimates / information blocks from the oracle's full re-solve — the drift the HIP kernel has to stay within
(estimates 1e-9 abs, information 1e-7 rel).
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def pose_mat(x, y, t):
    c, s = math.cos(t), math.sin(t)
    return np.array([x, y, c, s])


def between(p1, p2):
    """gtsam Pose2::between with H1 (oracle between())."""
    c = p1[2] * p2[2] + p1[3] * p2[3]
    s = -p1[3] * p2[2] + p1[2] * p2[3]
    dx, dy = p2[0] - p1[0], p2[1] - p1[1]
    rx = p1[2] * dx + p1[3] * dy
    ry = -p1[3] * dx + p1[2] * dy
    dt1 = -p2[3] * dx + p2[2] * dy
    dt2 = -p2[2] * dx - p2[3] * dy
    H1 = np.array([[-c, -s, dt1], [s, -c, dt2], [0, 0, -1.0]])
    return np.array([rx, ry, c, s]), H1


def odo_factor(th1, th2, meas):
    hx, H1 = between(th1, th2)
    h, _ = between(meas, hx)
    e = np.array([h[0], h[1], math.atan2(h[3], h[2])])
    Hl = np.array([[h[2], h[3], 0], [-h[3], h[2], 0], [0, 0, 1.0]])
    return e, Hl @ H1, Hl


def br_factor(p, l, bearing, rng):
    dx, dy = l[0] - p[0], l[1] - p[1]
    qx = p[2] * dx + p[3] * dy
    qy = -p[3] * dx + p[2] * dy
    d2 = qx * qx + qy * qy
    a, b = -qy / d2, qx / d2
    Hbx = np.array([-a, -b, a * qy - b * qx])
    Hbl = np.array([a * p[2] - b * p[3], a * p[3] + b * p[2]])
    n = math.sqrt(d2)
    bp = math.atan2(qy / n, qx / n)
    r = math.sqrt(dx * dx + dy * dy)
    ux, uy = dx / r, dy / r
    Hrx = np.array([-ux * p[2] - uy * p[3], ux * p[3] - uy * p[2], 0.0])
    Hrl = np.array([ux, uy])
    eb = math.atan2(math.sin(bp - bearing), math.cos(bp - bearing))
    return np.array([eb, r - rng]), np.vstack([Hbx, Hrx]), np.vstack([Hbl, Hrl])


class IncState(object):
    """Joint covariance in the oracle's variable order [landmarks by slot (2), poses (3)] kept as index maps."""

    def __init__(self, sim):
        self.reinit(sim)

    def reinit(self, sim):
        cov, L, P = sim.full_covariance()
        thp, dp, thl, dl, cnt = sim.isam_state()
        # own order: creation order is not needed in numpy: keep [landmarks, poses] with explicit maps
        self.lm_idx = [2 * j for j in range(L)]
        self.ps_idx = [2 * L + 3 * i for i in range(P)]
        self.S = cov.copy()
        self.d = np.concatenate([dl.reshape(-1), dp.reshape(-1)])
        self.thp = [pose_mat(*r) for r in thp]
        self.thl = [r.copy() for r in thl]
        self.nfac = len(sim.factors()[0])

    def grow(self, rows, cross, diag, dnew):
        n = self.S.shape[0]
        k = rows.shape[0]
        S2 = np.zeros((n + k, n + k))
        S2[:n, :n] = self.S
        S2[n:, :n] = rows
        S2[:n, n:] = rows.T
        S2[n:, n:] = diag
        self.S = S2
        self.d = np.concatenate([self.d, dnew])
        return n

    def step(self, sim_after, odom, cfg):
        """Apply the step the oracle just took (sim_after = oracle after the step)."""
        thp, dp, thl, dl, cnt = sim_after.isam_state()
        P_new = len(thp)
        assert P_new == len(self.thp) + 1
        # --- new pose: odometry factor between pose p and p+1 at (theta_p, theta_{p+1})
        p = P_new - 2
        th2 = pose_mat(*thp[-1])
        meas = pose_mat(*odom)
        sig = np.array([cfg.translation_noise, cfg.translation_noise, cfg.rotation_noise])
        e0, J1, J2 = odo_factor(self.thp[p], th2, meas)
        J2i = np.linalg.inv(J2)
        F = -J2i @ J1
        c = -J2i @ e0
        Q = J2i @ np.diag(sig ** 2) @ J2i.T
        ip = self.ps_idx[p]
        rows = F @ self.S[ip:ip + 3, :]
        diag = F @ self.S[ip:ip + 3, ip:ip + 3] @ F.T + Q
        inew = self.grow(rows, None, diag, F @ self.d[ip:ip + 3] + c)
        self.ps_idx.append(inew)
        self.thp.append(th2)
        # --- this step's bearing-range factors
        fp, fk, fb, fr = sim_after.factors()
        keys = list(sim_after.slot_keys())
        new_f = range(self.nfac, len(fp))
        self.nfac = len(fp)
        old, new = [], []
        for f in new_f:
            slot = keys.index(fk[f])
            assert fp[f] == P_new - 1
            (new if slot >= len(self.thl) else old).append((slot, fb[f], fr[f]))
        R = np.array([cfg.bearing_noise ** 2, cfg.range_noise ** 2])
        n = self.S.shape[0]
        if old:
            k = 2 * len(old)
            A = np.zeros((k, n))
            e = np.zeros(k)
            for r, (slot, b, rg) in enumerate(old):
                e0, Jx, Jl = br_factor(th2, self.thl[slot], b, rg)
                A[2 * r:2 * r + 2, inew:inew + 3] = Jx
                il = self.lm_idx[slot]
                A[2 * r:2 * r + 2, il:il + 2] = Jl
                e[2 * r:2 * r + 2] = e0
            Y = self.S @ A.T
            T = np.diag(np.tile(R, len(old))) + A @ Y
            Lc = np.linalg.cholesky(T)
            U = np.linalg.solve(Lc, Y.T).T  # Y L^-T
            self.S = self.S - U @ U.T
            self.d = self.d + U @ np.linalg.solve(Lc, -e - A @ self.d)
        for (slot, b, rg) in sorted(new):
            assert slot == len(self.thl)
            l = thl[slot]
            e0, Jx, Jl = br_factor(th2, l, b, rg)
            Jli = np.linalg.inv(Jl)
            G = -Jli @ Jx
            c = -Jli @ e0
            Q = Jli @ np.diag(R) @ Jli.T
            rows = G @ self.S[inew:inew + 3, :]
            diag = G @ self.S[inew:inew + 3, inew:inew + 3] @ G.T + Q
            il = self.grow(rows, None, diag, G @ self.d[inew:inew + 3] + c)
            self.lm_idx.append(il)
            self.thl.append(l.copy())

    def compare(self, sim):
        xyt, info = sim.poses()
        keys, lxy, linfo = sim.landmarks()
        slot_keys = list(sim.slot_keys())
        e_est = e_info = 0.0
        for i, ip in enumerate(self.ps_idx):
            th = self.thp[i]
            d = self.d[ip:ip + 3]
            c, s = math.cos(d[2]), math.sin(d[2])
            x = th[0] + th[2] * d[0] - th[3] * d[1]
            y = th[1] + th[3] * d[0] + th[2] * d[1]
            t = math.atan2(th[3] * c + th[2] * s, th[2] * c - th[3] * s)
            e_est = max(e_est, abs(x - xyt[i, 0]), abs(y - xyt[i, 1]), abs(math.remainder(t - xyt[i, 2], 2 * math.pi)))
            inf = np.linalg.inv(self.S[ip:ip + 3, ip:ip + 3])
            e_info = max(e_info, np.max(np.abs(inf - info[i]) / (1e-6 + 1e-7 * np.abs(info[i]))))
        for slot, il in enumerate(self.lm_idx):
            k = list(keys).index(slot_keys[slot])
            est = self.thl[slot] + self.d[il:il + 2]
            e_est = max(e_est, np.max(np.abs(est - lxy[k])))
            inf = np.linalg.inv(self.S[il:il + 2, il:il + 2])
            e_info = max(e_info, np.max(np.abs(inf - linfo[k]) / (1e-6 + 1e-7 * np.abs(linfo[k]))))
        return e_est, e_info


FULL_EVERY = int(os.environ.get('FULL_EVERY', '0'))


def run(lo, script, cfg, verbose=False):
    sim = O.OracleSim(cfg, lo, lo)
    inc = None
    worst = (0.0, 0.0)
    n_inc = n_full = 0
    run_len = 0
    for k, act in enumerate(script):
        thp0 = sim.isam_state()[0].copy() if sim.num_poses() else None
        thl0 = sim.isam_state()[2].copy() if sim.num_poses() else None
        P0 = sim.num_poses()
        ok = sim.simulate(act)
        if sim.num_poses() == P0:
            continue  # rejected move
        thp1, _, thl1, _, cnt = sim.isam_state()
        relin = inc is None or (FULL_EVERY and cnt % FULL_EVERY == 0) or not (np.array_equal(thp0, thp1[:len(thp0)]) and np.array_equal(thl0, thl1[:len(thl0)]))
        if relin:
            inc = IncState(sim) if inc is None else inc
            inc.reinit(sim)
            n_full += 1
            run_len = 0
        else:
            inc.step(sim, act, cfg)
            n_inc += 1
            run_len += 1
            e = inc.compare(sim)
            worst = (max(worst[0], e[0]), max(worst[1], e[1]))
            if verbose:
                print("  step %3d count %3d P %3d L %3d n %4d run %2d  est %.2e info %.2e" % (
                    k, cnt, sim.num_poses(), sim.num_landmarks(), inc.S.shape[0], run_len, e[0], e[1]))
    return worst, n_inc, n_full


if __name__ == "__main__":
    cfg = O.default_config(40, num_landmarks=100)
    warm = [(1, 1, math.pi / 2)] * 4 + [(2, 0, 0), (2, 0, 0), (0, 0, 0.6)] * 10 + [(2, 0, 0)] + [(2, 0, 0)] * 8
    for lo in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        w, ni, nf = run(lo, warm, cfg, verbose=(lo == 0))
        print("seed %d bench script: %d incremental / %d full updates, worst est %.2e info(allclose ratio) %.2e" % (lo, ni, nf, w[0], w[1]))
        rng = np.random.RandomState(lo)
        walk = [(1, 1, math.pi / 2)] * 4 + [(2.0 * (rng.rand() < 0.7), 0, rng.uniform(-0.8, 0.8)) for _ in range(120)]
        w, ni, nf = run(lo, walk, cfg)
        print("seed %d random walk : %d incremental / %d full updates, worst est %.2e info(allclose ratio) %.2e" % (lo, ni, nf, w[0], w[1]))
