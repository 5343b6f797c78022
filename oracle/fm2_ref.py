"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's EKF-style covariance propagate / update without
re-solving - `FastMarginals2::update / propagate` (src/em_exploration/FastMarginals.cpp:188-321) fed the way
`EMPlanner2D::updateNodeInformation_EM` / `updateTrajectory_EM` feed it (src/em_exploration/Planner2D.cpp:652-737,
:472-551): one new pose per action (noise-free odometry factor from the parent's ESTIMATE), noise-free bearing-range
factors to the landmarks of the estimated map that pass the sensor gates, linearised at the iSAM2 linearisation point
(old variables) / the predicted poses (new variables).  Imported by tests only; never by the product.

    Sigma' = Sigma - Sigma A^T (I + A Sigma A^T)^-1 A Sigma     (diagonal blocks of the updated keys)
"""
import math

import numpy as np


def _pose(x, y, th):
    return np.array([x, y, math.cos(th), math.sin(th)])


def _compose(a, b):
    c, s = a[2] * b[2] - a[3] * b[3], a[3] * b[2] + a[2] * b[3]
    return np.array([a[0] + a[2] * b[0] - a[3] * b[1], a[1] + a[3] * b[0] + a[2] * b[1], c, s])


def _between(p1, p2):
    """gtsam Pose2::between with H1."""
    c, s = p1[2] * p2[2] + p1[3] * p2[3], -p1[3] * p2[2] + p1[2] * p2[3]
    dx, dy = p2[0] - p1[0], p2[1] - p1[1]
    r = np.array([p1[2] * dx + p1[3] * dy, -p1[3] * dx + p1[2] * dy, c, s])
    dt1, dt2 = -p2[3] * dx + p2[2] * dy, -p2[2] * dx - p2[3] * dy
    H1 = np.array([[-c, -s, dt1], [s, -c, dt2], [0, 0, -1.0]])
    return r, H1


def _br_jacobians(p, l):
    """Whitening aside, the Jacobians of BearingRangeFactor at (pose p, point l): (2x3, 2x2), rows bearing, range."""
    dx, dy = l[0] - p[0], l[1] - p[1]
    qx, qy = p[2] * dx + p[3] * dy, -p[3] * dx + p[2] * dy  # transform_to
    d2 = qx * qx + qy * qy
    a, b = -qy / d2, qx / d2
    Hbx = np.array([-a, -b, a * qy - b * qx])
    Hbl = np.array([a * p[2] - b * p[3], a * p[3] + b * p[2]])
    r = math.sqrt(dx * dx + dy * dy)
    ux, uy = dx / r, dy / r
    Hrx = np.array([-ux * p[2] - uy * p[3], ux * p[3] - uy * p[2], 0.0])
    Hrl = np.array([ux, uy])
    return np.vstack([Hbx, Hrx]), np.vstack([Hbl, Hrl])


def fm2_update(sim, actions):
    """Updated 3x3 marginal covariances of every pose (the P old ones, then one per action) after appending the actions'
    predicted poses and measurements to the belief of OracleSim `sim`.  Returns (cov [P + K, 3, 3], n_meas per new pose)."""
    cfg = sim.cfg
    Sig, L, P = sim.full_covariance()
    thp, _, thl, _, _ = sim.isam_state()         # iSAM2 linearisation point (values of the old keys)
    est_xyt, _ = sim.poses()
    _, est_lm_sorted, _ = sim.landmarks()        # estimated map (sorted by key) - only for the gates
    keys_sorted, _, _ = sim.landmarks()
    # landmark estimates in SLOT order (the covariance / linearisation point order)
    key_of_slot = sim.slot_keys()
    est_lm = np.zeros((L, 2))
    for k, xy in zip(keys_sorted, est_lm_sorted):
        est_lm[list(key_of_slot).index(k)] = xy
    K = len(actions)
    sig = np.array([cfg.translation_noise, cfg.translation_noise, cfg.rotation_noise])
    sig_br = np.array([cfg.bearing_noise, cfg.range_noise])
    lidx = lambda j: slice(2 * j, 2 * j + 2)                      # noqa: E731
    pidx = lambda i: slice(2 * L + 3 * i, 2 * L + 3 * i + 3)      # noqa: E731
    # ---- odometry: cov1 = H1 (H0 cov0 H0^T + I) H1^T, F chains (FastMarginals.cpp:201-223)
    new_pose = []
    origin = _pose(*est_xyt[-1])                                  # parent->state.pose: the ESTIMATE
    val0 = _pose(*thp[-1])                                        # values.at(key0): the linearisation point
    cov_new, Fs, F_ = [], [], []
    cov0 = Sig[pidx(P - 1), pidx(P - 1)]
    F = np.eye(3)
    for a in actions:
        odom = _pose(*a)
        end = _compose(origin, odom)
        new_pose.append(end)
        hx, H1b = _between(val0, end)
        h, _ = _between(odom, hx)
        Hl = np.array([[h[2], h[3], 0], [-h[3], h[2], 0], [0, 0, 1.0]])
        A0 = (Hl @ H1b) / sig[:, None]                            # whitened block on key0
        A1 = Hl / sig[:, None]                                    # whitened block on key1
        H1 = np.linalg.inv(A1)
        cov1 = H1 @ (A0 @ cov0 @ A0.T + np.eye(3)) @ H1.T
        F = -H1 @ A0 @ F
        Fs.append(F.copy()); F_.append(-H1 @ A0); cov_new.append(cov1)
        origin, val0, cov0 = end, end, cov1

    # ---- covariance between any two keys (FastMarginals2::propagate, :286-318); keys: ("l", j) | ("x", i), i >= P new
    def prop(k0, k1):
        if k0 == k1:
            return cov_new[k0[1] - P] if k0[0] == "x" and k0[1] >= P else Sig[_ix(k0), _ix(k0)]
        new0, new1 = k0[0] == "x" and k0[1] >= P, k1[0] == "x" and k1[1] >= P
        if new0 and not new1:
            return prop(k1, k0).T
        if new0 and new1 and k0[1] > k1[1]:
            return prop(k1, k0).T
        if new1:
            if not new0:
                return prop(k0, ("x", P - 1)) @ Fs[k1[1] - P].T
            return prop(k0, ("x", k1[1] - 1)) @ F_[k1[1] - P].T
        return Sig[_ix(k0), _ix(k1)]

    def _ix(k):
        return lidx(k[1]) if k[0] == "l" else pidx(k[1])

    # ---- predicted measurements (Simulator2D::measure on the estimated map, noise-free; Planner2D.cpp:717-731)
    rows = []
    n_meas = []
    for k, pose in enumerate(new_pose):
        cnt = 0
        for j in range(L):
            dx, dy = est_lm[j][0] - pose[0], est_lm[j][1] - pose[1]
            rng = math.sqrt(dx * dx + dy * dy)
            if not rng < cfg.max_range:                           # searchLandmarkNeighbors(pose, max_range)
                continue
            qx, qy = pose[2] * dx + pose[3] * dy, -pose[3] * dx + pose[2] * dy
            bearing = math.atan2(qy, qx)
            if not (cfg.min_bearing < bearing < cfg.max_bearing and cfg.min_range < rng < cfg.max_range):
                continue
            Jx, Jl = _br_jacobians(pose, thl[j])                  # linearised at values: new pose, landmark at theta
            rows.append((("x", P + k), ("l", j), Jx / sig_br[:, None], Jl / sig_br[:, None]))
            cnt += 1
        n_meas.append(cnt)
    out = np.zeros((P + K, 3, 3))
    for i in range(P):
        out[i] = Sig[pidx(i), pidx(i)]
    for k in range(K):
        out[P + k] = cov_new[k]
    if not rows:
        return out, n_meas
    meas_keys = []
    for kx, kl, _, _ in rows:
        for kk in (kx, kl):
            if kk not in meas_keys:
                meas_keys.append(kk)
    col, c = {}, 0
    for kk in meas_keys:
        col[kk] = c
        c += 3 if kk[0] == "x" else 2
    cols, dim = c, 2 * len(rows)
    A = np.zeros((dim, cols))
    for r, (kx, kl, Jx, Jl) in enumerate(rows):
        A[2 * r:2 * r + 2, col[kx]:col[kx] + 3] = Jx
        A[2 * r:2 * r + 2, col[kl]:col[kl] + 2] = Jl
    SA = np.zeros((cols, cols))
    for k0 in meas_keys:
        for k1 in meas_keys:
            b = prop(k0, k1)
            SA[col[k0]:col[k0] + b.shape[0], col[k1]:col[k1] + b.shape[1]] = b
    S = A.T @ np.linalg.inv(np.eye(dim) + A @ SA @ A.T) @ A
    for i in range(P + K):
        key = ("x", i)
        Sg = np.zeros((3, cols))
        for kk in meas_keys:
            b = prop(key, kk)
            Sg[:, col[kk]:col[kk] + b.shape[1]] = b
        out[i] += -Sg @ S @ Sg.T
    return out, n_meas
