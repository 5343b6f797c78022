// SLAM belief update for long trajectories (any number of poses): the same linear system and outputs as slam_body
// (k_slam.hip), solved in the fill-reducing order of this problem instead of densely on the poses.
//
// Included by k_slam.hip inside namespace kslam.  Restates SLAM2D::optimize / copy_optimize
// (src/em_exploration/SLAM2D.cpp:374-488; gtsam ISAM2::update policy in SURVEY.md App. A.3) and the block marginals of
// FastMarginals (src/em_exploration/FastMarginals.cpp:130-186).
//
// Structure.  Information matrix in the order (poses, landmarks):
//     Lambda = [ T  B ; B^T  Lambda_ll ],   T block tridiagonal (prior + odometry chain + the pose side of the
//     bearing-range factors), B sparse 3P x 2L (one 3x2 block per factor), Lambda_ll block diagonal (2x2).
// Eliminating the pose chain first leaves a dense system on the landmarks only ("arrowhead" / bordered block
// tridiagonal): cost O(P L^2 + L^3) instead of O(P^3) for the landmark-first Schur complement of the fast path, which
// is what the reference's own episodes need (100-200 poses, a handful of landmarks).  Selected inverse:
//     X = T^-1 [B  eta_p]                     block LDL^T of T (chain recursion) + forward / backward substitution,
//                                             one thread per column, the recurrence state in registers
//     C = Lambda_ll - B^T X_B                 landmark Schur complement, rhs eta_l - B^T x_eta
//     [C r] -> -C^-1, delta_l                 the symmetric Gauss-Jordan sweep of the fast path (fp64 matrix cores)
//     delta_p = x_eta - X_B delta_l
//     Sigma_ll = C^-1 (its 2x2 diagonal blocks);  Sigma_ii = (T^-1)_ii + X_i C^-1 X_i^T   for every pose i,
//     (T^-1)_ii by the backward recursion  (T^-1)_ii = D'_i^-1 + L_{i+1,i}^T (T^-1)_{i+1,i+1} L_{i+1,i}.
// X (3P x (2L + 1)) and, beyond 63 landmarks, C live in the HBM/L2 workspace; everything else is in LDS.

// symmetric 3x3 (a00 a01 a02 a11 a12 a22) -> inverse in the same storage; returns the determinant's sign test
__device__ __forceinline__ bool inv3s(const double *a, double *o) {
  const double c00 = a[3] * a[5] - a[4] * a[4], c01 = a[2] * a[4] - a[1] * a[5], c02 = a[1] * a[4] - a[2] * a[3];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const double id = fast_rcp(det);
  o[0] = c00 * id;
  o[1] = c01 * id;
  o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id;
  o[4] = (a[1] * a[2] - a[0] * a[4]) * id;
  o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
  return a[0] > 0 && c00 > 0 && det > 0;  // leading minors of the inverse's cofactors: SPD test
}
__device__ __forceinline__ double sym3(const double *a, int r, int c) {
  constexpr int idx[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
  return a[idx[r * 3 + c]];
}

// LDS bytes of the per-pose / per-landmark tables of arrow_body (without the landmark system, records and tables)
__host__ __device__ inline size_t arrow_small_bytes(int P, int L, int M) {
  const size_t MW = (size_t)(P + 63) >> 6;
  return (size_t)P * (4 + 6 + 9 + 3 + 6) * 8 + (size_t)L * (2 + 8) * 8 + (((size_t)(P + 2) * 4 + 7) & ~(size_t)7) +
         2 * (((size_t)M * 2 + 7) & ~(size_t)7) + (size_t)L * MW * 8 + 64;
}

template <int NTW>
__device__ __forceinline__ void arrow_body(const DrlgxState &S, const LaunchSel &sel, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int bi = blockIdx.x;
  if (!sel.on(bi)) return;
  const int inst = sel.base + bi;
  int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  if (cnt[C_FLAG]) return;
  const drlgx_config &cfg = S.cfg;
  const int P = cnt[C_P], L = cnt[C_L], M = cnt[C_M];
  const int n_old_p = cnt[C_NEWP], n_old_l = cnt[C_NEWL];
  const int count = cnt[C_ISAM] + 1;
  const int np = 2 * L, ncol = np + 1;              // landmark system: pivots [0, 2L), rhs row 2L
  const int Tn = (ncol + 15) / 16, N = 16 * Tn;
  const int ntiles = Tn * (Tn + 1) / 2;
  const bool c_lds = Tn <= kFastTilesArrow;          // <= 63 landmarks: packed system + sweep panels in LDS
  const int ldx = (ncol + 3) & ~3;                   // row stride of X
  if (!c_lds && (NTW == 0 || ntiles > NTW * (kWaves - 1))) {
    if (tid == 0) atomicMin(S.status, DRLGX_E_CAPACITY);
    return;
  }
  DRLGX_PROF(S, 0);

  // ---- LDS carve ----
  size_t off = 0;
  auto take = [&](size_t bytes) {
    unsigned char *q = smem_raw + off;
    off += up8(bytes);
    return q;
  };
  double *thp = reinterpret_cast<double *>(take((size_t)P * 4 * 8));
  double *Dd = reinterpret_cast<double *>(take((size_t)P * 6 * 8));   // D_i -> D'_i^-1 (symmetric)
  double *Oo = reinterpret_cast<double *>(take((size_t)P * 9 * 8));   // T_{i+1,i} -> L_{i+1,i}
  double *gp = reinterpret_cast<double *>(take((size_t)P * 3 * 8));   // eta_p = -g
  double *Ti = reinterpret_cast<double *>(take((size_t)P * 6 * 8));   // (T^-1)_ii (symmetric)
  double *thl = reinterpret_cast<double *>(take((size_t)L * 2 * 8));
  double *lamb = reinterpret_cast<double *>(take((size_t)L * 8 * 8));
  int *mstart = reinterpret_cast<int *>(take((size_t)(P + 2) * 4));
  unsigned short *mp = reinterpret_cast<unsigned short *>(take((size_t)M * 2));
  unsigned short *ml = reinterpret_cast<unsigned short *>(take((size_t)M * 2));
  int *bad = reinterpret_cast<int *>(take(8));
  const int MW = (P + 63) >> 6;
  unsigned long long *lmask = reinterpret_cast<unsigned long long *>(take((size_t)L * MW * 8));
  off = (off + 31) & ~(size_t)31;
  double *wsd = S.slam_ws + (size_t)inst * S.slam_ws_stride;
  double *X = wsd; wsd += (size_t)3 * S.P_max * (size_t)((2 * S.L_max + 1 + 3) & ~3);
  double *A;       // landmark system: packed lower triangle in LDS, or square (ld = N) in the workspace
  double *panels;  // sweep panels of the workspace variant
  if (c_lds) {
    A = reinterpret_cast<double *>(smem_raw + off);
    off += max((size_t)N * (N + 1) / 2, (size_t)48 * N + 1280) * 8;
    panels = nullptr;
  } else {
    A = wsd; wsd += (size_t)(2 * S.L_max + 17) * (2 * S.L_max + 17);
    panels = reinterpret_cast<double *>(smem_raw + off);
    off += ((size_t)32 * N + 1280) * 8;
  }
  auto AT = [&](int i, int j) -> int { return c_lds ? i * (i + 1) / 2 + j : i * N + j; };
  const size_t big = (size_t)M * REC * 8 + up8((size_t)L * P * 2);
  double *rec;
  unsigned short *obs;
  if (off + big <= (size_t)lds_bytes) {
    rec = reinterpret_cast<double *>(smem_raw + off); off += (size_t)M * REC * 8;
    obs = reinterpret_cast<unsigned short *>(smem_raw + off);
  } else {
    rec = wsd; wsd += (size_t)S.M_max * REC;
    obs = reinterpret_cast<unsigned short *>(wsd);
  }
  double *th_pose = S.th_pose + (size_t)inst * S.P_max * 4;
  double *d_pose = S.d_pose + (size_t)inst * S.P_max * 3;
  double *th_lm = S.th_lm + (size_t)inst * S.L_max * 2;
  double *d_lm = S.d_lm + (size_t)inst * S.L_max * 2;
  const int *meas_pose = S.meas_pose + (size_t)inst * S.M_max;
  const int *meas_lm = S.meas_lm + (size_t)inst * S.M_max;
  const double *meas_br = S.meas_br + (size_t)inst * S.M_max * 2;

  // ---- 1. relinearisation policy (gtsam ISAM2: relinearizeSkip 10, relinearizeThreshold 0.1); theta staged in LDS ----
  const bool relin = (count % 10 == 0);
  for (int i = tid; i < P; i += kThreads) {
    Pose t{th_pose[4 * i], th_pose[4 * i + 1], th_pose[4 * i + 2], th_pose[4 * i + 3]};
    if (relin && i < n_old_p) {
      const double a = fabs(d_pose[3 * i]), b = fabs(d_pose[3 * i + 1]), c = fabs(d_pose[3 * i + 2]);
      if (fmax(a, fmax(b, c)) >= 0.1) {
        t = compose(t, make_pose(d_pose[3 * i], d_pose[3 * i + 1], d_pose[3 * i + 2]));
        th_pose[4 * i] = t.x; th_pose[4 * i + 1] = t.y; th_pose[4 * i + 2] = t.c; th_pose[4 * i + 3] = t.s;
      }
    }
    thp[4 * i] = t.x; thp[4 * i + 1] = t.y; thp[4 * i + 2] = t.c; thp[4 * i + 3] = t.s;
  }
  for (int j = tid; j < L; j += kThreads) {
    double x = th_lm[2 * j], y = th_lm[2 * j + 1];
    if (relin && j < n_old_l && fmax(fabs(d_lm[2 * j]), fabs(d_lm[2 * j + 1])) >= 0.1) {
      x += d_lm[2 * j];
      y += d_lm[2 * j + 1];
      th_lm[2 * j] = x;
      th_lm[2 * j + 1] = y;
    }
    thl[2 * j] = x;
    thl[2 * j + 1] = y;
  }
  // ---- 2. tables + one linearisation per factor ----
  {
    const size_t nA = c_lds ? (size_t)N * (N + 1) / 2 : (size_t)N * N;
    for (size_t e = tid; e < nA; e += kThreads) A[e] = 0.0;
  }
  for (int e = tid; e < L * P; e += kThreads) obs[e] = 0;
  for (int e = tid; e < MW * L; e += kThreads) lmask[e] = 0ull;
  for (int e = tid; e <= P; e += kThreads) mstart[e] = M;
  if (tid == 0) bad[0] = 0;
  __syncthreads();
  for (int m = tid; m < M; m += kThreads) {
    const int p = meas_pose[m], j = meas_lm[m];
    mp[m] = (unsigned short)p;
    ml[m] = (unsigned short)j;
    if (m == 0 || meas_pose[m - 1] != p) mstart[p] = m;
    obs[j * P + p] = (unsigned short)(m + 1);
    atomicOr(&lmask[MW * j + (p >> 6)], 1ull << (p & 63));
    linearize_br(thp + 4 * p, thl + 2 * j, meas_br[2 * m], meas_br[2 * m + 1], rec + (size_t)REC * m);
  }
  __syncthreads();
  for (int p0 = 0; p0 < P; p0 += kThreads) {
    // poses without factors get the empty range [next pose's start, same): first assigned start at or after p
    const int q0 = p0 + tid;
    int v = M;
    if (q0 < P) {
      int q = q0;
      v = mstart[q];
      while (v == M && q < P) v = mstart[++q];  // mstart[P] = M
    }
    __syncthreads();
    if (q0 < P) mstart[q0] = v;
    __syncthreads();
  }
  DRLGX_PROF(S, 1);
  // ---- 3. blocks: Lambda_jj, eta_j per landmark; D_i, eta_p,i, T_{i+1,i} per pose ----
  const double wb = 1.0 / (cfg.bearing_noise * cfg.bearing_noise), wr = 1.0 / (cfg.range_noise * cfg.range_noise);
  const int pose_t0 = ((L + 63) & ~63) % kThreads;
  for (int j = tid; j < L; j += kThreads) {
    double a = 0, b = 0, d = 0, g0 = 0, g1 = 0;
    FOR_EACH_OBSERVING_POSE(lmask + MW * j, MW, p) {
      const double *r = rec + (size_t)REC * (obs[j * P + p] - 1);
      a += r[6] * wb * r[6] + r[8] * wr * r[8];
      b += r[6] * wb * r[7] + r[8] * wr * r[9];
      d += r[7] * wb * r[7] + r[9] * wr * r[9];
      g0 += r[6] * wb * r[10] + r[8] * wr * r[11];
      g1 += r[7] * wb * r[10] + r[9] * wr * r[11];
    }
    double *lb = lamb + 8 * j;
    lb[0] = a; lb[1] = b; lb[2] = d;
    lb[6] = -g0; lb[7] = -g1;  // eta_j
  }
  for (int i = (tid - pose_t0 + kThreads) % kThreads; i < P; i += kThreads) {
    double B[9], g[3], O[9];
    pose_block(S, inst, thp, rec, mstart, i, P, wb, wr, B, g, O);
    double *dd = Dd + 6 * i;
    dd[0] = B[0]; dd[1] = B[3]; dd[2] = B[6]; dd[3] = B[4]; dd[4] = B[7]; dd[5] = B[8];
    for (int r = 0; r < 3; ++r) gp[3 * i + r] = -g[r];
    if (i + 1 < P)
      for (int k = 0; k < 9; ++k) Oo[9 * i + k] = O[k];
  }
  __syncthreads();
  // ---- 4. B_m = Jx^T W Jl (3x2, row major) replaces Jx in the factor record ----
  for (int m = tid; m < M; m += kThreads) {
    double *l = rec + (size_t)REC * m;
    double bm[6];
    for (int r = 0; r < 3; ++r) {
      bm[r * 2 + 0] = l[r] * wb * l[6] + l[3 + r] * wr * l[8];
      bm[r * 2 + 1] = l[r] * wb * l[7] + l[3 + r] * wr * l[9];
    }
    for (int k = 0; k < 6; ++k) l[k] = bm[k];
  }
  DRLGX_PROF(S, 2);
  // ---- 5. block LDL^T of the chain: D'_0 = D_0, L_{i+1,i} = T_{i+1,i} D'_i^-1, D'_{i+1} = D_{i+1} - L_{i+1,i} T_{i+1,i}^T
  //         (sequential; every lane of wave 0 computes it, lane 0 stores).  Dd <- D'^-1, Oo <- L. ----
  if (tid < 64) {
    double dcur[6];
    for (int k = 0; k < 6; ++k) dcur[k] = Dd[k];
    bool ok = true;
    for (int i = 0; i < P; ++i) {
      double di[6];
      ok = inv3s(dcur, di) && ok;
      double lf[9], o[9];
      if (i + 1 < P) {
        for (int k = 0; k < 9; ++k) o[k] = Oo[9 * i + k];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c)
            lf[r * 3 + c] = o[r * 3] * sym3(di, 0, c) + o[r * 3 + 1] * sym3(di, 1, c) + o[r * 3 + 2] * sym3(di, 2, c);
        const double *dn = Dd + 6 * (i + 1);
        // D'_{i+1} = D_{i+1} - L T^T (symmetric)
        dcur[0] = dn[0] - (lf[0] * o[0] + lf[1] * o[1] + lf[2] * o[2]);
        dcur[1] = dn[1] - (lf[0] * o[3] + lf[1] * o[4] + lf[2] * o[5]);
        dcur[2] = dn[2] - (lf[0] * o[6] + lf[1] * o[7] + lf[2] * o[8]);
        dcur[3] = dn[3] - (lf[3] * o[3] + lf[4] * o[4] + lf[5] * o[5]);
        dcur[4] = dn[4] - (lf[3] * o[6] + lf[4] * o[7] + lf[5] * o[8]);
        dcur[5] = dn[5] - (lf[6] * o[6] + lf[7] * o[7] + lf[8] * o[8]);
      }
      if (tid == 0) {
        for (int k = 0; k < 6; ++k) Dd[6 * i + k] = di[k];
        if (i + 1 < P)
          for (int k = 0; k < 9; ++k) Oo[9 * i + k] = lf[k];
      }
    }
    if (tid == 0 && !ok) bad[0] = 1;
  }
  __syncthreads();
  DRLGX_PROF(S, 3);
  // ---- 6. X = T^-1 [B eta_p]: one thread per column, forward then backward substitution with the state in
  //         registers; the last wave runs the (T^-1)_ii recursion meanwhile ----
  if (tid < ncol) {
    const int c = tid;
    const bool is_rhs = c == np;
    const int j = c >> 1, a = c & 1;
    double y0 = 0, y1 = 0, y2 = 0;
    for (int i = 0; i < P; ++i) {
      double b0, b1, b2;
      if (is_rhs) {
        b0 = gp[3 * i]; b1 = gp[3 * i + 1]; b2 = gp[3 * i + 2];
      } else {
        const int m1 = obs[j * P + i];
        const double *bm = rec + (size_t)REC * (m1 ? m1 - 1 : 0);
        b0 = m1 ? bm[a] : 0.0; b1 = m1 ? bm[2 + a] : 0.0; b2 = m1 ? bm[4 + a] : 0.0;
      }
      if (i > 0) {
        const double *lf = Oo + 9 * (i - 1);
        const double t0 = b0 - (lf[0] * y0 + lf[1] * y1 + lf[2] * y2);
        const double t1 = b1 - (lf[3] * y0 + lf[4] * y1 + lf[5] * y2);
        const double t2 = b2 - (lf[6] * y0 + lf[7] * y1 + lf[8] * y2);
        y0 = t0; y1 = t1; y2 = t2;
      } else {
        y0 = b0; y1 = b1; y2 = b2;
      }
      X[(size_t)(3 * i) * ldx + c] = y0;
      X[(size_t)(3 * i + 1) * ldx + c] = y1;
      X[(size_t)(3 * i + 2) * ldx + c] = y2;
    }
    double x0 = 0, x1 = 0, x2 = 0;
    for (int i = P - 1; i >= 0; --i) {
      const double *di = Dd + 6 * i;
      double t0 = di[0] * y0 + di[1] * y1 + di[2] * y2;
      double t1 = di[1] * y0 + di[3] * y1 + di[4] * y2;
      double t2 = di[2] * y0 + di[4] * y1 + di[5] * y2;
      if (i + 1 < P) {
        const double *lf = Oo + 9 * i;  // L_{i+1,i}^T x_{i+1}
        t0 -= lf[0] * x0 + lf[3] * x1 + lf[6] * x2;
        t1 -= lf[1] * x0 + lf[4] * x1 + lf[7] * x2;
        t2 -= lf[2] * x0 + lf[5] * x1 + lf[8] * x2;
      }
      x0 = t0; x1 = t1; x2 = t2;
      X[(size_t)(3 * i) * ldx + c] = x0;
      X[(size_t)(3 * i + 1) * ldx + c] = x1;
      X[(size_t)(3 * i + 2) * ldx + c] = x2;
      if (i > 0) {  // this thread's own forward values of the previous pose
        y0 = X[(size_t)(3 * i - 3) * ldx + c];
        y1 = X[(size_t)(3 * i - 2) * ldx + c];
        y2 = X[(size_t)(3 * i - 1) * ldx + c];
      }
    }
  } else if (tid >= kThreads - 64) {
    // (T^-1)_ii = D'_i^-1 + L_{i+1,i}^T (T^-1)_{i+1,i+1} L_{i+1,i}, from the last pose backwards
    double t[6];
    for (int k = 0; k < 6; ++k) t[k] = Dd[6 * (P - 1) + k];
    if (tid == kThreads - 64)
      for (int k = 0; k < 6; ++k) Ti[6 * (P - 1) + k] = t[k];
    for (int i = P - 2; i >= 0; --i) {
      const double *lf = Oo + 9 * i, *di = Dd + 6 * i;
      double w[9];  // W = (T^-1)_{i+1,i+1} L
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) w[r * 3 + c] = sym3(t, r, 0) * lf[c] + sym3(t, r, 1) * lf[3 + c] + sym3(t, r, 2) * lf[6 + c];
      double n[6];
      n[0] = di[0] + lf[0] * w[0] + lf[3] * w[3] + lf[6] * w[6];
      n[1] = di[1] + lf[0] * w[1] + lf[3] * w[4] + lf[6] * w[7];
      n[2] = di[2] + lf[0] * w[2] + lf[3] * w[5] + lf[6] * w[8];
      n[3] = di[3] + lf[1] * w[1] + lf[4] * w[4] + lf[7] * w[7];
      n[4] = di[4] + lf[1] * w[2] + lf[4] * w[5] + lf[7] * w[8];
      n[5] = di[5] + lf[2] * w[2] + lf[5] * w[5] + lf[8] * w[8];
      for (int k = 0; k < 6; ++k) t[k] = n[k];
      if (tid == kThreads - 64)
        for (int k = 0; k < 6; ++k) Ti[6 * i + k] = t[k];
    }
  }
  __syncthreads();
  DRLGX_PROF(S, 4);
  // ---- 7. landmark system [C r]: rows 2j, 2j+1 at column c (lower triangle + the rhs column) ----
  for (int e = tid; e < L * ncol; e += kThreads) {
    const int j = e / ncol, c = e - j * ncol;
    if (c != np && c > 2 * j + 1) continue;
    double a0 = 0, a1 = 0;
    FOR_EACH_OBSERVING_POSE(lmask + MW * j, MW, i) {
      const double *bm = rec + (size_t)REC * (obs[j * P + i] - 1);
      const double x0 = X[(size_t)(3 * i) * ldx + c], x1 = X[(size_t)(3 * i + 1) * ldx + c], x2 = X[(size_t)(3 * i + 2) * ldx + c];
      a0 += bm[0] * x0 + bm[2] * x1 + bm[4] * x2;
      a1 += bm[1] * x0 + bm[3] * x1 + bm[5] * x2;
    }
    const double *lb = lamb + 8 * j;
    if (c == np) {
      A[AT(np, 2 * j)] = lb[6] - a0;
      A[AT(np, 2 * j + 1)] = lb[7] - a1;
    } else {
      const bool own = (c >> 1) == j;
      const double l0 = own ? ((c & 1) ? lb[1] : lb[0]) : 0.0, l1 = own ? ((c & 1) ? lb[2] : lb[1]) : 0.0;
      if (c <= 2 * j) A[AT(2 * j, c)] = l0 - a0;
      A[AT(2 * j + 1, c)] = l1 - a1;
    }
  }
  __syncthreads();
  DRLGX_PROF(S, 5);
  // ---- 8. sweep: A <- -C^-1 (lower triangle), row np <- delta_l ----
  if (c_lds)
    sweep_packed_fast<kFastTilesArrow>(S, A, np, N, Tn, bad, tid);
  else if constexpr (NTW > 0)
    sweep_regtiles<false, NTW>(A, panels, np, N, Tn, ntiles, bad, tid);
  __syncthreads();
  DRLGX_PROF(S, 6);
  // ---- 9. landmark outputs ----
  double *est_lm = S.est_lm + (size_t)inst * S.L_max * 2;
  double *lm_info = S.lm_info + (size_t)inst * S.L_max * 3;
  double *lm_tr = S.lm_tr + (size_t)inst * S.L_max;
  for (int j = tid; j < L; j += kThreads) {
    const double dx = A[AT(np, 2 * j)], dy = A[AT(np, 2 * j + 1)];
    d_lm[2 * j] = dx;
    d_lm[2 * j + 1] = dy;
    est_lm[2 * j] = thl[2 * j] + dx;
    est_lm[2 * j + 1] = thl[2 * j + 1] + dy;
    const double c00 = -A[AT(2 * j, 2 * j)], cs = -A[AT(2 * j + 1, 2 * j)], c11 = -A[AT(2 * j + 1, 2 * j + 1)];
    lm_tr[j] = c00 + c11;
    const double id = 1.0 / (c00 * c11 - cs * cs);  // marginalCovariance(l).inverse() (SLAM2D.cpp:417)
    lm_info[3 * j] = c11 * id;
    lm_info[3 * j + 1] = -cs * id;
    lm_info[3 * j + 2] = c00 * id;
  }
  // ---- 10. pose outputs: delta_p = x_eta - X_B delta_l; Sigma_ii = (T^-1)_ii + X_i C^-1 X_i^T.  One 16-lane row per
  //          pose: lane l handles the columns c = l, l + 16, ...; partial sums reduced over the row in a fixed order ----
  double *est_pose = S.est_pose + (size_t)inst * S.P_max * 4;
  double *pose_info = S.pose_info + (size_t)inst * S.P_max * 6;
  double *pose_tr = S.pose_tr + (size_t)inst * S.P_max;
  {
    const int sub = tid & 15, grp = tid >> 4, ngrp = kThreads / 16;
    for (int i = grp; i < P; i += ngrp) {
      const double *x0r = X + (size_t)(3 * i) * ldx, *x1r = x0r + ldx, *x2r = x1r + ldx;
      double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // dp0 dp1 dp2, then the 6 entries of X_i (-C^-1) X_i^T
      for (int c = sub; c < np; c += 16) {
        // z_r = sum_k X_i[r][k] (-C^-1)[k][c]
        double z0 = 0, z1 = 0, z2 = 0;
        for (int k = 0; k < np; ++k) {
          const double m = A[AT(max(k, c), min(k, c))];
          z0 += x0r[k] * m;
          z1 += x1r[k] * m;
          z2 += x2r[k] * m;
        }
        const double xc0 = x0r[c], xc1 = x1r[c], xc2 = x2r[c], dl = A[AT(np, c)];
        s[0] += xc0 * dl; s[1] += xc1 * dl; s[2] += xc2 * dl;
        s[3] += z0 * xc0; s[4] += z0 * xc1; s[5] += z0 * xc2;
        s[6] += z1 * xc1; s[7] += z1 * xc2; s[8] += z2 * xc2;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        double v = s[k];
        v += __shfl_xor(v, 8, 16);
        v += __shfl_xor(v, 4, 16);
        v += __shfl_xor(v, 2, 16);
        v += __shfl_xor(v, 1, 16);
        s[k] = v;
      }
      if (sub == 0) {
        const double dp0 = x0r[np] - s[0], dp1 = x1r[np] - s[1], dp2 = x2r[np] - s[2];
        d_pose[3 * i] = dp0; d_pose[3 * i + 1] = dp1; d_pose[3 * i + 2] = dp2;
        const Pose t{thp[4 * i], thp[4 * i + 1], thp[4 * i + 2], thp[4 * i + 3]};
        const Pose e = compose(t, make_pose(dp0, dp1, dp2));
        est_pose[4 * i] = e.x; est_pose[4 * i + 1] = e.y; est_pose[4 * i + 2] = e.c; est_pose[4 * i + 3] = e.s;
        const double *ti = Ti + 6 * i;
        const double c00 = ti[0] - s[3], c10 = ti[1] - s[4], c20 = ti[2] - s[5];
        const double c11 = ti[3] - s[6], c21 = ti[4] - s[7], c22 = ti[5] - s[8];
        pose_tr[i] = c00 + c11 + c22;
        LLT3 llt(c00, c10, c20, c11, c21, c22);  // information = inverse(covariance) by LLT (SLAM2D.cpp:395-408)
        double q0, q1, q2;
        double *pi = pose_info + 6 * i;
        llt.solve(1, 0, 0, q0, q1, q2);
        pi[0] = q0; pi[1] = q1; pi[2] = q2;
        llt.solve(0, 1, 0, q0, q1, q2);
        pi[3] = q1; pi[4] = q2;
        llt.solve(0, 0, 1, q0, q1, q2);
        pi[5] = q2;
      }
    }
  }
  DRLGX_PROF(S, 7);
  if (tid == 0) {
    cnt[C_ISAM] = count;
    cnt[C_NEWP] = P;
    cnt[C_NEWL] = L;
    if (bad[0]) atomicMin(S.status, DRLGX_E_NUMERIC);
  }
}

template <int NTW>
__global__ __launch_bounds__(kThreads) void k_slam_arrow(DrlgxState S, LaunchSel sel, int lds_bytes) {
  arrow_body<NTW>(S, sel, lds_bytes);
}
