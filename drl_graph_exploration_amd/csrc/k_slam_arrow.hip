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
//     X = T^-1 [B  eta_p]                     block cyclic reduction of the chain (= nested dissection of the pose
//                                             chain: log2 P levels, every level parallel over its poses x columns)
//     C = Lambda_ll - B^T X_B                 landmark Schur complement, rhs eta_l - B^T x_eta
//     [C r] -> -C^-1, delta_l                 the symmetric Gauss-Jordan sweep of the fast path (fp64 matrix cores)
//     delta_p = x_eta - X_B delta_l
//     Sigma_ll = C^-1 (its 2x2 diagonal blocks);  Sigma_ii = (T^-1)_ii + X_i C^-1 X_i^T   for every pose i,
//     (T^-1)_ii by the Takahashi recursion over the same elimination tree (top level down, parallel per level).
// X (3P x (2L + 1)), the selected-inverse blocks and, beyond 63 landmarks, C live in the HBM/L2 workspace; the chain
// factors and the landmark tables are in LDS.
//
// Cyclic reduction.  Level l (stride s = 2^l) eliminates the poses i = s, 3s, 5s, ... from the poses that are multiples of
// s; their neighbours i - s and i + s survive.  With A_i = T_{i,i-s} (coupling to the left neighbour at that level):
//     E_i = D_i^-1,  GL_i = E_i A_i,  GR_i = E_i A_{i+s}^T                                      (eliminated i)
//     D_j -= A_j GR_{j-s} + A_{j+s}^T GL_{j+s},  A_j <- -A_j GL_{j-s},
//     b_j -= GR_{j-s}^T b_{j-s} + GL_{j+s}^T b_{j+s}                                           (surviving j)
// and back down:  x_i = E_i b_i - GL_i x_{i-s} - GR_i x_{i+s};  Takahashi: S_{l,i} = -(S_ll GL_i^T + S_lr GR_i^T),
// S_{r,i} = -(S_rl GL_i^T + S_rr GR_i^T), S_ii = E_i - GL_i S_{l,i} - GR_i S_{r,i}, where the cross block S_lr of the two
// neighbours is the one stored by whichever of them is eliminated at the next level.

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

// ascending iteration over the set bits of a W-word mask, up to four at a time (so that the loads of four
// observations can be issued before the first one is used)
struct MaskIter {
  const unsigned long long *mk;
  int W, w;
  unsigned long long m;
  __device__ __forceinline__ MaskIter(const unsigned long long *mk_, int W_) : mk(mk_), W(W_), w(0), m(mk_[0]) {}
  __device__ __forceinline__ int next4(int (&ip)[4]) {
    int n = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      while (!m && w + 1 < W) m = mk[++w];
      if (m) {
        ip[u] = 64 * w + __ffsll((long long)m) - 1;
        m &= m - 1;
        n = u + 1;
      } else {
        ip[u] = ip[0];
      }
    }
    return n;
  }
};

// LDS bytes of the per-pose / per-landmark tables of arrow_body (without the landmark system, records and tables)
__host__ __device__ inline size_t arrow_small_bytes(int P, int L, int M) {
  const size_t MW = (size_t)(P + 63) >> 6;
  return (size_t)P * (4 + 6 + 9 + 9 + 9) * 8 + (size_t)L * (2 + 8) * 8 + (((size_t)(P + 2) * 4 + 7) & ~(size_t)7) +
         (size_t)L * MW * 8 + 64;
}

constexpr int kSegLog = 3, kSeg = 1 << kSegLog;  // leaf segments of the chain: 7 interior poses between separators

template <int NTW>
__device__ __forceinline__ void arrow_body(const DrlgxState &S, const LaunchSel &sel, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = drlgx_tid();
  const int bi = drlgx_bid();
  if (!sel.on(bi)) return;
  if (S.prof && tid == 0 && bi < 448) S.prof[128 + 2 * bi] = wall_clock64();  // (dev aid: per-workgroup start / end, as k_step)
  if (inc_stage<ISNT>(S, sel, lds_bytes, 0)) {  // between relinearisations: the rank-k covariance update (k_inc.hip)
    if (S.prof && tid == 0 && bi < 448) S.prof[129 + 2 * bi] = wall_clock64();
    return;
  }
  const int inst = sel.base + bi;
  int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  // (`full` / `refresh`: see slam_body - intermediate look-ahead steps solve for the estimates only.  Not with the
  // incremental update on: a full solve is what leaves the covariance panel, and the rollout's remaining actions then cost
  // a rank-k update each instead of another solve - at the bench state the relinearising 10th update is action 9 of up to
  // 11, and the two actions behind it were 1.3 of the look-ahead's 5.2 ms)
  const bool want = sel.map_on(bi), full = want || S.jc != nullptr;
  const bool refresh = cnt[C_FLAG] != 0;
  if (refresh && !(sel.map_last_only && sel.n_act && want)) return;
  const drlgx_config &cfg = S.cfg;
  const int P = cnt[C_P], L = cnt[C_L], M = cnt[C_M];
  const int n_old_p = cnt[C_NEWP], n_old_l = cnt[C_NEWL];
  const int count = cnt[C_ISAM] + (refresh ? 0 : 1);
  const int np = 2 * L, ncol = np + 1;              // landmark system: pivots [0, 2L), rhs row 2L
  const int Tn = (ncol + 15) / 16, N = 16 * Tn;
  const int ntiles = Tn * (Tn + 1) / 2;
  const bool c_lds = Tn <= kFastTilesArrow;          // <= 63 landmarks: packed system + sweep panels in LDS
  // With the incremental update on (S.jc), the chain solve carries three more right-hand sides - the unit columns of the
  // newest pose - so that X also holds (T^-1)[., pn]: the covariance panel needs every pose's cross block with it.
  const bool mk_panel = S.jc != nullptr && !refresh && full && P >= 1;
  const int ncx = mk_panel ? ncol + 3 : ncol;        // columns of [B eta_p (E_pn)]
  const int ldx = (ncol + 3 + 3) & ~3;               // row stride of X (room for those columns whether used or not)
  const bool c_reg = !c_lds && NTW > 0 && ntiles <= NTW * (kWaves - 1);  // lower tiles in registers, panels in LDS
  if (!c_lds && NTW == 0) {  // (this instantiation serves engines whose landmark capacity always fits the LDS)
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
  double *Dd = reinterpret_cast<double *>(take((size_t)P * 6 * 8));   // D_i (symmetric) -> E_i = its inverse when i is eliminated
  double *Al = reinterpret_cast<double *>(take((size_t)P * 9 * 8));   // A_i = T_{i,i-s}: coupling to the current left neighbour
  double *GL = reinterpret_cast<double *>(take((size_t)P * 9 * 8));   // E_i A_i
  double *GR = reinterpret_cast<double *>(take((size_t)P * 9 * 8));   // E_i A_{i+s}^T
  double *thl = reinterpret_cast<double *>(take((size_t)L * 2 * 8));
  double *lamb = reinterpret_cast<double *>(take((size_t)L * 8 * 8));
  int *mstart = reinterpret_cast<int *>(take((size_t)(P + 2) * 4));
  int *bad = reinterpret_cast<int *>(take(8));
  const int MW = (P + 63) >> 6;
  unsigned long long *lmask = reinterpret_cast<unsigned long long *>(take((size_t)L * MW * 8));
  off = (off + 31) & ~(size_t)31;
  double *wsd = S.slam_ws + (size_t)inst * S.slam_ws_stride;
  double *X = wsd; wsd += (size_t)3 * S.P_max * (size_t)((2 * S.L_max + 1 + 3 + 3) & ~3);
  double *Ti = wsd; wsd += (size_t)6 * S.P_max;    // (T^-1)_ii, symmetric
  double *Sl = wsd; wsd += (size_t)9 * S.P_max;    // (T^-1)_{i-s,i} at i's elimination level
  double *Sr = wsd; wsd += (size_t)9 * S.P_max;    // (T^-1)_{i+s,i}
  double *sepR = wsd; wsd += (size_t)(S.P_max / kSeg + 2) * 3 * (size_t)((2 * S.L_max + 1 + 3 + 3) & ~3);  // leaf -> right separator rhs
  // the landmark x pose observation table: LDS when it fits
  unsigned short *obs;
  double *rec_ws = wsd; wsd += (size_t)S.M_max * REC;
  double *Aws = wsd; wsd += (size_t)(2 * S.L_max + 17) * (2 * S.L_max + 17);
  double *pws = wsd; wsd += (size_t)32 * (2 * S.L_max + 17);  // panels of the streamed sweep
  const size_t sys_bytes = (c_lds ? sweep_region_doubles(N) : c_reg ? (size_t)32 * N + 1280 : (size_t)1280) * 8;
  if (off + sys_bytes + up8((size_t)L * P * 2) + 32 <= (size_t)lds_bytes) {
    obs = reinterpret_cast<unsigned short *>(smem_raw + off); off += (up8((size_t)L * P * 2) + 31) & ~(size_t)31;
  } else {
    obs = reinterpret_cast<unsigned short *>(wsd);
  }
  // union region: during the chain solve the rhs rows of the separator poses (3 rows x ldx per separator), afterwards
  // the landmark system (packed lower triangle + its sweep panels) or the panels of the workspace variant
  const int nsep = (P + kSeg - 1) / kSeg;
  const size_t xs_bytes = (size_t)nsep * 3 * ldx * 8;
  double *U = reinterpret_cast<double *>(smem_raw + off);
  const bool xs_lds = off + max(sys_bytes, xs_bytes) <= (size_t)lds_bytes;
  off += xs_lds ? max(sys_bytes, xs_bytes) : sys_bytes;
  double *A = c_lds ? U : Aws;          // landmark system: packed lower triangle in LDS, or square (ld = N) in the workspace
  double *panels = c_lds ? nullptr : U;  // sweep panels of the workspace variant
  auto AT = [&](int i, int j) -> int { return c_lds ? i * (i + 1) / 2 + j : i * N + j; };
  // rhs rows of separator pose j (a multiple of kSeg), component r, column c:  SB[(j >> sshift) * 3 ldx + r ldx + c]
  double *SB = xs_lds ? U : X;
  const int sshift = xs_lds ? kSegLog : 0;
  auto srow = [&](int j) -> double * { return SB + (size_t)(j >> sshift) * 3 * ldx; };
  // per-factor records: LDS when they fit
  double *rec;
  // bytes of LDS from U on that are free once the union region's contents are dead (phases 7 and 10 stage operands there): up to
  // the factor records if those live in LDS, else to the end
  size_t u_free = (size_t)(smem_raw + off - reinterpret_cast<unsigned char *>(U));
  if (off + (size_t)M * REC * 8 <= (size_t)lds_bytes) {
    rec = reinterpret_cast<double *>(smem_raw + off); off += (size_t)M * REC * 8;
  } else {
    rec = rec_ws;
    u_free = (size_t)lds_bytes - (size_t)(reinterpret_cast<unsigned char *>(U) - smem_raw);
  }
  double *th_pose = S.th_pose + (size_t)inst * S.P_max * 4;
  double *d_pose = S.d_pose + (size_t)inst * S.P_max * 3;
  double *th_lm = S.th_lm + (size_t)inst * S.L_max * 2;
  double *d_lm = S.d_lm + (size_t)inst * S.L_max * 2;
  const int *meas_pose = S.meas_pose + (size_t)inst * S.M_max;
  const int *meas_lm = S.meas_lm + (size_t)inst * S.M_max;
  const double *meas_br = S.meas_br + (size_t)inst * S.M_max * 2;

  // ---- 1. relinearisation policy (gtsam ISAM2: relinearizeSkip 10, relinearizeThreshold 0.1); theta staged in LDS ----
  const bool relin = !refresh && (count % 10 == 0);
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
  for (int e = tid; e < L * P; e += kThreads) obs[e] = 0;
  for (int e = tid; e < MW * L; e += kThreads) lmask[e] = 0ull;
  for (int e = tid; e <= P; e += kThreads) mstart[e] = M;
  if (tid == 0) bad[0] = 0;
  __syncthreads();
  for (int m = tid; m < M; m += kThreads) {
    const int p = meas_pose[m], j = meas_lm[m];
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
  {
    // landmark blocks: the observations of a landmark are split over S3 lanes (see the landmark system below for why),
    // partial sums combined by a butterfly
    int S3 = 1;
    while (S3 < 64 && L * (S3 * 2) <= kThreads) S3 <<= 1;
    const int per_pass = kThreads / S3;
    for (int j0 = 0; j0 < L; j0 += per_pass) {
      const int j = j0 + tid / S3, s3 = tid & (S3 - 1);
      const bool work = j < L;
      double acc[5] = {0, 0, 0, 0, 0};  // Lambda_jj (a b d), J^T W e (g0 g1)
      auto visit = [&](const int (&ip)[4], int n) {
        double2 rv[4][3];  // Jl (4) and e (2) of up to four observations, loaded together
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double2 *r2 = reinterpret_cast<const double2 *>(rec + (size_t)REC * (obs[j * P + ip[u < n ? u : 0]] - 1) + 6);
          rv[u][0] = r2[0]; rv[u][1] = r2[1]; rv[u][2] = r2[2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u >= n) break;
          const double r6 = rv[u][0].x, r7 = rv[u][0].y, r8 = rv[u][1].x, r9 = rv[u][1].y, r10 = rv[u][2].x, r11 = rv[u][2].y;
          acc[0] += r6 * wb * r6 + r8 * wr * r8;
          acc[1] += r6 * wb * r7 + r8 * wr * r9;
          acc[2] += r7 * wb * r7 + r9 * wr * r9;
          acc[3] += r6 * wb * r10 + r8 * wr * r11;
          acc[4] += r7 * wb * r10 + r9 * wr * r11;
        }
      };
      if (work) {
        const unsigned long long *mk = lmask + MW * j;
        int ip[4] = {0, 0, 0, 0}, n = 0;
        if (S3 == 1) {
          MaskIter it(mk, MW);
          for (;;) {
            n = it.next4(ip);
            if (n == 0) break;
            visit(ip, n);
            if (n < 4) break;
          }
        } else {
          for (int i = s3; i < P; i += S3)
            if ((mk[i >> 6] >> (i & 63)) & 1ull) {
              ip[n++] = i;
              if (n == 4) {
                visit(ip, 4);
                n = 0;
              }
            }
          if (n) visit(ip, n);
        }
      }
      for (int o = S3 >> 1; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += __shfl_xor(acc[k], o);
      if (work && s3 == 0) {
        double *lb = lamb + 8 * j;
        lb[0] = acc[0]; lb[1] = acc[1]; lb[2] = acc[2];
        lb[6] = -acc[3]; lb[7] = -acc[4];  // eta_j
      }
    }
  }
  const int pose_t0 = 0;
  for (int i = (tid - pose_t0 + kThreads) % kThreads; i < P; i += kThreads) {
    double B[9], g[3], O[9];
    pose_block(S, inst, thp, S.odo + (size_t)inst * S.P_max * 4, rec, mstart, i, P, wb, wr, B, g, O);
    double *dd = Dd + 6 * i;
    dd[0] = B[0]; dd[1] = B[3]; dd[2] = B[6]; dd[3] = B[4]; dd[4] = B[7]; dd[5] = B[8];
    for (int r = 0; r < 3; ++r) X[(size_t)(3 * i + r) * ldx + np] = -g[r];  // eta_p: the rhs column of X
    if (i + 1 < P)
      for (int k = 0; k < 9; ++k) Al[9 * (i + 1) + k] = O[k];  // T_{i+1,i}
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
  __syncthreads();
  // ---- 5. chain solve, leaves: the 7 poses between two separator poses (multiples of 8) are eliminated in order, each
  //         against its successor and the segment's left separator l (fill).  One thread per segment:
  //         E_i = D_i^-1, GL_i = E_i T_{i,l}, GR_i = E_i T_{n,i}^T (n = i + 1);  D_l -= T_{i,l}^T GL_i,  D_n -= T_{n,i} GR_i,
  //         T_{n,l} = -T_{n,i} GL_i.  The two separators' diagonal updates go through scratch (GL / GR slots of l). ----
  auto mat_ab = [](const double *a, const double *b, double *o) {  // o = a b (3x3 row major)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
  };
  for (int g = tid; g < nsep; g += kThreads) {
    const int l = g << kSegLog, nk = min(kSeg - 1, P - 1 - l);
    double dl[6] = {0, 0, 0, 0, 0, 0}, dr[6] = {0, 0, 0, 0, 0, 0}, cl[9];
    for (int q = 0; q < 9; ++q) cl[q] = nk > 0 ? Al[9 * (l + 1) + q] : 0.0;
    for (int k = 1; k <= nk; ++k) {
      const int i = l + k, n = i + 1;
      const bool hn = n < P;
      double e[6], glf[9], grf[9], cn[9];
      if (!inv3s(Dd + 6 * i, e)) bad[0] = 1;
      for (int q = 0; q < 6; ++q) Dd[6 * i + q] = e[q];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          glf[r * 3 + c] = sym3(e, r, 0) * cl[c] + sym3(e, r, 1) * cl[3 + c] + sym3(e, r, 2) * cl[6 + c];
        }
      for (int q = 0; q < 9; ++q) cn[q] = hn ? Al[9 * n + q] : 0.0;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          grf[r * 3 + c] = sym3(e, r, 0) * cn[c * 3] + sym3(e, r, 1) * cn[c * 3 + 1] + sym3(e, r, 2) * cn[c * 3 + 2];
      for (int q = 0; q < 9; ++q) {
        GL[9 * i + q] = glf[q];
        GR[9 * i + q] = grf[q];
      }
      for (int r = 0, q = 0; r < 3; ++r)  // D_l -= T_{i,l}^T GL_i
        for (int c = r; c < 3; ++c, ++q) dl[q] -= cl[r] * glf[c] + cl[3 + r] * glf[3 + c] + cl[6 + r] * glf[6 + c];
      if (hn) {
        double t[9], nc[9];
        mat_ab(cn, grf, t);   // T_{n,i} E T_{n,i}^T
        mat_ab(cn, glf, nc);  // T_{n,i} E T_{i,l}
        if (k < kSeg - 1) {
          double *dn = Dd + 6 * n;
          dn[0] -= t[0]; dn[1] -= t[1]; dn[2] -= t[2]; dn[3] -= t[4]; dn[4] -= t[5]; dn[5] -= t[8];
          for (int q = 0; q < 9; ++q) cl[q] = -nc[q];
        } else {  // n is the right separator
          dr[0] = -t[0]; dr[1] = -t[1]; dr[2] = -t[2]; dr[3] = -t[4]; dr[4] = -t[5]; dr[5] = -t[8];
          for (int q = 0; q < 9; ++q) Al[9 * n + q] = -nc[q];  // T_{r,l}: the separator system's coupling
        }
      }
    }
    for (int q = 0; q < 6; ++q) {
      GL[9 * l + q] = dl[q];
      GR[9 * l + q] = dr[q];
    }
  }
  __syncthreads();
  for (int g = tid; g < nsep; g += kThreads) {
    const int l = g << kSegLog;
    for (int q = 0; q < 6; ++q) Dd[6 * l + q] += GL[9 * l + q] + (g > 0 ? GR[9 * (l - kSeg) + q] : 0.0);
  }
  DRLGX_PROF(S, 3);
  // ---- 6. rhs columns [B eta_p] through the leaves: one thread per (segment, column), the segment's rows in registers;
  //         y_i (the rhs at i's elimination) goes to X, the contributions to the two separators to the separator rows ----
  auto rhs_of = [&](int i, int c, double &b0, double &b1, double &b2) {  // column c of [B eta_p] at pose i
    if (c == np) {
      b0 = X[(size_t)(3 * i) * ldx + np]; b1 = X[(size_t)(3 * i + 1) * ldx + np]; b2 = X[(size_t)(3 * i + 2) * ldx + np];
    } else if (c > np) {  // unit column c - np - 1 of the newest pose
      const bool at = i == P - 1;
      b0 = (at && c == np + 1) ? 1.0 : 0.0; b1 = (at && c == np + 2) ? 1.0 : 0.0; b2 = (at && c == np + 3) ? 1.0 : 0.0;
    } else {
      const int m1 = obs[(c >> 1) * P + i], a = c & 1;
      const double *bm = rec + (size_t)REC * (m1 ? m1 - 1 : 0);
      b0 = m1 ? bm[a] : 0.0; b1 = m1 ? bm[2 + a] : 0.0; b2 = m1 ? bm[4 + a] : 0.0;
    }
  };
  for (int e = tid; e < nsep * ncx; e += kThreads) {
    const int g = e / ncx, c = e - g * ncx;
    const int l = g << kSegLog, nk = min(kSeg - 1, P - 1 - l);
    double bk[kSeg][3];
#pragma unroll
    for (int k = 0; k < kSeg; ++k) {
      bk[k][0] = bk[k][1] = bk[k][2] = 0.0;
      if (k <= nk) rhs_of(l + k, c, bk[k][0], bk[k][1], bk[k][2]);
    }
    double d0 = bk[0][0], d1 = bk[0][1], d2 = bk[0][2];  // separator l: own value + the leaf's contributions
    double y0 = 0, y1 = 0, y2 = 0;
#pragma unroll
    for (int k = 1; k < kSeg; ++k) {
      if (k <= nk) {
        const int i = l + k;
        if (k > 1) {  // b_i -= GR_{i-1}^T y_{i-1}
          const double *g2 = GR + 9 * (i - 1);
          bk[k][0] -= g2[0] * y0 + g2[3] * y1 + g2[6] * y2;
          bk[k][1] -= g2[1] * y0 + g2[4] * y1 + g2[7] * y2;
          bk[k][2] -= g2[2] * y0 + g2[5] * y1 + g2[8] * y2;
        }
        y0 = bk[k][0]; y1 = bk[k][1]; y2 = bk[k][2];
        X[(size_t)(3 * i) * ldx + c] = y0;
        X[(size_t)(3 * i + 1) * ldx + c] = y1;
        X[(size_t)(3 * i + 2) * ldx + c] = y2;
        const double *g1 = GL + 9 * i;  // b_l -= GL_i^T y_i
        d0 -= g1[0] * y0 + g1[3] * y1 + g1[6] * y2;
        d1 -= g1[1] * y0 + g1[4] * y1 + g1[7] * y2;
        d2 -= g1[2] * y0 + g1[5] * y1 + g1[8] * y2;
      }
    }
    double *sl = srow(l) + c;
    sl[0] = d0; sl[ldx] = d1; sl[2 * ldx] = d2;
    if (l + kSeg < P) {  // contribution to the right separator: added by the next loop (one writer per element)
      const double *g2 = GR + 9 * (l + kSeg - 1);
      double *pr = sepR + (size_t)g * 3 * ldx + c;
      pr[0] = -(g2[0] * y0 + g2[3] * y1 + g2[6] * y2);
      pr[ldx] = -(g2[1] * y0 + g2[4] * y1 + g2[7] * y2);
      pr[2 * ldx] = -(g2[2] * y0 + g2[5] * y1 + g2[8] * y2);
    }
  }
  __syncthreads();
  for (int e = tid; e < (nsep - 1) * ncx; e += kThreads) {  // separator g + 1 += its left leaf's contribution
    const int g = e / ncx, c = e - g * ncx;
    const double *pr = sepR + (size_t)g * 3 * ldx + c;
    double *sr = srow((g + 1) << kSegLog) + c;
    sr[0] += pr[0]; sr[ldx] += pr[ldx]; sr[2 * ldx] += pr[2 * ldx];
  }
  __syncthreads();
  DRLGX_PROF(S, 4);
  // ---- ... cyclic reduction of the separator system (strides 8, 16, ...), its rhs rows in LDS: down ----
  for (int s = kSeg; s < P; s <<= 1) {
    // eliminated poses i = s, 3s, ...: E_i, GL_i, GR_i
    for (int k = tid; (2 * k + 1) * s < P; k += kThreads) {
      const int i = (2 * k + 1) * s;
      double e[6];
      if (!inv3s(Dd + 6 * i, e)) bad[0] = 1;
      for (int q = 0; q < 6; ++q) Dd[6 * i + q] = e[q];
      const double *ai = Al + 9 * i;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) GL[9 * i + r * 3 + c] = sym3(e, r, 0) * ai[c] + sym3(e, r, 1) * ai[3 + c] + sym3(e, r, 2) * ai[6 + c];
      const bool hr = i + s < P;
      const double *ar = Al + 9 * (hr ? i + s : i);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)  // E A_{i+s}^T
          GR[9 * i + r * 3 + c] = hr ? sym3(e, r, 0) * ar[c * 3] + sym3(e, r, 1) * ar[c * 3 + 1] + sym3(e, r, 2) * ar[c * 3 + 2] : 0.0;
    }
    __syncthreads();
    // surviving poses j = 0, 2s, 4s, ...: diagonal block, coupling to j - 2s
    for (int k = tid; 2 * k * s < P; k += kThreads) {
      const int j = 2 * k * s;
      double d[6];
      for (int q = 0; q < 6; ++q) d[q] = Dd[6 * j + q];
      if (j >= s) {  // left eliminated neighbour i1 = j - s:  D_j -= A_j GR_i1,  A_j <- -A_j GL_i1
        const double *aj = Al + 9 * j, *gr = GR + 9 * (j - s), *gl = GL + 9 * (j - s);
        double t[9], na[9];
        mat_ab(aj, gr, t);
        mat_ab(aj, gl, na);
        d[0] -= t[0]; d[1] -= t[1]; d[2] -= t[2]; d[3] -= t[4]; d[4] -= t[5]; d[5] -= t[8];
        for (int q = 0; q < 9; ++q) Al[9 * j + q] = -na[q];
      }
      if (j + s < P) {  // right eliminated neighbour i2 = j + s:  D_j -= A_i2^T GL_i2
        const double *a2 = Al + 9 * (j + s), *gl = GL + 9 * (j + s);
        for (int r = 0, q = 0; r < 3; ++r)
          for (int c = r; c < 3; ++c, ++q) d[q] -= a2[r] * gl[c] + a2[3 + r] * gl[3 + c] + a2[6 + r] * gl[6 + c];
      }
      for (int q = 0; q < 6; ++q) Dd[6 * j + q] = d[q];
    }
    // ... and their rhs rows: b_j -= GR_i1^T b_i1 + GL_i2^T b_i2
    {
      const int nsurv = (P - 1) / (2 * s) + 1;
      for (int e = tid; e < nsurv * ncx; e += kThreads) {
        const int k = e / ncx, c = e - k * ncx, j = 2 * k * s;
        double *bj = srow(j) + c;
        double b0 = bj[0], b1 = bj[ldx], b2 = bj[2 * ldx];
        if (j >= s) {
          const double *g = GR + 9 * (j - s), *bi = srow(j - s) + c;
          const double v0 = bi[0], v1 = bi[ldx], v2 = bi[2 * ldx];
          b0 -= g[0] * v0 + g[3] * v1 + g[6] * v2;
          b1 -= g[1] * v0 + g[4] * v1 + g[7] * v2;
          b2 -= g[2] * v0 + g[5] * v1 + g[8] * v2;
        }
        if (j + s < P) {
          const double *g = GL + 9 * (j + s), *bi = srow(j + s) + c;
          const double v0 = bi[0], v1 = bi[ldx], v2 = bi[2 * ldx];
          b0 -= g[0] * v0 + g[3] * v1 + g[6] * v2;
          b1 -= g[1] * v0 + g[4] * v1 + g[7] * v2;
          b2 -= g[2] * v0 + g[5] * v1 + g[8] * v2;
        }
        bj[0] = b0; bj[ldx] = b1; bj[2 * ldx] = b2;
      }
    }
    __syncthreads();
  }
  // root (pose 0): E_0, x_0 = E_0 b_0, (T^-1)_00 = E_0
  if (tid == 0) {
    double e[6];
    if (!inv3s(Dd, e)) bad[0] = 1;
    for (int q = 0; q < 6; ++q) {
      Dd[q] = e[q];
      Ti[q] = e[q];
    }
  }
  __syncthreads();
  for (int c = tid; c < ncx; c += kThreads) {
    double *b = srow(0) + c;
    const double v0 = b[0], v1 = b[ldx], v2 = b[2 * ldx];
    b[0] = Dd[0] * v0 + Dd[1] * v1 + Dd[2] * v2;
    b[ldx] = Dd[1] * v0 + Dd[3] * v1 + Dd[4] * v2;
    b[2 * ldx] = Dd[2] * v0 + Dd[4] * v1 + Dd[5] * v2;
  }
  __syncthreads();
  // ---- ... and back up: separator solutions and the selected inverse, level by level ----
  auto ld_sym = [](const double *t, double *o) {
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[1]; o[4] = t[3]; o[5] = t[4]; o[6] = t[2]; o[7] = t[4]; o[8] = t[5];
  };
  // S_lr (row l, column r) of two poses adjacent at stride 2s: stored by the one that is eliminated at that stride
  auto cross_lr = [&](int l, int r, int s, double *slr) {
    if ((l / (2 * s)) & 1) {  // l: its right neighbour at stride 2s is r:  Sr[l] = S_{r,l}
      const double *t = Sr + 9 * l;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) slr[a * 3 + b] = t[b * 3 + a];
    } else {                  // r: its left neighbour at stride 2s is l:   Sl[r] = S_{l,r}
      const double *t = Sl + 9 * r;
      for (int q = 0; q < 9; ++q) slr[q] = t[q];
    }
  };
  // Takahashi step of pose i with neighbours (l, n):  S_{l,i} = -(S_ll GL^T + S_ln GR^T),  S_{n,i} = -(S_nl GL^T + S_nn GR^T),
  // S_ii = E_i - GL S_{l,i} - GR S_{n,i}
  auto takahashi = [&](int i, const double *sll, const double *snn, const double *sln, double *sli, double *sni, double *sii) {
    const double *gl = GL + 9 * i, *gr = GR + 9 * i, *e = Dd + 6 * i;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double v = 0, w = 0;
        for (int q = 0; q < 3; ++q) {
          v += sll[a * 3 + q] * gl[b * 3 + q] + sln[a * 3 + q] * gr[b * 3 + q];
          w += sln[q * 3 + a] * gl[b * 3 + q] + snn[a * 3 + q] * gr[b * 3 + q];
        }
        sli[a * 3 + b] = -v;
        sni[a * 3 + b] = -w;
      }
    for (int a = 0, q = 0; a < 3; ++a)
      for (int b = a; b < 3; ++b, ++q) {
        double v = e[q];
        for (int t = 0; t < 3; ++t) v -= gl[a * 3 + t] * sli[t * 3 + b] + gr[a * 3 + t] * sni[t * 3 + b];
        sii[q] = v;
      }
  };
  {
    int s_top = kSeg;
    while (2 * s_top < P) s_top <<= 1;
    for (int s = (P > kSeg) ? s_top : 0; s >= kSeg; s >>= 1) {
      const int nel = (P - 1 - s) / (2 * s) + 1;  // eliminated poses of this level: s, 3s, ... < P
      for (int k = tid; full && k < nel; k += kThreads) {
        const int i = (2 * k + 1) * s, l = i - s, r = i + s;
        const bool hr = r < P;
        double sll[9], srr[9], slr[9], sli[9], sri[9];
        ld_sym(Ti + 6 * l, sll);
        for (int q = 0; q < 9; ++q) srr[q] = slr[q] = 0.0;
        if (hr) {
          ld_sym(Ti + 6 * r, srr);
          cross_lr(l, r, s, slr);
        }
        takahashi(i, sll, srr, slr, sli, sri, Ti + 6 * i);
        for (int q = 0; q < 9; ++q) {
          Sl[9 * i + q] = sli[q];
          Sr[9 * i + q] = sri[q];
        }
      }
      // solutions x_i = E_i b_i - GL_i x_l - GR_i x_r, one thread per (pose, column)
      for (int e = tid; e < nel * ncx; e += kThreads) {
        const int k = e / ncx, c = e - k * ncx, i = (2 * k + 1) * s;
        double *bi = srow(i) + c;
        const double *ei = Dd + 6 * i, *gl = GL + 9 * i, *gr = GR + 9 * i;
        const double v0 = bi[0], v1 = bi[ldx], v2 = bi[2 * ldx];
        double x0 = ei[0] * v0 + ei[1] * v1 + ei[2] * v2;
        double x1 = ei[1] * v0 + ei[3] * v1 + ei[4] * v2;
        double x2 = ei[2] * v0 + ei[4] * v1 + ei[5] * v2;
        {
          const double *xl = srow(i - s) + c;
          const double l0 = xl[0], l1 = xl[ldx], l2 = xl[2 * ldx];
          x0 -= gl[0] * l0 + gl[1] * l1 + gl[2] * l2;
          x1 -= gl[3] * l0 + gl[4] * l1 + gl[5] * l2;
          x2 -= gl[6] * l0 + gl[7] * l1 + gl[8] * l2;
        }
        if (i + s < P) {
          const double *xr = srow(i + s) + c;
          const double r0 = xr[0], r1 = xr[ldx], r2 = xr[2 * ldx];
          x0 -= gr[0] * r0 + gr[1] * r1 + gr[2] * r2;
          x1 -= gr[3] * r0 + gr[4] * r1 + gr[5] * r2;
          x2 -= gr[6] * r0 + gr[7] * r1 + gr[8] * r2;
        }
        bi[0] = x0; bi[ldx] = x1; bi[2 * ldx] = x2;
      }
      __syncthreads();
    }
  }
  DRLGX_PROF(S, 5);
  // ---- ... leaves: selected inverse (one thread per segment, last interior pose first) and solutions (one thread per
  //         (segment, column): x_i = E_i y_i - GL_i x_l - GR_i x_{i+1}); the separators' solutions go to X too ----
  for (int g = tid; full && g < nsep; g += kThreads) {
    const int l = g << kSegLog, nk = min(kSeg - 1, P - 1 - l), r = l + kSeg;
    if (nk > 0) {
      double sll[9], snn[9], sln[9], sli[9], sni[9], sii[6];
      ld_sym(Ti + 6 * l, sll);
      for (int q = 0; q < 9; ++q) snn[q] = sln[q] = 0.0;
      if (r < P) {
        ld_sym(Ti + 6 * r, snn);
        cross_lr(l, r, kSeg >> 1, sln);
      }
      for (int k = nk; k >= 1; --k) {
        const int i = l + k;
        takahashi(i, sll, snn, sln, sli, sni, sii);
        for (int q = 0; q < 6; ++q) Ti[6 * i + q] = sii[q];
        ld_sym(sii, snn);                           // the next pose down has this one as its successor
        for (int q = 0; q < 9; ++q) sln[q] = sli[q];
      }
    }
  }
  for (int e = tid; e < nsep * ncx; e += kThreads) {
    const int g = e / ncx, c = e - g * ncx;
    const int l = g << kSegLog, nk = min(kSeg - 1, P - 1 - l);
    double yk[kSeg][3];
#pragma unroll
    for (int k = 1; k < kSeg; ++k) {
      const bool ok = k <= nk;
      const double *yp = X + (size_t)(3 * (ok ? l + k : l)) * ldx + c;
      yk[k][0] = ok ? yp[0] : 0.0; yk[k][1] = ok ? yp[ldx] : 0.0; yk[k][2] = ok ? yp[2 * ldx] : 0.0;
    }
    const double *xlp = srow(l) + c;
    const double xl0 = xlp[0], xl1 = xlp[ldx], xl2 = xlp[2 * ldx];
    double n0 = 0, n1 = 0, n2 = 0;  // solution of the successor (the right separator for the last interior pose)
    if (l + kSeg < P) {
      const double *xrp = srow(l + kSeg) + c;
      n0 = xrp[0]; n1 = xrp[ldx]; n2 = xrp[2 * ldx];
    }
    X[(size_t)(3 * l) * ldx + c] = xl0;
    X[(size_t)(3 * l + 1) * ldx + c] = xl1;
    X[(size_t)(3 * l + 2) * ldx + c] = xl2;
#pragma unroll
    for (int k = kSeg - 1; k >= 1; --k) {
      if (k <= nk) {
        const int i = l + k;
        const double *ei = Dd + 6 * i, *gl = GL + 9 * i, *gr = GR + 9 * i;
        const double v0 = yk[k][0], v1 = yk[k][1], v2 = yk[k][2];
        const double x0 = ei[0] * v0 + ei[1] * v1 + ei[2] * v2 - (gl[0] * xl0 + gl[1] * xl1 + gl[2] * xl2) - (gr[0] * n0 + gr[1] * n1 + gr[2] * n2);
        const double x1 = ei[1] * v0 + ei[3] * v1 + ei[4] * v2 - (gl[3] * xl0 + gl[4] * xl1 + gl[5] * xl2) - (gr[3] * n0 + gr[4] * n1 + gr[5] * n2);
        const double x2 = ei[2] * v0 + ei[4] * v1 + ei[5] * v2 - (gl[6] * xl0 + gl[7] * xl1 + gl[8] * xl2) - (gr[6] * n0 + gr[7] * n1 + gr[8] * n2);
        X[(size_t)(3 * i) * ldx + c] = x0;
        X[(size_t)(3 * i + 1) * ldx + c] = x1;
        X[(size_t)(3 * i + 2) * ldx + c] = x2;
        n0 = x0; n1 = x1; n2 = x2;
      }
    }
  }
  __syncthreads();
  {
    const size_t nA = c_lds ? (size_t)N * (N + 1) / 2 : (size_t)N * N;
    for (size_t e = tid; e < nA; e += kThreads) A[e] = 0.0;  // (the region held the separator rows until here)
  }
  __syncthreads();
  DRLGX_PROF(S, 10);
  // ---- 7. landmark system [C r]: rows 2j, 2j+1 at column c (lower triangle + the rhs column) ----
  {
    // one work item per (landmark, 4 columns).  Its observing poses are split over S_ lanes (as many as the workgroup
    // has to spare: a sparse world has a handful of landmarks, each seen from dozens of poses - one thread per item would
    // walk them in dependent rounds of L2 latency); every lane visits its poses four at a time with all their loads (32-byte
    // rows of X, the factor's B block) issued before the first use, and the lanes' partial sums are combined by a
    // butterfly (a fixed tree: deterministic)
    const int nq = ldx >> 2, q_rhs = np >> 2;
    const int items = L * nq;
    int S_ = 1;
    while (S_ < 64 && items * (S_ * 2) <= kThreads) S_ <<= 1;
    const int per_pass = kThreads / S_;
    // (Round 6 measured X walked in chunks of 32 columns staged in LDS for the wide systems - every row of X is wanted ~22 times at
    // BASELINE config 5 scale, 13 MB of 32-byte pieces per instance -: 309 against 286 us.  What an item waits for is not X but the
    // chain obs -> factor record -> its B block, one L2 round trip per four visits; the records (120 KB) do not fit beside a chunk.)
    for (int e0 = 0; e0 < items; e0 += per_pass) {
      const int e = e0 + tid / S_, s = tid & (S_ - 1);
      const int j = e < items ? e / nq : 0, q = e < items ? e - j * nq : 0, c0 = 4 * q;
      const bool work = e < items && !(c0 > 2 * j + 1 && q != q_rhs);
      double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
      auto visit = [&](const int (&ip)[4], int n) {
        double2 xv[4][3][2];
        double bm[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = (u < n) ? ip[u] : ip[0];
          const double *xr = X + (size_t)(3 * i) * ldx + c0;
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            xv[u][r][0] = *reinterpret_cast<const double2 *>(xr + (size_t)r * ldx);
            xv[u][r][1] = *reinterpret_cast<const double2 *>(xr + (size_t)r * ldx + 2);
          }
          const double *bp = rec + (size_t)REC * (obs[j * P + i] - 1);
#pragma unroll
          for (int k = 0; k < 6; ++k) bm[u][k] = bp[k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u >= n) break;
          const double x0[4] = {xv[u][0][0].x, xv[u][0][0].y, xv[u][0][1].x, xv[u][0][1].y};
          const double x1[4] = {xv[u][1][0].x, xv[u][1][0].y, xv[u][1][1].x, xv[u][1][1].y};
          const double x2[4] = {xv[u][2][0].x, xv[u][2][0].y, xv[u][2][1].x, xv[u][2][1].y};
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            a0[cc] += bm[u][0] * x0[cc] + bm[u][2] * x1[cc] + bm[u][4] * x2[cc];
            a1[cc] += bm[u][1] * x0[cc] + bm[u][3] * x1[cc] + bm[u][5] * x2[cc];
          }
        }
      };
      if (work) {
        const unsigned long long *mk = lmask + MW * j;
        int ip[4] = {0, 0, 0, 0}, n = 0;
        if (S_ == 1) {  // every observing pose, ascending
          MaskIter it(mk, MW);
          for (;;) {
            n = it.next4(ip);
            if (n == 0) break;
            visit(ip, n);
            if (n < 4) break;
          }
        } else {        // this lane's share: the poses i = s (mod S_)
          for (int i = s; i < P; i += S_)
            if ((mk[i >> 6] >> (i & 63)) & 1ull) {
              ip[n++] = i;
              if (n == 4) {
                visit(ip, 4);
                n = 0;
              }
            }
          if (n) visit(ip, n);
        }
      }
      for (int o = S_ >> 1; o > 0; o >>= 1) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          a0[cc] += __shfl_xor(a0[cc], o);
          a1[cc] += __shfl_xor(a1[cc], o);
        }
      }
      if (!work || s != 0) continue;
      const double *lb = lamb + 8 * j;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = c0 + cc;
        if (c == np) {
          A[AT(np, 2 * j)] = lb[6] - a0[cc];
          A[AT(np, 2 * j + 1)] = lb[7] - a1[cc];
        } else if (c < np && c <= 2 * j + 1) {
          const bool own = (c >> 1) == j;
          const double l0 = own ? ((c & 1) ? lb[1] : lb[0]) : 0.0, l1 = own ? ((c & 1) ? lb[2] : lb[1]) : 0.0;
          if (c <= 2 * j) A[AT(2 * j, c)] = l0 - a0[cc];
          A[AT(2 * j + 1, c)] = l1 - a1[cc];
        }
      }
    }
  }
  __syncthreads();
  DRLGX_PROF(S, 6);
  // ---- 8. sweep: A <- -C^-1 (lower triangle), row np <- delta_l ----
  if (c_lds)
    sweep_packed_fast<kFastTilesArrow>(S, A, np, N, Tn, bad, tid);
  else if constexpr (NTW > 0) {
    if (c_reg) sweep_regtiles<false, NTW>(A, panels, np, N, Tn, ntiles, bad, tid);
    else sweep_streamed(A, pws, panels, np, N, Tn, bad, tid);
  }
  __syncthreads();
  DRLGX_PROF(S, 7);
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
    if (!full) continue;
    const double c00 = -A[AT(2 * j, 2 * j)], cs = -A[AT(2 * j + 1, 2 * j)], c11 = -A[AT(2 * j + 1, 2 * j + 1)];
    lm_tr[j] = c00 + c11;
    const double id = 1.0 / (c00 * c11 - cs * cs);  // marginalCovariance(l).inverse() (SLAM2D.cpp:417)
    lm_info[3 * j] = c11 * id;
    lm_info[3 * j + 1] = -cs * id;
    lm_info[3 * j + 2] = c00 * id;
  }
  DRLGX_PROF(S, 8);
  // ---- 10. pose outputs: delta_p = x_eta - X_B delta_l; Sigma_ii = (T^-1)_ii + X_i C^-1 X_i^T.
  //      Z = X_B [-C^-1 | delta_l] on the fp64 matrix cores (one wave per 16 rows of X, tiles of 16 columns, the
  //      A operand = 32-byte rows of X from L2 / HBM, the B operand from the swept system), then per row the products
  //      with the three X rows of its pose, reduced over the 16 lanes of an accumulator row in a fixed order ----
  double *est_pose = S.est_pose + (size_t)inst * S.P_max * 4;
  double *pose_info = S.pose_info + (size_t)inst * S.P_max * 6;
  double *pose_tr = S.pose_tr + (size_t)inst * S.P_max;
  double *pan_pose = mk_panel ? S.jc + (size_t)inst * S.jc_stride : nullptr;  // the covariance panel's pose rows (k_inc.hip)
  double *Sc = Sl;  // [3P][3]: rows of X_i (-C^-1) X_i^T   (the Takahashi cross blocks are dead by now)
  double *dz = Sr;  // [3P]:    X_B delta_l
  bool pn_done = false;  // the panel's columns of the newest pose were formed with the tiles of Z
  if (!full) {
    // estimates only: dz = X_B delta_l, one 16-lane row per row of X
    const int sub = tid & 15, grp = tid >> 4, ngrp = kThreads / 16;
    for (int row = grp; row < 3 * P; row += ngrp) {
      const double *xr = X + (size_t)row * ldx;
      double v = 0;
      for (int c = sub; c < np; c += 16) v += xr[c] * A[AT(np, c)];
      v += __shfl_xor(v, 8, 16);
      v += __shfl_xor(v, 4, 16);
      v += __shfl_xor(v, 2, 16);
      v += __shfl_xor(v, 1, 16);
      if (sub == 0) dz[row] = v;
    }
  } else if (c_lds && np <= 12) {
    // a handful of landmarks: one thread per pose does its three rows of Z = X_B [-C^-1 | delta_l] and the 3 x 3 product
    // with X_i^T on the vector units (at most 3 * 13 * 12 + 108 multiply-adds) - four rounds of matrix-core tiles with
    // their operand loads would cost more than that
    for (int i = tid; i < P; i += kThreads) {
      double xr[3][12];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 12; ++k) xr[r][k] = k < np ? X[(size_t)(3 * i + r) * ldx + k] : 0.0;
      double sc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, dzr[3] = {0, 0, 0};
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        if (c < np) {
          double z[3] = {0, 0, 0};
#pragma unroll
          for (int k = 0; k < 12; ++k)
            if (k < np) {
              const double m = A[AT(max(k, c), min(k, c))];
              z[0] += xr[0][k] * m; z[1] += xr[1][k] * m; z[2] += xr[2][k] * m;
            }
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            sc[r][0] += z[r] * xr[0][c]; sc[r][1] += z[r] * xr[1][c]; sc[r][2] += z[r] * xr[2][c];
            if (mk_panel) pan_pose[(size_t)(3 * i + r) * S.jc_ld + 3 + c] = z[r];  // Sigma[pose i][landmark column c]
          }
          const double dl = A[AT(np, c)];
          dzr[0] += xr[0][c] * dl; dzr[1] += xr[1][c] * dl; dzr[2] += xr[2][c] * dl;
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        dz[3 * i + r] = dzr[r];
        Sc[9 * i + 3 * r] = sc[r][0]; Sc[9 * i + 3 * r + 1] = sc[r][1]; Sc[9 * i + 3 * r + 2] = sc[r][2];
      }
    }
  } else {
    if (!c_lds) {
      // the swept system lives in the workspace as a square matrix of which the lower triangle is valid: mirror it, so that
      // the B operand below is read along rows (16 lanes = 128 contiguous bytes) whichever side of the diagonal a tile is on
      for (int e = tid; e < np * np; e += kThreads) {
        const int i = e / np, j = e - i * np;
        if (j > i) A[(size_t)i * N + j] = A[(size_t)j * N + i];
      }
      __syncthreads();
    }
    DRLGX_PROF(S, 100);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lc = lane & 15, lr = lane >> 4;
    const int nrows = 3 * P, nrt = (nrows + 15) / 16, nK = (np + 15) / 16;
    const int pn = P - 1;
    pn_done = mk_panel;  // (the tile path below also forms the panel's columns of the newest pose)
    constexpr int kPanel = 8;
    // epilogue of one 16 x 16 tile of Z (accumulator `acc`, row tile I, column tile J): products with the three X rows of each
    // accumulator row's pose (sp) and - for the covariance panel - with the three X rows of the newest pose (sq:
    // Sigma[i][pn] = (T^-1)[i][pn] + X_i C^-1 X_pn^T = X[., np+1..np+3] - Sigma_pl[i] X_pn^T; a separate pass used to form these sums
    // with one thread per (row, column): 214 dependent trips each, 85 us at BASELINE config 5 scale), and the delta_l column
    auto epilogue = [&](int I, const v4d &acc, int J, double (&sp)[4][3], double (&sq)[4][3]) {
      const int c = 16 * J + lc;
      double xe[4][3], xn[3] = {0.0, 0.0, 0.0};
      if (mk_panel && c < np) {
        const double *xb = X + (size_t)(3 * pn) * ldx + c;
        xn[0] = xb[0]; xn[1] = xb[ldx]; xn[2] = xb[2 * ldx];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + lr + 4 * r;
        const bool ok = row < nrows && c < np;
        const double *xb = X + (size_t)(ok ? 3 * (row / 3) : 0) * ldx + (ok ? c : 0);
        xe[r][0] = ok ? xb[0] : 0.0;
        xe[r][1] = ok ? xb[ldx] : 0.0;
        xe[r][2] = ok ? xb[2 * ldx] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + lr + 4 * r;
        sp[r][0] += acc[r] * xe[r][0];  // (zeros outside the matrix)
        sp[r][1] += acc[r] * xe[r][1];
        sp[r][2] += acc[r] * xe[r][2];
        if (c < np) {
          sq[r][0] += acc[r] * xn[0];
          sq[r][1] += acc[r] * xn[1];
          sq[r][2] += acc[r] * xn[2];
        }
        if (row < nrows && c == np) dz[row] = acc[r];
        if (mk_panel && row < nrows && c < np) pan_pose[(size_t)row * S.jc_ld + 3 + c] = acc[r];  // Sigma_pl = X_B (-C^-1)
      }
    };
    // the sums over the 16 lanes of an accumulator row, in a fixed order: DPP row shifts (1, 2, 4, 8), the row's sum in its lane 15
    // (butterflies through ds_bpermute cost 8 us per call at BASELINE config 5 scale: 96 LDS round trips)
    auto rowsum = [&](double (&sv)[4][3]) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b3 = 0; b3 < 3; ++b3) {
          double v = sv[r][b3];
          v = dpp_add_f64<0x111>(v);
          v = dpp_add_f64<0x112>(v);
          v = dpp_add_f64<0x114>(v);
          v = dpp_add_f64<0x118>(v);
          sv[r][b3] = v;
        }
    };
    if (nK <= kPanel) {
      for (int I = wave; I < nrt; I += kWaves) {
        const int arow = 16 * I + lc;
        const bool arow_ok = arow < nrows;
        const double *xa = X + (size_t)(arow_ok ? arow : 0) * ldx;
        // the A operand of the whole tile row (its 16 x np panel of X) is loaded once: every load is in flight before the
        // first matrix instruction (up to 8 column chunks = 127 landmark columns: the LDS-resident systems)
        double pan[kPanel][4];
#pragma unroll
        for (int K = 0; K < kPanel; ++K)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int k = 16 * K + 4 * lr + t;
            pan[K][t] = (k < np && arow_ok) ? xa[k] : 0.0;
          }
        double sp[4][3], sq[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) sp[r][0] = sp[r][1] = sp[r][2] = sq[r][0] = sq[r][1] = sq[r][2] = 0.0;
        for (int J = 0; J < Tn; ++J) {
          const int c = 16 * J + lc;
          v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int K = 0; K < kPanel; ++K) {
            if (K < nK) {
              double bv[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int k = 16 * K + 4 * lr + t;
                bv[t] = (k < np && c <= np) ? A[AT(max(k, c), min(k, c))] : 0.0;
              }
              acc = mfma4(pan[K], bv, acc);
            }
          }
          epilogue(I, acc, J, sp, sq);
        }
        rowsum(sp);
        if (mk_panel) rowsum(sq);
        if (lc == 15) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + lr + 4 * r;
            if (row < nrows) {
#pragma unroll
              for (int b3 = 0; b3 < 3; ++b3) {
                Sc[3 * row + b3] = sp[r][b3];
                if (mk_panel) pan_pose[(size_t)row * S.jc_ld + b3] = X[(size_t)row * ldx + np + 1 + b3] - sq[r][b3];
              }
            }
          }
        }
      }
    } else {
      // Wide systems (workspace variant, mirrored above).  Round 6: the B operand - a group of column tiles of [-C^-1 | delta_l], all
      // K steps - is STAGED IN LDS for the whole workgroup (the sweep's panels are dead by now) as operand images (a lane's four K
      // entries contiguous: conflict-free 16-byte reads), and every wave walks its row tiles against it with the A operand - 32-byte
      // rows of X - requested one K step ahead.  Before, every wave fetched BOTH operands from the workspace per (row tile, column
      // group, K step): the B operand alone 21 row tiles x 385 KB = 8 MB per instance, 2 GB per 256-instance launch out of the
      // Infinity Cache - this product ran at a quarter of the fp64 matrix rate (470 of the relinearising update's 1 580 us at
      // BASELINE config 5 scale).  The row sums are accumulated per column group (the same lane adds to the same entries).
      constexpr int kZG = 3;
      // as many column tiles per group as the LDS behind the pose tables holds (one tile = nK x 2 KB); none: one tile, staged in
      // the workspace's sweep panels (dead as well) - the L2 then serves what the LDS would
      const int zfit = (int)(u_free / ((size_t)nK * 2048));
      const int zg = zfit >= 1 ? min(kZG, zfit) : 1;
      double *Bs = zfit >= 1 ? U : pws;  // [zg][nK][4 lr][16 lc][4 t]
      // the row sums are accumulated per column group in LDS - the chain factors' tables (9 P doubles each) are dead since the
      // selected inverse - and go to the workspace / the panel at the end (as read-modify-writes of the workspace they cost 8 us
      // per (row tile, column group): a round trip to L2 each)
      double *ScL = Al, *SqL = GL;  // [3P][3]
      for (int e = tid; e < nrows * 3; e += kThreads) ScL[e] = SqL[e] = 0.0;
      const bool zprof = S.prof && blockIdx.x == S.prof_block && tid == 0;
      long long zt_stage = 0, zt_k = 0, zt_epi = 0, zt_sum = 0;
      for (int J0 = 0; J0 < Tn; J0 += zg) {
        __syncthreads();  // (the previous group's images are read; the first time: the initialisation above)
        const long long zt0 = zprof ? wall_clock64() : 0;
        for (int e0 = tid; e0 < nK * 16 * zg * 16; e0 += 4 * kThreads) {  // (four loads in flight per thread)
          double v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * kThreads;
            const int k = e / (zg * 16), cc = e - k * (zg * 16), g = cc >> 4, c = 16 * J0 + cc;
            const bool in = e < nK * 16 * zg * 16 && k < np && c <= np && J0 + g < Tn;
            v[u] = in ? A[c == np ? (size_t)np * N + k : (size_t)k * N + c] : 0.0;  // delta_l row / mirrored -C^-1
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * kThreads;
            const int k = e / (zg * 16), cc = e - k * (zg * 16), g = cc >> 4;
            if (e < nK * 16 * zg * 16) Bs[((((size_t)g * nK + (k >> 4)) * 4 + ((k >> 2) & 3)) * 16 + (cc & 15)) * 4 + (k & 3)] = v[u];
          }
        }
        __syncthreads();
        if (zprof) zt_stage += wall_clock64() - zt0;
        for (int I = wave; I < nrt; I += kWaves) {
          const long long zt1 = zprof ? wall_clock64() : 0;
          const int arow = 16 * I + lc;
          const bool arow_ok = arow < nrows;
          const double *xa = X + (size_t)(arow_ok ? arow : 0) * ldx + 4 * lr;
          double sp[4][3], sq[4][3];
#pragma unroll
          for (int r = 0; r < 4; ++r) sp[r][0] = sp[r][1] = sp[r][2] = sq[r][0] = sq[r][1] = sq[r][2] = 0.0;
          v4d acc[kZG];
#pragma unroll
          for (int g = 0; g < kZG; ++g) acc[g] = v4d{0.0, 0.0, 0.0, 0.0};
          // (the A operand comes from L2 / the Infinity Cache, 1-2 us away: requested FOUR K steps ahead - one step's twelve matrix
          // instructions last 0.3 us)
          auto load_a4 = [&](int K0, double (&av)[4][4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int t = 0; t < 4; ++t) av[u][t] = (16 * (K0 + u) + 4 * lr + t < np && arow_ok) ? xa[16 * (K0 + u) + t] : 0.0;
          };
          double avn[4][4];
          load_a4(0, avn);
          for (int K0 = 0; K0 < nK; K0 += 4) {
            double av[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int t = 0; t < 4; ++t) av[u][t] = avn[u][t];
            if (K0 + 4 < nK) load_a4(K0 + 4, avn);
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (K0 + u < nK) {
#pragma unroll
                for (int g = 0; g < kZG; ++g)
                  if (g < zg) {
                    double bv[4];
                    ld4(Bs + ((((size_t)g * nK + K0 + u) * 4 + lr) * 16 + lc) * 4, bv);
                    acc[g] = mfma4(av[u], bv, acc[g]);
                  }
              }
          }
          const long long zt2 = zprof ? wall_clock64() : 0;
#pragma unroll
          for (int g = 0; g < kZG; ++g)
            if (g < zg && J0 + g < Tn) epilogue(I, acc[g], J0 + g, sp, sq);
          const long long zt3 = zprof ? wall_clock64() : 0;
          rowsum(sp);
          if (mk_panel) rowsum(sq);
          if (lc == 15) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * I + lr + 4 * r;
              if (row < nrows) {
#pragma unroll
                for (int b3 = 0; b3 < 3; ++b3) {
                  ScL[3 * row + b3] += sp[r][b3];
                  if (mk_panel) SqL[3 * row + b3] += sq[r][b3];
                }
              }
            }
          }
          if (zprof) {
            const long long zt4 = wall_clock64();
            zt_k += zt2 - zt1; zt_epi += zt3 - zt2; zt_sum += zt4 - zt3;
          }
        }
      }
      if (zprof) {  // (dev aid: wave 0's time in the staging, the K loops, the epilogues, the row sums)
        S.prof[105] = zt_stage; S.prof[106] = zt_k; S.prof[107] = zt_epi; S.prof[108] = zt_sum;
      }
      __syncthreads();
      for (int e = tid; e < nrows * 3; e += kThreads) {
        const int row = e / 3, b3 = e - 3 * row;
        Sc[e] = ScL[e];
        if (mk_panel) pan_pose[(size_t)row * S.jc_ld + b3] = X[(size_t)row * ldx + np + 1 + b3] - SqL[e];
      }
    }
  }
  __syncthreads();
  DRLGX_PROF(S, 101);
  for (int i = tid; i < P; i += kThreads) {
    const double dp0 = X[(size_t)(3 * i) * ldx + np] - dz[3 * i], dp1 = X[(size_t)(3 * i + 1) * ldx + np] - dz[3 * i + 1],
                 dp2 = X[(size_t)(3 * i + 2) * ldx + np] - dz[3 * i + 2];
    d_pose[3 * i] = dp0; d_pose[3 * i + 1] = dp1; d_pose[3 * i + 2] = dp2;
    const Pose t{thp[4 * i], thp[4 * i + 1], thp[4 * i + 2], thp[4 * i + 3]};
    const Pose e = compose(t, make_pose(dp0, dp1, dp2));
    est_pose[4 * i] = e.x; est_pose[4 * i + 1] = e.y; est_pose[4 * i + 2] = e.c; est_pose[4 * i + 3] = e.s;
    if (!full) continue;
    const double *ti = Ti + 6 * i, *sc = Sc + 9 * i;  // sc[3 a + b] = (X_i (-C^-1) X_i^T)[a][b]: symmetric up to round-off
    const double c00 = ti[0] - sc[0], c10 = ti[1] - 0.5 * (sc[1] + sc[3]), c20 = ti[2] - 0.5 * (sc[2] + sc[6]);
    const double c11 = ti[3] - sc[4], c21 = ti[4] - 0.5 * (sc[5] + sc[7]), c22 = ti[5] - sc[8];
    pose_tr[i] = c00 + c11 + c22;
    inv3_sym_fast(c00, c10, c20, c11, c21, c22, pose_info + 6 * i);  // information = inverse(covariance) (SLAM2D.cpp:395-408)
    if (mk_panel) {
      double *go = S.jd + ((size_t)inst * S.P_max + i) * 6;
      go[0] = c00; go[1] = c10; go[2] = c11; go[3] = c20; go[4] = c21; go[5] = c22;
    }
  }
  DRLGX_PROF(S, 9);
  if (mk_panel) {
    // The covariance panel the incremental updates continue from (k_inc.hip): every variable against the active set
    // (newest pose pn, landmarks).  Sigma_pl = X_B (-C^-1) was stored by the pose outputs above, Sigma_ll = C^-1; the
    // cross blocks with the newest pose are  Sigma[i][pn] = (T^-1)[i][pn] + X_i C^-1 X_pn^T = X[., np+1..np+3] - Sigma_pl[i] X_pn^T.
    __syncthreads();
    DRLGX_PROF(S, 102);
    const int pn = P - 1, ldg = S.jc_ld;
    double *pan_lm = pan_pose + (size_t)3 * S.P_max * ldg;
    for (int e = tid; e < 3 * P * 3 && !pn_done; e += kThreads) {  // (the few-landmark branch above: one thread per entry)
      const int row = e / 3, b = e - 3 * row;
      const double *zr = pan_pose + (size_t)row * ldg + 3, *xp = X + (size_t)(3 * pn + b) * ldx;
      double v = X[(size_t)row * ldx + np + 1 + b];
      for (int c = 0; c < np; ++c) v -= zr[c] * xp[c];
      pan_pose[(size_t)row * ldg + b] = v;
    }
    DRLGX_PROF(S, 103);
    for (int e = tid; e < np * np; e += kThreads) {
      const int r = e / np, c = e - r * np;
      pan_lm[(size_t)r * ldg + 3 + c] = -A[c_lds ? AT(max(r, c), min(r, c)) : (size_t)max(r, c) * N + min(r, c)];
    }
    for (int e = tid; e < np * 3; e += kThreads) {
      const int r = e / 3, b = e - 3 * r;
      pan_lm[(size_t)r * ldg + b] = pan_pose[(size_t)(3 * pn + b) * ldg + 3 + r];  // Sigma[l][pn] = Sigma[pn][l]^T
    }
    DRLGX_PROF(S, 104);
    if (tid == 0) {
      int *meta = inc_meta(S, inst);
      meta[0] = 1; meta[1] = P; meta[2] = L; meta[3] = M;
      if (S.inc_stats) atomicAdd(S.inc_stats + 1, 1ull);
    }
  }
  if (tid == 0) {
    if (!refresh) {
      cnt[C_ISAM] = count;
      cnt[C_NEWP] = P;
      cnt[C_NEWL] = L;
    }
    if (bad[0]) atomicMin(S.status, DRLGX_E_NUMERIC);
  }
  if (!refresh && !mk_panel) panel_invalidate(S, inst, tid);  // (a solve for the estimates only leaves no covariance panel)
}

template <int NTW>
__global__ __launch_bounds__(kThreads) void k_slam_arrow(DRLGX_KS_PARAM, LaunchSel sel, int lds_bytes) {
  const DrlgxState &S = DRLGX_KS_REF;
  arrow_body<NTW>(S, sel, lds_bytes);
}
