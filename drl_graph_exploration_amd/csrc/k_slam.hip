// Batched SLAM belief update: one 256-thread workgroup per instance.
//
// Restates SLAM2D::optimize / copy_optimize (src/em_exploration/SLAM2D.cpp:374-488) — one iSAM2
// update (gtsam ISAM2::update, third-party; policy in SURVEY.md App. A.3) followed by the block
// marginals of FastMarginals (src/em_exploration/FastMarginals.cpp:130-186) — as a dense,
// structure-exploiting solve that lives in LDS:
//   1. relinearisation policy (every 10th update, |delta|_inf >= 0.1)
//   2. linearise every factor at theta (one thread per factor), deterministic accumulation
//   3. eliminate the landmarks analytically (2x2 blocks)  -> Schur complement S on the poses (3P x 3P)
//   4. Cholesky of [S rhs] (block-3 right-looking, 2 barriers per pose), forward substitution for free
//   5. in-place triangular inverse + Sigma_pp = Linv^T Linv  (all pose marginals and cross blocks)
//   6. delta_p = Linv^T y;   landmark deltas and 2x2 landmark marginals by back-substitution
//   7. estimates theta (+) delta, information blocks (3x3 LLT inverse / 2x2 inverse), traces
// The 3P x 3P system sits in LDS (<= 45 poses: 155 KB of the 160 KB) or in an HBM/L2 workspace.
#include "drlgx_dev.h"

namespace {

constexpr int kThreads = 256;
constexpr int LIN = 20;  // per-measurement linearisation record: Bxx(6) Bxl(6) Bll(3) gx(3) gl(2)

struct Ws {
  double *lin;   // [M][20]
  double *G;     // [M][6]   Bxl * Lambda_jj^-1  (3x2 row-major)
  double *part;  // [M][4]
  double *lamb;  // [L][8]   Lambda_jj (3), Lambda_jj^-1 (3), eta_j (2)
  double *rhs;   // [3P+1]
  double *Ag;    // global copy of the dense system (when not in LDS)
  int *obs;      // [L][P_max]
  int *mstart;   // [P_max+1]
};

__device__ __forceinline__ Ws carve_ws(const DrlgxState &S, int inst) {
  Ws w;
  double *b = S.slam_ws + (size_t)inst * S.slam_ws_stride;
  w.lin = b;
  b += (size_t)S.M_max * LIN;
  w.G = b;
  b += (size_t)S.M_max * 6;
  w.part = b;
  b += (size_t)S.M_max * 4;
  w.lamb = b;
  b += (size_t)S.L_max * 8;
  w.rhs = b;
  b += (size_t)3 * S.P_max + 4;
  w.Ag = b;
  int *ib = S.slam_iws + (size_t)inst * S.slam_iws_stride;
  w.obs = ib;
  w.mstart = ib + (size_t)S.L_max * S.P_max;
  return w;
}

template <bool kLds>
__global__ __launch_bounds__(kThreads) void k_slam(DrlgxState S, LaunchSel sel) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
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
  const int np = 3 * P, na = np + 1;
  const int ld = (na & 1) ? na : na + 1;  // odd leading dimension: conflict-free column walks in LDS
  Ws w = carve_ws(S, inst);
  double *A = kLds ? smem : w.Ag;          // na x ld, lower triangle + augmented row na-1 = rhs
  double *D = A + (size_t)na * ld;         // [P][6] diagonal blocks of Linv (l00 l10 l11 l20 l21 l22)
  double *th_pose = S.th_pose + (size_t)inst * S.P_max * 4;
  double *d_pose = S.d_pose + (size_t)inst * S.P_max * 3;
  double *th_lm = S.th_lm + (size_t)inst * S.L_max * 2;
  double *d_lm = S.d_lm + (size_t)inst * S.L_max * 2;
  const int *meas_pose = S.meas_pose + (size_t)inst * S.M_max;
  const int *meas_lm = S.meas_lm + (size_t)inst * S.M_max;
  const double *meas_br = S.meas_br + (size_t)inst * S.M_max * 2;

  // ---- 1. relinearisation policy (gtsam ISAM2: relinearizeSkip 10, relinearizeThreshold 0.1) ----
  if (count % 10 == 0) {
    for (int i = tid; i < n_old_p; i += kThreads) {
      double a = fabs(d_pose[3 * i]), b = fabs(d_pose[3 * i + 1]), c = fabs(d_pose[3 * i + 2]);
      double m = fmax(a, fmax(b, c));
      if (m >= 0.1) {
        Pose t{th_pose[4 * i], th_pose[4 * i + 1], th_pose[4 * i + 2], th_pose[4 * i + 3]};
        Pose e = compose(t, make_pose(d_pose[3 * i], d_pose[3 * i + 1], d_pose[3 * i + 2]));
        th_pose[4 * i] = e.x; th_pose[4 * i + 1] = e.y; th_pose[4 * i + 2] = e.c; th_pose[4 * i + 3] = e.s;
        d_pose[3 * i] = d_pose[3 * i + 1] = d_pose[3 * i + 2] = 0;
      }
    }
    for (int j = tid; j < n_old_l; j += kThreads) {
      double m = fmax(fabs(d_lm[2 * j]), fabs(d_lm[2 * j + 1]));
      if (m >= 0.1) {
        th_lm[2 * j] += d_lm[2 * j];
        th_lm[2 * j + 1] += d_lm[2 * j + 1];
        d_lm[2 * j] = d_lm[2 * j + 1] = 0;
      }
    }
  }
  // ---- 2. clear system, observation table, per-pose factor ranges ----
  for (int e = tid; e < na * ld; e += kThreads) A[e] = 0.0;
  for (int e = tid; e < L * S.P_max; e += kThreads) w.obs[e] = -1;
  for (int e = tid; e <= P; e += kThreads) w.mstart[e] = M;
  __syncthreads();
  __threadfence_block();
  // measurement factors are appended in pose order: ranges are contiguous
  for (int m = tid; m < M; m += kThreads) {
    int p = meas_pose[m];
    if (m == 0 || meas_pose[m - 1] != p) w.mstart[p] = m;
    w.obs[(size_t)meas_lm[m] * S.P_max + p] = m;
  }
  __syncthreads();
  // poses without measurements get an empty range [mstart[p+1], mstart[p+1])
  if (tid == 0) {
    for (int p = P - 1; p >= 0; --p)
      if (w.mstart[p] == M) w.mstart[p] = w.mstart[p + 1];
  }
  // ---- 3a. linearise bearing-range factors (SLAM2D.cpp:91-124), one thread per factor ----
  const double wb = 1.0 / (cfg.bearing_noise * cfg.bearing_noise), wr = 1.0 / (cfg.range_noise * cfg.range_noise);
  for (int m = tid; m < M; m += kThreads) {
    const int p = meas_pose[m], j = meas_lm[m];
    Pose ps{th_pose[4 * p], th_pose[4 * p + 1], th_pose[4 * p + 2], th_pose[4 * p + 3]};
    P2 lm{th_lm[2 * j], th_lm[2 * j + 1]};
    double Hbx[3], Hbl[2], Hrx[3], Hrl[2];
    double bp = bearing_of<true>(ps, lm, Hbx, Hbl);
    double rp = range_of<true>(ps, lm, Hrx, Hrl);
    double bm = meas_br[2 * m], rm = meas_br[2 * m + 1];
    double cm = cos(bm), sm = sin(bm), cp = cos(bp), sp = sin(bp);
    double eb = atan2(-sm * cp + cm * sp, cm * cp + sm * sp);  // Rot2 Local(measured, predicted)
    double er = rp - rm;
    double *o = w.lin + (size_t)m * LIN;
    // Bxx (sym 6): xx xy xt yy yt tt
    o[0] = Hbx[0] * wb * Hbx[0] + Hrx[0] * wr * Hrx[0];
    o[1] = Hbx[0] * wb * Hbx[1] + Hrx[0] * wr * Hrx[1];
    o[2] = Hbx[0] * wb * Hbx[2] + Hrx[0] * wr * Hrx[2];
    o[3] = Hbx[1] * wb * Hbx[1] + Hrx[1] * wr * Hrx[1];
    o[4] = Hbx[1] * wb * Hbx[2] + Hrx[1] * wr * Hrx[2];
    o[5] = Hbx[2] * wb * Hbx[2] + Hrx[2] * wr * Hrx[2];
    // Bxl (3x2)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 2; ++c) o[6 + r * 2 + c] = Hbx[r] * wb * Hbl[c] + Hrx[r] * wr * Hrl[c];
    // Bll (sym 3)
    o[12] = Hbl[0] * wb * Hbl[0] + Hrl[0] * wr * Hrl[0];
    o[13] = Hbl[0] * wb * Hbl[1] + Hrl[0] * wr * Hrl[1];
    o[14] = Hbl[1] * wb * Hbl[1] + Hrl[1] * wr * Hrl[1];
    // gx (3), gl (2)  (J^T W e)
    for (int r = 0; r < 3; ++r) o[15 + r] = Hbx[r] * wb * eb + Hrx[r] * wr * er;
    o[18] = Hbl[0] * wb * eb + Hrl[0] * wr * er;
    o[19] = Hbl[1] * wb * eb + Hrl[1] * wr * er;
  }
  __syncthreads();
  __threadfence_block();
  // ---- 3b. per-landmark 2x2 blocks (deterministic: thread j walks its observers in pose order) ----
  for (int j = tid; j < L; j += kThreads) {
    double a = 0, b = 0, d = 0, g0 = 0, g1 = 0;
    for (int p = 0; p < P; ++p) {
      int m = w.obs[(size_t)j * S.P_max + p];
      if (m < 0) continue;
      const double *o = w.lin + (size_t)m * LIN;
      a += o[12]; b += o[13]; d += o[14];
      g0 += o[18]; g1 += o[19];
    }
    double id = 1.0 / (a * d - b * b);
    double *lb = w.lamb + (size_t)j * 8;
    lb[0] = a; lb[1] = b; lb[2] = d;
    lb[3] = d * id; lb[4] = -b * id; lb[5] = a * id;  // Lambda_jj^-1
    lb[6] = -g0; lb[7] = -g1;                           // eta_j
  }
  // ---- 3c. per-pose diagonal / sub-diagonal blocks: prior + odometry + own measurements ----
  for (int i = tid; i < P; i += kThreads) {
    double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // A_ii (full 3x3, symmetric)
    double g[3] = {0, 0, 0};
    Pose ti{th_pose[4 * i], th_pose[4 * i + 1], th_pose[4 * i + 2], th_pose[4 * i + 3]};
    if (i == 0) {  // prior (SLAM2D.cpp:44-57): e = Local(prior, x0), J = diag(R_h^T, 1)
      const double *pr = S.prior + (size_t)inst * DRLGX_PRIOR_STRIDE;
      Pose pp{pr[0], pr[1], pr[2], pr[3]};
      Pose h = between(pp, ti, nullptr);
      double e[3] = {h.x, h.y, theta_of(h)};
      double J[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
      const double *W = pr + 4;
      double WJ[9], We[3];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) WJ[r * 3 + c] = W[r * 3] * J[c] + W[r * 3 + 1] * J[3 + c] + W[r * 3 + 2] * J[6 + c];
        We[r] = W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2];
      }
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) B[r * 3 + c] += J[r] * WJ[c] + J[3 + r] * WJ[3 + c] + J[6 + r] * WJ[6 + c];
        g[r] += J[r] * We[0] + J[3 + r] * We[1] + J[6 + r] * We[2];
      }
    }
    const double wo[3] = {1.0 / (cfg.translation_noise * cfg.translation_noise),
                          1.0 / (cfg.translation_noise * cfg.translation_noise),
                          1.0 / (cfg.rotation_noise * cfg.rotation_noise)};
    if (i > 0) {  // odometry factor i-1 seen from its second key: J2 = Hlocal
      const double *oo = S.odo + ((size_t)inst * S.P_max + (i - 1)) * 4;
      Pose tm{th_pose[4 * (i - 1)], th_pose[4 * (i - 1) + 1], th_pose[4 * (i - 1) + 2], th_pose[4 * (i - 1) + 3]};
      Pose hx = between(tm, ti, nullptr);
      Pose h = between(Pose{oo[0], oo[1], oo[2], oo[3]}, hx, nullptr);
      double e[3] = {h.x, h.y, theta_of(h)};
      double J2[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          B[r * 3 + c] += J2[r] * wo[0] * J2[c] + J2[3 + r] * wo[1] * J2[3 + c] + J2[6 + r] * wo[2] * J2[6 + c];
        g[r] += J2[r] * wo[0] * e[0] + J2[3 + r] * wo[1] * e[1] + J2[6 + r] * wo[2] * e[2];
      }
    }
    if (i + 1 < P) {  // odometry factor i seen from its first key: J1 = Hlocal * H1; also the (i+1,i) block
      const double *oo = S.odo + ((size_t)inst * S.P_max + i) * 4;
      Pose tn{th_pose[4 * (i + 1)], th_pose[4 * (i + 1) + 1], th_pose[4 * (i + 1) + 2], th_pose[4 * (i + 1) + 3]};
      double H1[9];
      Pose hx = between(ti, tn, H1);
      Pose h = between(Pose{oo[0], oo[1], oo[2], oo[3]}, hx, nullptr);
      double e[3] = {h.x, h.y, theta_of(h)};
      double Hl[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
      double J1[9];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) J1[r * 3 + c] = Hl[r * 3] * H1[c] + Hl[r * 3 + 1] * H1[3 + c] + Hl[r * 3 + 2] * H1[6 + c];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          B[r * 3 + c] += J1[r] * wo[0] * J1[c] + J1[3 + r] * wo[1] * J1[3 + c] + J1[6 + r] * wo[2] * J1[6 + c];
        g[r] += J1[r] * wo[0] * e[0] + J1[3 + r] * wo[1] * e[1] + J1[6 + r] * wo[2] * e[2];
      }
      // block (i+1, i) = J2^T W J1
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          A[(size_t)(3 * (i + 1) + r) * ld + 3 * i + c] =
              Hl[r] * wo[0] * J1[c] + Hl[3 + r] * wo[1] * J1[3 + c] + Hl[6 + r] * wo[2] * J1[6 + c];
    }
    // own bearing-range factors
    for (int m = w.mstart[i]; m < w.mstart[i + 1]; ++m) {
      const double *o = w.lin + (size_t)m * LIN;
      B[0] += o[0]; B[1] += o[1]; B[2] += o[2];
      B[3] += o[1]; B[4] += o[3]; B[5] += o[4];
      B[6] += o[2]; B[7] += o[4]; B[8] += o[5];
      g[0] += o[15]; g[1] += o[16]; g[2] += o[17];
    }
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c <= r; ++c) A[(size_t)(3 * i + r) * ld + 3 * i + c] = B[r * 3 + c];
      w.rhs[3 * i + r] = -g[r];
    }
  }
  __syncthreads();
  __threadfence_block();
  // ---- 4. landmark elimination: G_m = Bxl_m Lambda_jj^-1, Schur complement on the poses ----
  for (int m = tid; m < M; m += kThreads) {
    const double *o = w.lin + (size_t)m * LIN;
    const double *lb = w.lamb + (size_t)meas_lm[m] * 8;
    double *G = w.G + (size_t)m * 6;
    for (int r = 0; r < 3; ++r) {
      G[r * 2 + 0] = o[6 + r * 2] * lb[3] + o[6 + r * 2 + 1] * lb[4];
      G[r * 2 + 1] = o[6 + r * 2] * lb[4] + o[6 + r * 2 + 1] * lb[5];
    }
  }
  __syncthreads();
  __threadfence_block();
  {
    const int npairs = P * (P + 1) / 2;
    for (int e = tid; e < npairs; e += kThreads) {
      // unrank (p >= q) from e = p(p+1)/2 + q
      int p = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((p + 1) * (p + 2) / 2 <= e) ++p;
      while (p * (p + 1) / 2 > e) --p;
      int q = e - p * (p + 1) / 2;
      double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      bool any = false;
      for (int m = w.mstart[p]; m < w.mstart[p + 1]; ++m) {
        int mq = w.obs[(size_t)meas_lm[m] * S.P_max + q];
        if (mq < 0) continue;
        any = true;
        const double *G = w.G + (size_t)m * 6;
        const double *Bq = w.lin + (size_t)mq * LIN + 6;  // Bxl of (q, j): 3x2
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) acc[r * 3 + c] += G[r * 2] * Bq[c * 2] + G[r * 2 + 1] * Bq[c * 2 + 1];
      }
      if (any) {
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) {
            if (p == q && c > r) continue;
            A[(size_t)(3 * p + r) * ld + 3 * q + c] -= acc[r * 3 + c];
          }
      }
    }
    // rhs_p -= sum_m G_m eta_j ; then copy rhs into the augmented row
    for (int p = tid; p < P; p += kThreads) {
      double s0 = 0, s1 = 0, s2 = 0;
      for (int m = w.mstart[p]; m < w.mstart[p + 1]; ++m) {
        const double *G = w.G + (size_t)m * 6;
        const double *lb = w.lamb + (size_t)meas_lm[m] * 8;
        // Lambda_pl Lambda_ll^-1 eta_l = Bxl (Linv eta) = G eta  (G already contains Lambda^-1)
        s0 += G[0] * lb[6] + G[1] * lb[7];
        s1 += G[2] * lb[6] + G[3] * lb[7];
        s2 += G[4] * lb[6] + G[5] * lb[7];
      }
      A[(size_t)np * ld + 3 * p + 0] = w.rhs[3 * p + 0] - s0;
      A[(size_t)np * ld + 3 * p + 1] = w.rhs[3 * p + 1] - s1;
      A[(size_t)np * ld + 3 * p + 2] = w.rhs[3 * p + 2] - s2;
    }
  }
  __syncthreads();
  // ---- 5. block-3 right-looking Cholesky of the augmented system (rows 0..np incl. rhs row) ----
  bool bad = false;
  for (int kb = 0; kb < P; ++kb) {
    const int k0 = 3 * kb;
    // every thread factors the 3x3 diagonal block redundantly (broadcast LDS reads)
    const double a00 = A[(size_t)k0 * ld + k0], a10 = A[(size_t)(k0 + 1) * ld + k0], a11 = A[(size_t)(k0 + 1) * ld + k0 + 1];
    const double a20 = A[(size_t)(k0 + 2) * ld + k0], a21 = A[(size_t)(k0 + 2) * ld + k0 + 1],
                 a22 = A[(size_t)(k0 + 2) * ld + k0 + 2];
    const double l00 = sqrt(a00), l10 = a10 / l00, l20 = a20 / l00;
    const double l11 = sqrt(a11 - l10 * l10), l21 = (a21 - l20 * l10) / l11;
    const double l22 = sqrt(a22 - l20 * l20 - l21 * l21);
    if (!(l00 > 0) || !(l11 > 0) || !(l22 > 0)) bad = true;
    __syncthreads();  // all reads of the diagonal block done before it is overwritten
    if (tid == 0) {
      A[(size_t)k0 * ld + k0] = l00;
      A[(size_t)(k0 + 1) * ld + k0] = l10; A[(size_t)(k0 + 1) * ld + k0 + 1] = l11;
      A[(size_t)(k0 + 2) * ld + k0] = l20; A[(size_t)(k0 + 2) * ld + k0 + 1] = l21; A[(size_t)(k0 + 2) * ld + k0 + 2] = l22;
    }
    // panel: rows below the block (incl. the rhs row)
    for (int i = k0 + 3 + tid; i < na; i += kThreads) {
      double *row = A + (size_t)i * ld + k0;
      double x0 = row[0] / l00;
      double x1 = (row[1] - x0 * l10) / l11;
      double x2 = (row[2] - x0 * l20 - x1 * l21) / l22;
      row[0] = x0; row[1] = x1; row[2] = x2;
    }
    __syncthreads();
    // trailing update: A[i][c] -= L[i][k0..k0+2] . L[c][k0..k0+2]   (i >= c >= k0+3)
    {
      const int r0 = k0 + 3;
      const int tx = tid & 15, ty = tid >> 4;
      for (int i = r0 + ty; i < na; i += 16) {
        const double *li = A + (size_t)i * ld + k0;
        const double li0 = li[0], li1 = li[1], li2 = li[2];
        const int cmax = (i < np) ? i : np - 1;  // rhs row only updates columns < np
        for (int c = r0 + tx; c <= cmax; c += 16) {
          const double *lc = A + (size_t)c * ld + k0;
          A[(size_t)i * ld + c] -= li0 * lc[0] + li1 * lc[1] + li2 * lc[2];
        }
      }
    }
    __syncthreads();
  }
  // y = L^-1 rhs now sits in row np (A[np][0..np-1])
  // ---- 6. triangular inverse: Linv strictly-lower stored TRANSPOSED in the upper triangle,
  //         diagonal 3x3 blocks of Linv in D.  Block row kb depends on block rows < kb. ----
  for (int kb = tid; kb < P; kb += kThreads) {
    const int k0 = 3 * kb;
    const double l00 = A[(size_t)k0 * ld + k0], l10 = A[(size_t)(k0 + 1) * ld + k0], l11 = A[(size_t)(k0 + 1) * ld + k0 + 1];
    const double l20 = A[(size_t)(k0 + 2) * ld + k0], l21 = A[(size_t)(k0 + 2) * ld + k0 + 1],
                 l22 = A[(size_t)(k0 + 2) * ld + k0 + 2];
    double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    double i10 = -l10 * i00 * i11;
    double i21 = -l21 * i11 * i22;
    double i20 = -(l20 * i00 + l21 * i10) * i22;
    double *d = D + (size_t)kb * 6;
    d[0] = i00; d[1] = i10; d[2] = i11; d[3] = i20; d[4] = i21; d[5] = i22;
  }
  __syncthreads();
  for (int kb = 1; kb < P; ++kb) {
    const int k0 = 3 * kb;
    // X[k0..k0+2][c] for c < k0:  X_k = Linv_kk * ( - sum_{j<k, j>=cb} L[k][j] X[j][c] )
    const double *d = D + (size_t)kb * 6;
    for (int c = tid; c < k0; c += kThreads) {
      const int cb = c / 3;
      // contribution of the diagonal block of column-block cb: X[3cb+r][c] = Dcb[r][c-3cb] (lower tri)
      const double *dc = D + (size_t)cb * 6;
      const int cc = c - 3 * cb;
      double x[3];  // Linv[3cb + r][c], r = 0..2
      x[0] = (cc == 0) ? dc[0] : 0.0;
      x[1] = (cc == 0) ? dc[1] : (cc == 1 ? dc[2] : 0.0);
      x[2] = (cc == 0) ? dc[3] : (cc == 1 ? dc[4] : dc[5]);
      double s0 = 0, s1 = 0, s2 = 0;
      const double *r0p = A + (size_t)k0 * ld, *r1p = A + (size_t)(k0 + 1) * ld, *r2p = A + (size_t)(k0 + 2) * ld;
      for (int r = 0; r < 3; ++r) {
        const int j = 3 * cb + r;
        s0 += r0p[j] * x[r]; s1 += r1p[j] * x[r]; s2 += r2p[j] * x[r];
      }
      const double *xr = A + (size_t)c * ld;  // X[j][c] for j > 3cb+2 lives at A[c][j]
      for (int j = 3 * cb + 3; j < k0; ++j) {
        const double xv = xr[j];
        s0 += r0p[j] * xv; s1 += r1p[j] * xv; s2 += r2p[j] * xv;
      }
      const double y0 = -(d[0] * s0);
      const double y1 = -(d[1] * s0 + d[2] * s1);
      const double y2 = -(d[3] * s0 + d[4] * s1 + d[5] * s2);
      A[(size_t)c * ld + k0] = y0;
      A[(size_t)c * ld + k0 + 1] = y1;
      A[(size_t)c * ld + k0 + 2] = y2;
    }
    __syncthreads();
  }
  // ---- 7a. delta_p = Linv^T y  (one thread per component) ----
  for (int a = tid; a < np; a += kThreads) {
    const int ab = a / 3, ac = a - 3 * ab;
    const double *dd = D + (size_t)ab * 6;
    const double *yrow = A + (size_t)np * ld;
    // within the diagonal block: Linv[3ab + r][a] for r >= ac
    double s = 0;
    if (ac == 0) s = dd[0] * yrow[3 * ab] + dd[1] * yrow[3 * ab + 1] + dd[3] * yrow[3 * ab + 2];
    else if (ac == 1) s = dd[2] * yrow[3 * ab + 1] + dd[4] * yrow[3 * ab + 2];
    else s = dd[5] * yrow[3 * ab + 2];
    const double *xr = A + (size_t)a * ld;
    for (int k = 3 * ab + 3; k < np; ++k) s += xr[k] * yrow[k];
    d_pose[a] = s;
  }
  __syncthreads();
  // ---- 7b. Sigma_pp = Linv^T Linv into the lower triangle (L no longer needed) ----
  {
    const int nent = np * (np + 1) / 2;
    for (int e = tid; e < nent; e += kThreads) {
      int a = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while ((a + 1) * (a + 2) / 2 <= e) ++a;
      while (a * (a + 1) / 2 > e) --a;
      const int b = e - a * (a + 1) / 2;  // a >= b
      const int ab = a / 3, ac = a - 3 * ab;
      // sum over k >= a:  Linv[k][a] * Linv[k][b]
      // k inside a's diagonal block: Linv[k][a] from D;  Linv[k][b] from D (same block) or upper storage
      const double *da = D + (size_t)ab * 6;
      double la[3];  // Linv[3ab + r][a]
      la[0] = (ac == 0) ? da[0] : 0.0;
      la[1] = (ac == 0) ? da[1] : (ac == 1 ? da[2] : 0.0);
      la[2] = (ac == 0) ? da[3] : (ac == 1 ? da[4] : da[5]);
      const int bb = b / 3, bc = b - 3 * bb;
      double lbv[3];
      if (bb == ab) {
        lbv[0] = (bc == 0) ? da[0] : 0.0;
        lbv[1] = (bc == 0) ? da[1] : (bc == 1 ? da[2] : 0.0);
        lbv[2] = (bc == 0) ? da[3] : (bc == 1 ? da[4] : da[5]);
      } else {
        const double *xb = A + (size_t)b * ld + 3 * ab;
        lbv[0] = xb[0]; lbv[1] = xb[1]; lbv[2] = xb[2];
      }
      double s = la[0] * lbv[0] + la[1] * lbv[1] + la[2] * lbv[2];
      const double *xa = A + (size_t)a * ld, *xb = A + (size_t)b * ld;
      for (int k = 3 * ab + 3; k < np; ++k) s += xa[k] * xb[k];
      // in place: this phase reads only the upper triangle and D, and writes only the lower triangle
      A[(size_t)a * ld + b] = s;
    }
  }
  __syncthreads();
  __threadfence_block();
  // ---- 8. landmark marginals and deltas ----
  // part_m = G_m^T * ( sum_{m' of same landmark} Sigma[p_m][p_m'] G_m' )
  for (int m = tid; m < M; m += kThreads) {
    const int j = meas_lm[m], p = meas_pose[m];
    double Wm[6] = {0, 0, 0, 0, 0, 0};  // 3x2
    for (int q = 0; q < P; ++q) {
      const int mq = w.obs[(size_t)j * S.P_max + q];
      if (mq < 0) continue;
      const double *Gq = w.G + (size_t)mq * 6;
      for (int r = 0; r < 3; ++r) {
        double s0 = 0, s1 = 0;
        for (int c = 0; c < 3; ++c) {
          const int ra = 3 * p + r, cb2 = 3 * q + c;
          const double sg = (ra >= cb2) ? A[(size_t)ra * ld + cb2] : A[(size_t)cb2 * ld + ra];
          s0 += sg * Gq[c * 2];
          s1 += sg * Gq[c * 2 + 1];
        }
        Wm[r * 2] += s0;
        Wm[r * 2 + 1] += s1;
      }
    }
    const double *G = w.G + (size_t)m * 6;
    double *pt = w.part + (size_t)m * 4;
    pt[0] = G[0] * Wm[0] + G[2] * Wm[2] + G[4] * Wm[4];
    pt[1] = G[0] * Wm[1] + G[2] * Wm[3] + G[4] * Wm[5];
    pt[2] = G[1] * Wm[0] + G[3] * Wm[2] + G[5] * Wm[4];
    pt[3] = G[1] * Wm[1] + G[3] * Wm[3] + G[5] * Wm[5];
  }
  __syncthreads();
  __threadfence_block();
  double *est_lm = S.est_lm + (size_t)inst * S.L_max * 2;
  double *lm_info = S.lm_info + (size_t)inst * S.L_max * 3;
  double *lm_tr = S.lm_tr + (size_t)inst * S.L_max;
  for (int j = tid; j < L; j += kThreads) {
    const double *lb = w.lamb + (size_t)j * 8;
    double c00 = lb[3], c01 = lb[4], c10 = lb[4], c11 = lb[5];
    // delta_j = Lambda^-1 eta_j - sum_m G_m^T delta_p
    double dx = lb[3] * lb[6] + lb[4] * lb[7], dy = lb[4] * lb[6] + lb[5] * lb[7];
    for (int p = 0; p < P; ++p) {
      const int m = w.obs[(size_t)j * S.P_max + p];
      if (m < 0) continue;
      const double *pt = w.part + (size_t)m * 4;
      c00 += pt[0]; c01 += pt[1]; c10 += pt[2]; c11 += pt[3];
      const double *G = w.G + (size_t)m * 6;
      const double *dp = d_pose + 3 * p;
      dx -= G[0] * dp[0] + G[2] * dp[1] + G[4] * dp[2];
      dy -= G[1] * dp[0] + G[3] * dp[1] + G[5] * dp[2];
    }
    d_lm[2 * j] = dx;
    d_lm[2 * j + 1] = dy;
    est_lm[2 * j] = th_lm[2 * j] + dx;
    est_lm[2 * j + 1] = th_lm[2 * j + 1] + dy;
    const double cs = 0.5 * (c01 + c10);
    lm_tr[j] = c00 + c11;
    // marginalCovariance(l).inverse() (SLAM2D.cpp:417)
    const double id = 1.0 / (c00 * c11 - cs * cs);
    lm_info[3 * j] = c11 * id;
    lm_info[3 * j + 1] = -cs * id;
    lm_info[3 * j + 2] = c00 * id;
  }
  // ---- 9. pose estimates, information = inverse(covariance) by LLT (SLAM2D.cpp:395-408) ----
  double *est_pose = S.est_pose + (size_t)inst * S.P_max * 4;
  double *pose_info = S.pose_info + (size_t)inst * S.P_max * 6;
  double *pose_tr = S.pose_tr + (size_t)inst * S.P_max;
  for (int i = tid; i < P; i += kThreads) {
    Pose t{th_pose[4 * i], th_pose[4 * i + 1], th_pose[4 * i + 2], th_pose[4 * i + 3]};
    Pose e = compose(t, make_pose(d_pose[3 * i], d_pose[3 * i + 1], d_pose[3 * i + 2]));
    est_pose[4 * i] = e.x; est_pose[4 * i + 1] = e.y; est_pose[4 * i + 2] = e.c; est_pose[4 * i + 3] = e.s;
    const int k0 = 3 * i;
    const double c00 = A[(size_t)k0 * ld + k0], c10 = A[(size_t)(k0 + 1) * ld + k0], c11 = A[(size_t)(k0 + 1) * ld + k0 + 1];
    const double c20 = A[(size_t)(k0 + 2) * ld + k0], c21 = A[(size_t)(k0 + 2) * ld + k0 + 1],
                 c22 = A[(size_t)(k0 + 2) * ld + k0 + 2];
    pose_tr[i] = c00 + c11 + c22;
    LLT3 llt(c00, c10, c20, c11, c21, c22);
    double x0, x1, x2;
    double *pi = pose_info + 6 * i;
    llt.solve(1, 0, 0, x0, x1, x2);
    pi[0] = x0; pi[1] = x1; pi[2] = x2;
    llt.solve(0, 1, 0, x0, x1, x2);
    pi[3] = x1; pi[4] = x2;
    llt.solve(0, 0, 1, x0, x1, x2);
    pi[5] = x2;
  }
  if (tid == 0) {
    cnt[C_ISAM] = count;
    cnt[C_NEWP] = P;
    cnt[C_NEWL] = L;
    if (bad) atomicMin(S.status, DRLGX_E_NUMERIC);
  }
}

}  // namespace

size_t drlgx_slam_lds_bytes(int P_max, int, int) {
  int np = 3 * P_max, na = np + 1, ld = (na & 1) ? na : na + 1;
  return ((size_t)na * ld + (size_t)P_max * 6) * sizeof(double);
}

void drlgx_launch_slam(const DrlgxState &S, hipStream_t st, LaunchSel sel) {
  const size_t lds = drlgx_slam_lds_bytes(S.P_max, S.L_max, S.M_max);
  if (lds <= 160 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute(reinterpret_cast<const void *>(&k_slam<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(160 * 1024));
      attr_set = true;
    }
    hipLaunchKernelGGL(k_slam<true>, dim3(sel.n), dim3(kThreads), lds, st, S, sel);
  } else {
    hipLaunchKernelGGL(k_slam<false>, dim3(sel.n), dim3(kThreads), 0, st, S, sel);
  }
}
