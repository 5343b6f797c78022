// EKF-style covariance propagate / update without re-solving: FastMarginals2::update / propagate
// (src/em_exploration/FastMarginals.cpp:188-321) fed the way the EM planner feeds it
// (src/em_exploration/Planner2D.cpp:652-737 updateNodeInformation_EM, :472-551 updateTrajectory_EM):
//   * one new pose per action, predicted noise-free from the last pose's ESTIMATE; its odometry factor linearised at
//     (linearisation point of the last pose | predicted pose) gives   cov1 = H1 (H0 cov0 H0^T + I) H1^T,
//     F_b = -H1 H0, Fs_b = F_b Fs_{b-1}   (H0, H1^-1 = the whitened Jacobian blocks);
//   * noise-free bearing-range factors from every new pose to the landmarks of the ESTIMATED map that pass the sensor
//     gates, linearised at (predicted pose | landmark linearisation point): whitened rows A;
//   * Sigma' = Sigma - Sigma A^T (I + A Sigma A^T)^-1 A Sigma on the diagonal block of every pose.
// The joint prior Sigma (FastMarginals::recover) is the dense inverse of the information matrix linearised at the iSAM
// linearisation point: k_fm2_prior builds and inverts it per environment (in HBM, n = 3P + 2L); k_fm2_update then needs
// only blocks of it: T = I + A Sigma A^T is assembled measurement pair by measurement pair from
//   Sigma(l, l'), Sigma(l, x_new b) = Sigma(l, x_last) Fs_b^T, Sigma(x_new a, x_new b) = cov_a F_{a+1}^T ... F_b^T,
// inverted (SPD, Gauss-Jordan), and  Delta_i = -U_i^T T^-1 U_i  with  U_i = A Sigma(., x_i)  (2 rows per measurement).
// This is the planner-side primitive of SURVEY.md section 8(f)-1; it is not on the belief-step hot path and is written for
// clarity: one workgroup per environment / candidate, matrices in an HBM scratch, O(n^3) / O(dim^3) sweeps.
#include "drlgx_dev.h"

namespace kfm2 {

constexpr int kT = 256;
constexpr int REC = 12;

// BearingRangeFactor Jacobians at (pose, point): Jx 2x3 (bearing row, range row), Jl 2x2
__device__ inline void br_jac(const Pose &ps, const P2 &lm, double *Jx, double *Jl) {
  (void)bearing_of<true>(ps, lm, Jx, Jl);
  (void)range_of<true>(ps, lm, Jx + 3, Jl + 2);
}

__device__ inline void mat3_mul(const double *a, const double *b, double *o) {  // o = a b
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
}
__device__ inline void mat3_mul_bt(const double *a, const double *b, double *o) {  // o = a b^T
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[r * 3] * b[c * 3] + a[r * 3 + 1] * b[c * 3 + 1] + a[r * 3 + 2] * b[c * 3 + 2];
}

// In-place inverse of a symmetric positive definite n x n matrix (row major, leading dimension ld) by Gauss-Jordan
// sweeps; `col` / `row`: LDS scratch of n doubles each.  Whole workgroup.
__device__ inline void spd_inverse(double *A, int n, int ld, double *col, double *row, int *bad) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int k = 0; k < n; ++k) {
    for (int i = tid; i < n; i += nt) {
      col[i] = A[(size_t)i * ld + k];
      row[i] = A[(size_t)k * ld + i];
    }
    __syncthreads();
    const double p = col[k];
    if (tid == 0 && !(p > 0)) *bad = 1;
    const double ip = 1.0 / p;
    for (size_t e = tid; e < (size_t)n * n; e += nt) {
      const int i = (int)(e / n), j = (int)(e - (size_t)i * n);
      double v;
      if (i == k) v = (j == k) ? ip : row[j] * ip;
      else if (j == k) v = -col[i] * ip;
      else v = A[(size_t)i * ld + j] - col[i] * row[j] * ip;
      A[(size_t)i * ld + j] = v;
    }
    __syncthreads();
  }
}

// Sigma = (J^T W J)^-1 at the linearisation point of environment env_ids[blockIdx.x]; order: poses (3 each), then landmarks
// by slot (2 each); written to sig + blockIdx.x * nmax * nmax with leading dimension n = 3P + 2L.
__global__ __launch_bounds__(kT) void k_fm2_prior(DrlgxState S, const int32_t *env_ids, int n_env, double *sig, size_t sig_stride) {
  __shared__ int bad;
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const int inst = env_ids ? env_ids[blockIdx.x] : blockIdx.x;
  const int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  const int P = cnt[C_P], L = cnt[C_L], M = cnt[C_M], n = 3 * P + 2 * L;
  const drlgx_config &cfg = S.cfg;
  double *A = sig + (size_t)blockIdx.x * sig_stride;
  const double *thp = S.th_pose + (size_t)inst * S.P_max * 4;
  const double *thl = S.th_lm + (size_t)inst * S.L_max * 2;
  if (tid == 0) bad = 0;
  for (size_t e = tid; e < (size_t)n * n; e += kT) A[e] = 0.0;
  __syncthreads();
  auto add = [&](int r0, int c0, int nr, int nc, const double *m) {
    for (int r = 0; r < nr; ++r)
      for (int c = 0; c < nc; ++c) atomicAdd(&A[(size_t)(r0 + r) * n + c0 + c], m[r * nc + c]);
  };
  // (fp64 atomics: the summation order of a block's contributions is not fixed; this primitive is tolerance-level)
  const double wo[3] = {1.0 / (cfg.translation_noise * cfg.translation_noise), 1.0 / (cfg.translation_noise * cfg.translation_noise),
                        1.0 / (cfg.rotation_noise * cfg.rotation_noise)};
  const double wb = 1.0 / (cfg.bearing_noise * cfg.bearing_noise), wr = 1.0 / (cfg.range_noise * cfg.range_noise);
  for (int i = tid; i < P; i += kT) {
    const Pose ti{thp[4 * i], thp[4 * i + 1], thp[4 * i + 2], thp[4 * i + 3]};
    if (i == 0) {  // prior: J = diag(R_h^T, 1), W = the prior information
      const double *pr = S.prior + (size_t)inst * DRLGX_PRIOR_STRIDE;
      const Pose h = between(Pose{pr[0], pr[1], pr[2], pr[3]}, ti, nullptr);
      const double J[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
      const double *W = pr + 4;
      double WJ[9], B[9];
      mat3_mul(W, J, WJ);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) B[r * 3 + c] = J[r] * WJ[c] + J[3 + r] * WJ[3 + c] + J[6 + r] * WJ[6 + c];
      add(0, 0, 3, 3, B);
    }
    if (i + 1 < P) {  // odometry factor i: blocks (i,i), (i,i+1), (i+1,i), (i+1,i+1)
      const double *oo = S.odo + ((size_t)inst * S.P_max + i) * 4;
      const Pose tn{thp[4 * (i + 1)], thp[4 * (i + 1) + 1], thp[4 * (i + 1) + 2], thp[4 * (i + 1) + 3]};
      double H1[9];
      const Pose hx = between(ti, tn, H1);
      const Pose h = between(Pose{oo[0], oo[1], oo[2], oo[3]}, hx, nullptr);
      const double J2[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
      double J1[9], B11[9], B12[9], B21[9], B22[9];
      mat3_mul(J2, H1, J1);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
          double s11 = 0, s12 = 0, s22 = 0;
          for (int k = 0; k < 3; ++k) {
            s11 += J1[k * 3 + r] * wo[k] * J1[k * 3 + c];
            s12 += J1[k * 3 + r] * wo[k] * J2[k * 3 + c];
            s22 += J2[k * 3 + r] * wo[k] * J2[k * 3 + c];
          }
          B11[r * 3 + c] = s11; B12[r * 3 + c] = s12; B21[c * 3 + r] = s12; B22[r * 3 + c] = s22;
        }
      add(3 * i, 3 * i, 3, 3, B11);
      add(3 * i, 3 * i + 3, 3, 3, B12);
      add(3 * i + 3, 3 * i, 3, 3, B21);
      add(3 * i + 3, 3 * i + 3, 3, 3, B22);
    }
  }
  const int *mp = S.meas_pose + (size_t)inst * S.M_max, *ml = S.meas_lm + (size_t)inst * S.M_max;
  for (int m = tid; m < M; m += kT) {
    const int p = mp[m], j = ml[m];
    double Jx[6], Jl[4];
    br_jac(Pose{thp[4 * p], thp[4 * p + 1], thp[4 * p + 2], thp[4 * p + 3]}, P2{thl[2 * j], thl[2 * j + 1]}, Jx, Jl);
    double Bxx[9], Bxl[6], Blx[6], Bll[4];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) Bxx[r * 3 + c] = Jx[r] * wb * Jx[c] + Jx[3 + r] * wr * Jx[3 + c];
      for (int c = 0; c < 2; ++c) {
        const double v = Jx[r] * wb * Jl[c] + Jx[3 + r] * wr * Jl[2 + c];
        Bxl[r * 2 + c] = v;
        Blx[c * 3 + r] = v;
      }
    }
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c) Bll[r * 2 + c] = Jl[r] * wb * Jl[c] + Jl[2 + r] * wr * Jl[2 + c];
    add(3 * p, 3 * p, 3, 3, Bxx);
    add(3 * p, 3 * P + 2 * j, 3, 2, Bxl);
    add(3 * P + 2 * j, 3 * p, 2, 3, Blx);
    add(3 * P + 2 * j, 3 * P + 2 * j, 2, 2, Bll);
  }
  __syncthreads();
  spd_inverse(A, n, n, lds, lds + n, &bad);
  if (tid == 0 && bad) atomicMin(S.status, DRLGX_E_NUMERIC);
}

struct Cand {
  int P, L, K, nm;            // old poses, landmarks, new poses, measurements
  const double *Sig;          // prior covariance of the candidate's environment, ld = n
  int n;
  double *cov_new, *F, *Fs;   // [K][9] each
  double *NN;                 // [K][K][9]: Sigma(new a, new b), a <= b
  const int *mk, *mj;         // [nm] measurement -> new pose index, landmark slot
  const double *J;            // [nm][10]: whitened Jx (2x3), Jl (2x2)
};
// Sigma(new pose a, new pose b), any order (3x3)
__device__ inline void cov_nn(const Cand &c, int a, int b, double *o) {
  if (a <= b) {
    for (int q = 0; q < 9; ++q) o[q] = c.NN[((size_t)a * c.K + b) * 9 + q];
  } else {
    const double *t = c.NN + ((size_t)b * c.K + a) * 9;
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) o[r * 3 + q] = t[q * 3 + r];
  }
}
// Sigma(old variable at rows r0..r0+nr of Sigma, new pose b) = Sigma(., x_last) Fs_b^T   (nr x 3)
__device__ inline void cov_old_new(const Cand &c, int r0, int nr, int b, double *o) {
  const double *fs = c.Fs + 9 * b;
  const int cl = 3 * (c.P - 1);
  for (int r = 0; r < nr; ++r) {
    const double *s = c.Sig + (size_t)(r0 + r) * c.n + cl;
    for (int q = 0; q < 3; ++q) o[r * 3 + q] = s[0] * fs[q * 3] + s[1] * fs[q * 3 + 1] + s[2] * fs[q * 3 + 2];
  }
}

// One workgroup per candidate.  scratch (per candidate, doubles): cov_new 9K | F 9K | Fs 9K | NN 9K^2 | J 10 nm_max | T dim_max^2
// | U 3 dim_max ; ints: mk nm_max | mj nm_max.
__global__ __launch_bounds__(kT) void k_fm2_update(DrlgxState S, int c0, const int32_t *cand_env, const int32_t *env_slot,
                                                   const double *actions, const int32_t *n_actions, const double *sig,
                                                   size_t sig_stride, double *scratch, size_t scratch_stride, int *iscratch,
                                                   int nm_max, double *cov_out, int out_stride, int32_t *n_out) {
  __shared__ int nm_s, bad;
  __shared__ int wcount[kT / 64];
  __shared__ double gj[2 * 512];  // Gauss-Jordan row / column of T: dim = 2 nm <= 512
  const int tid = threadIdx.x, ci = c0 + blockIdx.x;
  const int env = cand_env[ci];
  const drlgx_config &cfg = S.cfg;
  const int *cnt = S.cnt + (size_t)env * DRLGX_CNT_STRIDE;
  const int P = cnt[C_P], L = cnt[C_L];
  const int K = n_actions[ci];
  double *sc = scratch + (size_t)blockIdx.x * scratch_stride;
  double *cov_new = sc, *F = cov_new + 9 * S.A_max, *Fs = F + 9 * S.A_max, *NN = Fs + 9 * S.A_max;
  double *J = NN + (size_t)9 * S.A_max * S.A_max, *T = J + (size_t)10 * nm_max, *U = T + (size_t)4 * nm_max * nm_max;
  double *newp = U + (size_t)(kT / 64) * 6 * nm_max;  // [K][4] predicted poses (U: one [2 nm_max][3] slab per wave)
  int *mk = iscratch + (size_t)blockIdx.x * 2 * nm_max, *mj = mk + nm_max;
  const int n = 3 * P + 2 * L;
  const double *Sig = sig + (size_t)env_slot[env] * sig_stride;
  const double *thp = S.th_pose + (size_t)env * S.P_max * 4, *thl = S.th_lm + (size_t)env * S.L_max * 2;
  const double *ep = S.est_pose + (size_t)env * S.P_max * 4, *el = S.est_lm + (size_t)env * S.L_max * 2;
  double *out = cov_out + (size_t)ci * out_stride * 9;
  if (tid == 0) {
    nm_s = 0;
    bad = 0;
    n_out[ci] = P + K;
    // ---- odometry chain (FastMarginals.cpp:201-223): sequential over the new poses
    Pose origin{ep[4 * (P - 1)], ep[4 * (P - 1) + 1], ep[4 * (P - 1) + 2], ep[4 * (P - 1) + 3]};  // parent->state.pose (estimate)
    Pose val0{thp[4 * (P - 1)], thp[4 * (P - 1) + 1], thp[4 * (P - 1) + 2], thp[4 * (P - 1) + 3]};  // values.at(key0)
    double cov0[9], Fc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) cov0[r * 3 + q] = Sig[(size_t)(3 * (P - 1) + r) * n + 3 * (P - 1) + q];
    const double sg[3] = {cfg.translation_noise, cfg.translation_noise, cfg.rotation_noise};
    for (int b = 0; b < K; ++b) {
      const double *a = actions + ((size_t)ci * S.A_max + b) * 3;
      const Pose odom = make_pose(a[0], a[1], a[2]);
      const Pose end = compose(origin, odom);
      newp[4 * b] = end.x; newp[4 * b + 1] = end.y; newp[4 * b + 2] = end.c; newp[4 * b + 3] = end.s;
      double H1b[9];
      const Pose hx = between(val0, end, H1b);
      const Pose h = between(odom, hx, nullptr);
      const double Hl[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
      double A0[9], A1i[9];  // whitened block on key0; inverse of the whitened block on key1 (Hl is a rotation: Hl^-1 = Hl^T)
      mat3_mul(Hl, H1b, A0);
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) {
          A0[r * 3 + q] /= sg[r];
          A1i[r * 3 + q] = Hl[q * 3 + r] * sg[q];
        }
      double t1[9], t2[9], c1[9], fb[9], fsn[9];
      mat3_mul(A0, cov0, t1);
      mat3_mul_bt(t1, A0, t2);
      t2[0] += 1.0; t2[4] += 1.0; t2[8] += 1.0;
      mat3_mul(A1i, t2, t1);
      mat3_mul_bt(t1, A1i, c1);
      mat3_mul(A1i, A0, fb);
      for (int q = 0; q < 9; ++q) fb[q] = -fb[q];
      mat3_mul(fb, Fc, fsn);
      for (int q = 0; q < 9; ++q) {
        cov_new[9 * b + q] = c1[q];
        F[9 * b + q] = fb[q];
        Fs[9 * b + q] = fsn[q];
        Fc[q] = fsn[q];
        cov0[q] = c1[q];
      }
      origin = end;
      val0 = end;
    }
  }
  __syncthreads();
  // Sigma(new a, new b) = cov_a F_{a+1}^T ... F_b^T
  for (int a = tid; a < K; a += kT) {
    double cur[9];
    for (int q = 0; q < 9; ++q) cur[q] = NN[((size_t)a * K + a) * 9 + q] = cov_new[9 * a + q];
    for (int b = a + 1; b < K; ++b) {
      double nx[9];
      mat3_mul_bt(cur, F + 9 * b, nx);
      for (int q = 0; q < 9; ++q) cur[q] = NN[((size_t)a * K + b) * 9 + q] = nx[q];
    }
  }
  // ---- predicted measurements (Planner2D.cpp:717-731): new pose b x estimated landmark j, in (b, j) order
  for (int b = 0; b < K; ++b) {
    const Pose ps{newp[4 * b], newp[4 * b + 1], newp[4 * b + 2], newp[4 * b + 3]};
    for (int j0 = 0; j0 < L; j0 += kT) {
      const int j = j0 + tid;
      bool ok = false;
      if (j < L) {
        const P2 lm{el[2 * j], el[2 * j + 1]};
        const double dx = lm.x - ps.x, dy = lm.y - ps.y, rng = sqrt(dx * dx + dy * dy);
        const double bearing = bearing_of<false>(ps, lm, nullptr, nullptr);
        ok = rng < cfg.max_range && rng > cfg.min_range && bearing < cfg.max_bearing && bearing > cfg.min_bearing;
      }
      // ordered compaction: wave ballots + a serial scan over the waves (4 waves)
      const unsigned long long bal = __ballot(ok);
      if ((tid & 63) == 0) wcount[tid >> 6] = __popcll(bal);
      __syncthreads();
      int base = nm_s;
      for (int w = 0; w < (tid >> 6); ++w) base += wcount[w];
      const int idx = base + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
      if (ok && idx < nm_max) {
        mk[idx] = b;
        mj[idx] = j;
        double Jx[6], Jl[4];
        br_jac(ps, P2{thl[2 * j], thl[2 * j + 1]}, Jx, Jl);  // landmark at its linearisation point
        double *o = J + (size_t)10 * idx;
        for (int q = 0; q < 3; ++q) {
          o[q] = Jx[q] / cfg.bearing_noise;
          o[3 + q] = Jx[3 + q] / cfg.range_noise;
        }
        o[6] = Jl[0] / cfg.bearing_noise; o[7] = Jl[1] / cfg.bearing_noise;
        o[8] = Jl[2] / cfg.range_noise; o[9] = Jl[3] / cfg.range_noise;
      }
      __syncthreads();
      if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < kT / 64; ++w) tot += wcount[w];
        nm_s += tot;
      }
      __syncthreads();
    }
  }
  if (nm_s > nm_max) {
    if (tid == 0) atomicMin(S.status, DRLGX_E_CAPACITY);
    return;
  }
  const int nm = nm_s, dim = 2 * nm;
  const Cand c{P, L, K, nm, Sig, n, cov_new, F, Fs, NN, mk, mj, J};
  // ---- T = I + A Sigma_A A^T, one thread per measurement pair (2x2 block)
  for (int e = tid; e < nm * nm; e += kT) {
    const int r = e / nm, s = e - r * nm;
    const double *jr = J + (size_t)10 * r, *js = J + (size_t)10 * s;
    const int kr = mk[r], ks = mk[s], lr = mj[r], ls = mj[s];
    double xx[9], xl[6], lx[6], ll[4];  // Sigma(x_kr, x_ks) 3x3, Sigma(x_kr, l_ls) 3x2, Sigma(l_lr, x_ks) 2x3, Sigma(l_lr, l_ls) 2x2
    cov_nn(c, kr, ks, xx);
    {
      double t[6];  // Sigma(l_ls, x_kr) (2x3) -> transpose
      cov_old_new(c, 3 * P + 2 * ls, 2, kr, t);
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 2; ++b) xl[a * 2 + b] = t[b * 3 + a];
    }
    cov_old_new(c, 3 * P + 2 * lr, 2, ks, lx);
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) ll[a * 2 + b] = Sig[(size_t)(3 * P + 2 * lr + a) * n + 3 * P + 2 * ls + b];
    // M = [Jx_r | Jl_r] [xx xl; lx ll] [Jx_s | Jl_s]^T
    double v[2][5];  // [Jx_r | Jl_r] * Sigma block: 2 x 5
    for (int a = 0; a < 2; ++a) {
      for (int q = 0; q < 3; ++q)
        v[a][q] = jr[3 * a] * xx[q] + jr[3 * a + 1] * xx[3 + q] + jr[3 * a + 2] * xx[6 + q] + jr[6 + 2 * a] * lx[q] + jr[7 + 2 * a] * lx[3 + q];
      for (int q = 0; q < 2; ++q)
        v[a][3 + q] = jr[3 * a] * xl[q] + jr[3 * a + 1] * xl[2 + q] + jr[3 * a + 2] * xl[4 + q] + jr[6 + 2 * a] * ll[q] + jr[7 + 2 * a] * ll[2 + q];
    }
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        const double m = v[a][0] * js[3 * b] + v[a][1] * js[3 * b + 1] + v[a][2] * js[3 * b + 2] + v[a][3] * js[6 + 2 * b] + v[a][4] * js[7 + 2 * b];
        T[(size_t)(2 * r + a) * dim + 2 * s + b] = m + ((r == s && a == b) ? 1.0 : 0.0);
      }
  }
  __syncthreads();
  if (dim > 0) spd_inverse(T, dim, dim, gj, gj + 512, &bad);
  __syncthreads();
  // ---- every pose i: U_i = A Sigma(., x_i) (dim x 3), Sigma'_ii = Sigma_ii - U_i^T T^-1 U_i.  One wave per pose.
  const int lane = tid & 63, wave = tid >> 6, nw = kT / 64;
  for (int i = wave; i < P + K; i += nw) {
    double *Ui = U + (size_t)wave * 6 * nm_max;
    for (int r = lane; r < nm; r += 64) {
      const double *jr = J + (size_t)10 * r;
      const int kr = mk[r], lr = mj[r];
      double xi[9], li[6];  // Sigma(x_new kr, x_i) 3x3, Sigma(l_lr, x_i) 2x3
      if (i >= P) {
        cov_nn(c, kr, i - P, xi);
        cov_old_new(c, 3 * P + 2 * lr, 2, i - P, li);
      } else {
        double t[9];  // Sigma(x_i, x_new kr) -> transpose
        cov_old_new(c, 3 * i, 3, kr, t);
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) xi[a * 3 + b] = t[b * 3 + a];
        for (int a = 0; a < 2; ++a)
          for (int b = 0; b < 3; ++b) li[a * 3 + b] = Sig[(size_t)(3 * P + 2 * lr + a) * n + 3 * i + b];
      }
      for (int a = 0; a < 2; ++a)
        for (int q = 0; q < 3; ++q)
          Ui[(size_t)(2 * r + a) * 3 + q] = jr[3 * a] * xi[q] + jr[3 * a + 1] * xi[3 + q] + jr[3 * a + 2] * xi[6 + q] + jr[6 + 2 * a] * li[q] + jr[7 + 2 * a] * li[3 + q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double acc[6] = {0, 0, 0, 0, 0, 0};  // U^T T^-1 U, symmetric
    for (int r = lane; r < dim; r += 64) {
      double w0 = 0, w1 = 0, w2 = 0;  // (T^-1 U)[r][:]
      const double *tr = T + (size_t)r * dim;
      for (int s = 0; s < dim; ++s) {
        w0 += tr[s] * Ui[3 * s];
        w1 += tr[s] * Ui[3 * s + 1];
        w2 += tr[s] * Ui[3 * s + 2];
      }
      const double u0 = Ui[3 * r], u1 = Ui[3 * r + 1], u2 = Ui[3 * r + 2];
      acc[0] += u0 * w0; acc[1] += u0 * w1; acc[2] += u0 * w2; acc[3] += u1 * w1; acc[4] += u1 * w2; acc[5] += u2 * w2;
    }
    for (int q = 0; q < 6; ++q)
      for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o);
    if (lane == 0) {
      double base[9];
      if (i >= P)
        for (int q = 0; q < 9; ++q) base[q] = cov_new[9 * (i - P) + q];
      else
        for (int r = 0; r < 3; ++r)
          for (int q = 0; q < 3; ++q) base[r * 3 + q] = Sig[(size_t)(3 * i + r) * n + 3 * i + q];
      double *o = out + (size_t)i * 9;
      o[0] = base[0] - acc[0]; o[1] = base[1] - acc[1]; o[2] = base[2] - acc[2];
      o[3] = base[3] - acc[1]; o[4] = base[4] - acc[3]; o[5] = base[5] - acc[4];
      o[6] = base[6] - acc[2]; o[7] = base[7] - acc[4]; o[8] = base[8] - acc[5];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (tid == 0 && bad) atomicMin(S.status, DRLGX_E_NUMERIC);
}

}  // namespace kfm2

void drlgx_launch_fm2_prior(const DrlgxState &S, hipStream_t st, const int32_t *env_ids, int n_env, double *sig, size_t sig_stride,
                            int n_max) {
  hipLaunchKernelGGL(kfm2::k_fm2_prior, dim3(n_env), dim3(kfm2::kT), 2 * (size_t)n_max * sizeof(double), st, S, env_ids, n_env, sig,
                     sig_stride);
}
size_t drlgx_fm2_scratch_doubles(const DrlgxState &S, int nm_max) {
  return (size_t)27 * S.A_max + (size_t)9 * S.A_max * S.A_max + (size_t)10 * nm_max + (size_t)4 * nm_max * nm_max +
         (size_t)(kfm2::kT / 64) * 6 * nm_max + (size_t)4 * S.A_max + 64;
}
void drlgx_launch_fm2_update(const DrlgxState &S, hipStream_t st, int c0, int nc, const int32_t *cand_env, const int32_t *env_slot,
                             const double *actions, const int32_t *n_actions, const double *sig, size_t sig_stride, double *scratch,
                             size_t scratch_stride, int *iscratch, int nm_max, double *cov_out, int out_stride, int32_t *n_out) {
  hipLaunchKernelGGL(kfm2::k_fm2_update, dim3(nc), dim3(kfm2::kT), 0, st, S, c0, cand_env, env_slot, actions, n_actions, sig, sig_stride,
                     scratch, scratch_stride, iscratch, nm_max, cov_out, out_stride, n_out);
}
