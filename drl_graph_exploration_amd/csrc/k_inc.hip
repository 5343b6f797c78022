// Incremental belief update between relinearisations: the SLAM stage as a rank-k covariance update instead of a re-solve.
// Included by k_slam.hip (namespace kslam, after SlamCtx): it shares linearize_br, the fp64-MFMA helpers and inv16_blk.
//
// Reference arithmetic: FastMarginals2::propagate / update (src/em_exploration/FastMarginals.cpp:188-321) - the covariance-form
// (EKF) update the reference's EM planner applies to candidate trajectories - applied here to SLAM2D::optimize
// (SLAM2D.cpp:374-430) itself.  One iSAM2 update (gtsam ISAM2, policy in SURVEY.md App. A.3) keeps the linearisation point of
// every variable it does not relinearise, so between relinearisations the normal equations only GAIN the new pose's
// odometry factor and this step's bearing-range factors: with Sigma = Lambda^-1 and delta = Sigma eta of the previous update,
//     new pose x' (odometry factor e0 + J1 d_x + J2 d_x', J2 orthogonal):   d_x' = F d_x + c,  F = -J2^T J1,  c = -J2^T e0
//                                                Sigma[x', .] = F Sigma[x, .],  Sigma[x', x'] = F Sigma[x, x] F^T + Q
//     re-observed landmarks (rows A = [Jx at x', Jl at l], noise R):         T = R + A Sigma A^T
//                                                Sigma' = Sigma - Sigma A^T T^-1 A Sigma,   d' = d + Sigma A^T T^-1 (-e - A d)
//     a landmark seen for the first time (square Jl):                          d_l = G d_x' + c,  G = -Jl^-1 Jx,  rows like the pose
// which is exactly the re-solve of the grown system (scripts/emul/inc_update_emul.py holds the numpy form and the measured
// drift against the CPU oracle: estimates 1e-12, information blocks 3 % of the parity tolerance over 120-step runs).
//
// Only part of Sigma can ever be touched again: a factor always joins the NEWEST pose and a landmark, old poses are never
// revisited.  The state kept per instance is therefore the "panel" Sigma[:, active] - every variable (rows) against the active
// set (columns: the current pose, all landmarks) - and the 3x3 marginal of every pose: (3P + 2L) x (3 + 2L) + 6P doubles
// (59 KB at 37 poses / 22 landmarks against 97 KB for the packed joint covariance), DrlgxState::jc / jd / jc_meta.  All steps
// are row-local except the k x k system T (k = 2 x re-observed landmarks, in batches of <= 8 landmarks = one 16 x 16 tile inverted in
// registers by inv16_blk) and the rank-k update itself, which runs on the fp64 matrix cores:
//     U'^T = W' Y^T,  C <- C + U' Ya^T         (Y = Sigma A^T row by row, W' = -T^-1, Ya = the Y rows of the active variables)
// The panel lives in LDS for the step when it fits (flat pointers: the same code updates it in place in HBM / L2 otherwise).
//
// Who may take this path is decided per instance and per update (inc_precheck / inc_body's own checks): a valid panel, exactly
// one new pose since it was left, every new factor on that pose, no landmark twice, and no relinearisation due (update count
// % 10 == 0 with some |delta| >= 0.1).  Everything else - and every update that relinearises - runs the full solve, which
// leaves a fresh panel behind: panel_from_dense after the dense solver (<= 42 poses), the pose-chain solver of longer
// trajectories through its own three extra right-hand sides (k_slam_arrow.hip: mk_panel).  k_reset and k_rebase invalidate it.

// Contraction is decided in the front end here (a * b + c written in one expression becomes an fma, nothing else does): with
// contract(fast) the back end fuses differently in the fused step kernel and in the stage kernel, and the two must agree bit
// for bit (test_fused_step_kernel_equals_stage_kernels).
#pragma clang fp contract(on)

constexpr int IYS = 18;  // row stride (doubles) of the Y / U' / W' images: 16 columns in ks16 order + 2 pad (144 B)
constexpr int INF = 64;  // most bearing-range factors one step may add on this path
#ifndef INC_STREAM_KB
#define INC_STREAM_KB 512
#endif
#ifndef INC_ISNT
#define INC_ISNT 4
#endif
constexpr int ISNT = INC_ISNT;  // the streamed form of the update (panel in HBM / L2): at most 8 ISNT re-observed landmarks per walk over the panel

__device__ __forceinline__ int *inc_meta(const DrlgxState &S, int inst) { return S.jc_meta + (size_t)inst * 4; }

// Can the update that brings instance `inst` to P_after poses be incremental?  Uniform over the workgroup; every thread calls.
// scratch: one int of LDS nobody else uses at this point (the library's __syncthreads_or brings 256 B of STATIC LDS with it,
// which on top of the 160 KB of dynamic LDS these kernels request makes the launch fail).
// (meta0 / meta1 / isam: the panel's valid flag and pose count and the update counter when the caller has loaded them already -
// k_step's prelude issues them together with the instance's counts: one round trip to HBM before the simulator wave starts
// instead of two; meta0 < 0: loaded here)
__device__ __forceinline__ bool inc_precheck(const DrlgxState &S, int inst, int P_after, int tid, int *scratch, int meta0 = -1, int meta1 = 0,
                                             int isam = 0) {
  if (!S.jc) return false;
  const int *meta = inc_meta(S, inst);
  const int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  if (meta0 < 0) {
    meta0 = meta[0];
    meta1 = meta[1];
    isam = cnt[C_ISAM];
  }
  if (meta0 != 1 || meta1 + 1 != P_after || P_after < 2) return false;
  const int count = isam + 1;
  if (count % 10 != 0) return true;
  // relinearizeSkip = 10: the variables that existed at the previous update are checked against relinearizeThreshold = 0.1
  const double *d_pose = S.d_pose + (size_t)inst * S.P_max * 3;
  const double *d_lm = S.d_lm + (size_t)inst * S.L_max * 2;
  int any = 0;
  for (int k = tid; k < 3 * meta[1]; k += kThreads) any |= fabs(d_pose[k]) >= 0.1;
  for (int k = tid; k < 2 * meta[2]; k += kThreads) any |= fabs(d_lm[k]) >= 0.1;
  if (tid == 0) *scratch = 0;
  __syncthreads();
  if (any) atomicOr(scratch, 1);
  __syncthreads();
  const int r = *scratch;
  __syncthreads();
  return r == 0;
}

// value of the lane 16 or 32 lanes away (lane ^ kDist) - the other 16-lane rows of the wave - by the gfx950 permlane swaps
template <int kDist>
__device__ __forceinline__ double rowgroup_xor(double v) {
  const long long b = __double_as_longlong(v);
  unsigned w[2] = {(unsigned)b, (unsigned)(b >> 32)};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if constexpr (kDist == 16) {
      const auto p = __builtin_amdgcn_permlane16_swap(w[h], w[h], false, false);  // [x0 x0 x2 x2], [x1 x1 x3 x3]
      // rows 0 and 2 want their odd neighbour (x1, x3), rows 1 and 3 their even one (x0, x2)
      w[h] = ((threadIdx.x >> 4) & 1) ? p[0] : p[1];
    } else {
      const auto p = __builtin_amdgcn_permlane32_swap(w[h], w[h], false, false);  // [x0 x1 x0 x1], [x2 x3 x2 x3]
      w[h] = ((threadIdx.x >> 5) & 1) ? p[0] : p[1];
    }
  }
  return __longlong_as_double(((long long)w[1] << 32) | w[0]);
}

// value of the lane kShift (1 or 2) lanes below inside the 16-lane row (row_shr); lanes without such a neighbour read 0
template <int kShift>
__device__ __forceinline__ double row_shr_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x110 + kShift, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x110 + kShift, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// LDS carve of one incremental update.  Sizes follow what is known BEFORE the step (the pose count after it, the landmark
// count L0 of the previous update) plus room for up to INEW landmarks seen for the first time, so that the first half of the
// update can run beside the simulator wave of the fused step kernel.
constexpr int INEW = 12;
struct IncCtx {
  int inst, P, pn, pp, L0, M0, Lcap, n1, n1p, a0, ldw;
  int *fre, *fnew, *fslot, *ictl;
  double *fq, *recs, *gs, *vvec, *wks, *thp, *thl, *dl, *Dl, *Y, *Yh, *cwl;
  bool wide;  // room for the second half of Y: batches of up to 16 re-observed landmarks
  // the panel in HBM / L2 (inc_stream_batch): up to 8 snt re-observed landmarks per walk over the panel; LDS images
  int snt, yrows;
  double *yt, *wim, *jt, *svv;
};
// false: the step cannot take this path (LDS).  lds_panel: the panel is staged in LDS for the step (else updated in place in
// HBM / L2).  The decision is taken for the SAME LDS offset in every kernel (the fused step's: behind the simulator's region),
// so that the fused kernel and the stage kernels always run the same instantiation.
// pc: the launch's pose bound (LaunchSel::cap): the fused step's simulator region is sized by it.
// L0_known / M0_known (>= 0): the panel's landmark and factor counts as the caller loaded them already (k_step's prelude): the plan
// is then pure arithmetic - no load from HBM behind the stores the fused step has in flight when it plans the second half.
__device__ __forceinline__ bool inc_plan(const DrlgxState &S, int inst, int P, int lds_bytes, size_t smem_off, IncCtx &x, bool &lds_panel, int pc,
                                         int L0_known = -1, int M0_known = -1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int *meta = inc_meta(S, inst);
  x.inst = inst; x.P = P; x.pn = P - 1; x.pp = P - 2;
  x.L0 = L0_known >= 0 ? L0_known : meta[2];
  x.M0 = L0_known >= 0 ? M0_known : meta[3];
  x.Lcap = min(S.L_max, x.L0 + INEW);
  x.n1 = 3 * P + 2 * x.L0; x.n1p = (x.n1 + 15) & ~15;
  x.a0 = 3 + 2 * x.L0;
  x.ldw = (3 + 2 * x.Lcap + 31) & ~31;  // (whole pairs of 16-column tiles: inc_post, B3)
  const int ncap = max(3 * P + 2 * x.Lcap, x.n1p);
  size_t plan_off = drlgx_sim_lds_bytes(S.LG, pc);
  // (no fused kernel ever serves this state - more poses than the dense solver takes and more landmarks than k_step_arrow sweeps in
  // LDS, drlgx_step_arrow_fusable -: nothing to agree with, the simulator's region is not set aside)
  if ((3 * P + 1 + 15) / 16 > kDenseTiles && 2 * S.L_max + 1 > 16 * 8) plan_off = 0;
  if (smem_off > plan_off) plan_off = smem_off;
  size_t off = (smem_off + 15) & ~(size_t)15;
  const size_t off0 = off;
  auto take = [&](size_t bytes) { unsigned char *q = smem_raw + off; off += (bytes + 15) & ~(size_t)15; return q; };
  x.fre = reinterpret_cast<int *>(take(INF * 4));    // re-observed: index (0 .. nf-1) of the factor, in factor order
  x.fnew = reinterpret_cast<int *>(take(INF * 4));   // new landmark L0 + j: index of its factor
  x.fslot = reinterpret_cast<int *>(take(INF * 4));  // landmark slot of factor t
  x.ictl = reinterpret_cast<int *>(take(16));        // [0] re-observed count, [1] numerical flag (inv16_blk), [2] structure flag
  x.fq = reinterpret_cast<double *>(take(32 * 8));            // F (9), c (3), theta of the new pose (4), measured odometry (4)
  x.recs = reinterpret_cast<double *>(take(INF * REC * 8));   // linearised new factors
  x.gs = reinterpret_cast<double *>(take(INEW * 12 * 8));     // per new landmark: G (6), c (2), Q (xx xy yy), pad
  x.vvec = reinterpret_cast<double *>(take(32 * 8));
  x.wks = reinterpret_cast<double *>(take(5 * 16 * IYS * 8));  // W' as operand images: one tile, or four + a transposition scratch
  x.thp = reinterpret_cast<double *>(take((size_t)P * 4 * 8));
  x.thl = reinterpret_cast<double *>(take((size_t)x.Lcap * 2 * 8));
  x.dl = reinterpret_cast<double *>(take((size_t)ncap * 8));  // delta, logical row order
  x.Dl = reinterpret_cast<double *>(take((size_t)P * 6 * 8));
  x.Y = reinterpret_cast<double *>(take((size_t)(x.n1p + 1) * IYS * 8));  // (+ one row of zeros: the pad columns' operand)
  x.cwl = reinterpret_cast<double *>(smem_raw + off);
  const size_t fixed = ((plan_off + 15) & ~(size_t)15) + (off - off0);
  if (fixed > (size_t)lds_bytes) return false;
  lds_panel = fixed + (size_t)ncap * x.ldw * 8 <= (size_t)lds_bytes;
  // the second half of Y (wide batches) sits behind the panel, if there is room left
  const size_t pan = lds_panel ? (size_t)ncap * x.ldw * 8 : 0, yh = (size_t)(x.n1p + 1) * IYS * 8;
  x.Yh = x.cwl + pan / 8;
  x.wide = fixed + pan + yh <= (size_t)lds_bytes;
  x.snt = 0;
  // (the dense solver's kernels - k_step, k_slam: <= 53 poses - keep the two-walk form for the panels that miss the LDS: those are a
  // few hundred KB that stay in L2, and the streamed form's fixed costs - a k x k inverse, the gathers at the head of every row tile -
  // made the 48-landmark instance of the bench state 6 us slower)
  if (!lds_panel && (3 * P + 1 + 15) / 16 > kDenseTiles) {
    // The panel stays in HBM / L2: the update walks it ONCE per batch of up to 8 snt landmarks (inc_stream_batch), and what it
    // keeps in LDS is indexed by the panel's COLUMNS, not its rows: Y^T of the active variables as snt tiles of 16 factor columns
    // (rows = the columns of the panel + one row of zeros for the pad columns), the snt x snt operand images of
    // W' = -T^-1 and the batch's Jacobians in column order - everything from x.Y to the end of the LDS.
    x.yrows = x.a0 + 1;  // (+ one row of zeros: the operand of the pad columns)
    const size_t y_off = (size_t)(reinterpret_cast<unsigned char *>(x.Y) - smem_raw);
    // (sized like everything else in this plan: as if the carve began at plan_off - the fused step and the stage kernels agree)
    const size_t avail = (size_t)lds_bytes - (((plan_off + 15) & ~(size_t)15) + (y_off - off0));
    for (int t = ISNT; t >= 1 && !x.snt; --t)
      if ((size_t)t * x.yrows * IYS * 8 + (size_t)max(0, t * t - 5) * 16 * IYS * 8 + (size_t)16 * t * 8 * 8 + 64 * 8 <= avail) x.snt = t;
    x.yt = x.Y;
    x.wim = x.yt + (size_t)x.snt * x.yrows * IYS;  // images 5 .. snt^2 - 1 of W' (the first five: x.wks)
    x.jt = x.wim + (size_t)max(0, x.snt * x.snt - 5) * 16 * IYS;
    x.svv = x.jt + (size_t)16 * x.snt * 8;
  }
  return true;
}

// First half: everything that does not depend on this step's measurements - loads, the new pose (odometry factor between the
// last two poses, SLAM2D.cpp:59-89).  kSub: called by the threads 64 .. kThreads-1 while wave 0 simulates (k_step); the new
// pose's theta is then formed from the commanded odometry with the expressions the simulator evaluates (as SlamCtx::front).
// kLds: the panel is staged in LDS (two instantiations rather than one body on generic pointers: flat accesses count on both
// memory counters, so every LDS read waited for all panel loads in flight).
template <bool kLds, bool kSub>
__device__ __forceinline__ void inc_pre(const DrlgxState &S, const IncCtx &x, int tid, const double *odom3, SubBarrier sb) {
  const int ft = kSub ? tid - 64 : tid, fn = kSub ? kThreads - 64 : kThreads, lane = tid & 63;
  auto bar = [&]() {
    if constexpr (kSub) sb.sync(lane);
    else __syncthreads();
  };
  const int inst = x.inst, P = x.P, pn = x.pn, pp = x.pp, L0 = x.L0, n1 = x.n1, a0 = x.a0;
  double *gpan = S.jc + (size_t)inst * S.jc_stride;
  auto growp = [&](int q) -> double * { return gpan + (size_t)(q < 3 * P ? q : q + 3 * (S.P_max - P)) * S.jc_ld; };
  auto rowp = [&](int q) -> double * {
    if constexpr (kLds) return x.cwl + (size_t)q * x.ldw;
    else return gpan + (size_t)(q < 3 * P ? q : q + 3 * (S.P_max - P)) * S.jc_ld;
  };
  const double *th_pose = S.th_pose + (size_t)inst * S.P_max * 4;
  const double *th_lm = S.th_lm + (size_t)inst * S.L_max * 2;
  const double *d_pose = S.d_pose + (size_t)inst * S.P_max * 3;
  const double *d_lm = S.d_lm + (size_t)inst * S.L_max * 2;
  const double *jd = S.jd + (size_t)inst * S.P_max * 6;
  if (S.prof && blockIdx.x == S.prof_block && ft == 0) S.prof[0] = wall_clock64();
  if (ft < 4) x.ictl[ft] = 0;
  // the thread that linearises the odometry factor starts with its own loads
  if (ft == fn - 1) {
    const double *t1 = th_pose + 4 * pp;
    Pose p1{t1[0], t1[1], t1[2], t1[3]}, p2, om;
    if constexpr (kSub) {
      om = make_pose(odom3[0], odom3[1], odom3[2]);
      const double *ep = S.est_pose + ((size_t)inst * S.P_max + pp) * 4;
      p2 = compose(Pose{ep[0], ep[1], ep[2], ep[3]}, om);  // SLAM2D::addOdometry's initial guess (SLAM2D.cpp:70-89)
    } else {
      const double *t2 = th_pose + 4 * pn, *oo = S.odo + ((size_t)inst * S.P_max + pp) * 4;
      p2 = Pose{t2[0], t2[1], t2[2], t2[3]};
      om = Pose{oo[0], oo[1], oo[2], oo[3]};
    }
    // e = Local(measured, between(x1, x2)), J2 = Hlocal (a rotation), J1 = Hlocal H1  =>  F = -J2^T J1 = -H1, c = -J2^T e
    double H1[9];
    const Pose hx = between(p1, p2, H1);
    const Pose h = between(om, hx, nullptr);
    const double e0 = h.x, e1 = h.y, e2 = theta_of(h);
    double *fq = x.fq;
    for (int k = 0; k < 9; ++k) fq[k] = -H1[k];
    fq[9] = -(h.c * e0 - h.s * e1);
    fq[10] = -(h.s * e0 + h.c * e1);
    fq[11] = -e2;
    x.thp[4 * pn] = p2.x; x.thp[4 * pn + 1] = p2.y; x.thp[4 * pn + 2] = p2.c; x.thp[4 * pn + 3] = p2.s;
  }
  for (int e = ft; e < 4 * pn; e += fn) x.thp[e] = th_pose[e];
  for (int e = ft; e < 2 * L0; e += fn) x.thl[e] = th_lm[e];
  for (int e = ft; e < 3 * pn; e += fn) x.dl[e] = d_pose[e];
  for (int e = ft; e < 2 * L0; e += fn) x.dl[3 * P + e] = d_lm[e];
  for (int e = ft; e < 6 * pn; e += fn) x.Dl[e] = jd[e];
  if constexpr (kLds) {
    // (the rows of the new pose and of the new landmarks are made below; columns [0, a0) are live.)  32 threads x 16 bytes per
    // row, four rows' loads in flight per thread before the first store
    const int npair = (a0 + 1) >> 1, nr = n1 - 3;
    const int cp0 = ft & 31, r0 = ft >> 5, rs = fn >> 5;
    for (int cp = cp0; cp < npair; cp += 32)
      for (int qq = r0; qq < nr; qq += 4 * rs) {
        // (clamped row indices: the loads are unconditional, the stores are not)
        const int q0 = qq, q1 = qq + rs, q2 = qq + 2 * rs, q3 = qq + 3 * rs;
        auto src = [&](int q) { const int c = min(q, nr - 1); return reinterpret_cast<const double2 *>(growp(c < 3 * pn ? c : c + 3))[cp]; };
        const double2 v0 = src(q0), v1 = src(q1), v2 = src(q2), v3 = src(q3);
        auto dst = [&](int q, const double2 &v) { if (q < nr) reinterpret_cast<double2 *>(rowp(q < 3 * pn ? q : q + 3))[cp] = v; };
        dst(q0, v0); dst(q1, v1); dst(q2, v2); dst(q3, v3);
      }
  }
  for (int e = ft; e < (x.n1p + 1 - n1) * IYS; e += fn) {  // pad rows + the zero row
    x.Y[(size_t)n1 * IYS + e] = 0.0;
    if (x.wide) x.Yh[(size_t)n1 * IYS + e] = 0.0;
  }
  bar();
  if (S.prof && blockIdx.x == S.prof_block && ft == 0) S.prof[1] = wall_clock64();
  // ---- A. the new pose: every row's covariance with the current pose moves through F; the new pose's own rows ----
  const double *fq = x.fq;
  {
    const double f00 = fq[0], f01 = fq[1], f02 = fq[2], f10 = fq[3], f11 = fq[4], f12 = fq[5], f20 = fq[6], f21 = fq[7], f22 = fq[8];
    for (int qq = ft; qq < n1 - 3; qq += fn) {
      const int q = qq < 3 * pn ? qq : qq + 3;
      double *r = rowp(q);
      const double t0 = r[0], t1 = r[1], t2 = r[2];
      r[0] = f00 * t0 + f01 * t1 + f02 * t2;
      r[1] = f10 * t0 + f11 * t1 + f12 * t2;
      r[2] = f20 * t0 + f21 * t1 + f22 * t2;
    }
  }
  bar();
  for (int e = ft; e < 3 * a0; e += fn) {
    const int r = e / a0, c = e - r * a0;
    double v = fq[3 * r] * rowp(3 * pp)[c] + fq[3 * r + 1] * rowp(3 * pp + 1)[c] + fq[3 * r + 2] * rowp(3 * pp + 2)[c];
    if (c == r) v += 1.0 / (c < 2 ? S.w_trans : S.w_rot);  // Q = J2^T W^-1 J2 = diag(sigma_t^2, sigma_t^2, sigma_r^2)
    rowp(3 * pn + r)[c] = v;
  }
  if (ft < 3) x.dl[3 * pn + ft] = fq[3 * ft] * x.dl[3 * pp] + fq[3 * ft + 1] * x.dl[3 * pp + 1] + fq[3 * ft + 2] * x.dl[3 * pp + 2] + fq[9 + ft];
  if (S.prof && blockIdx.x == S.prof_block && ft == 0) S.prof[2] = wall_clock64();
}

// Second half (all kThreads threads, after the simulator and a workgroup barrier): this step's factors.  L, M: the final
// counts.  box: what the simulator wave left in LDS (k_step), else read from the instance's arrays in HBM.  The structure of
// the new factors is verified either way (the staged C ABI can append anything; the fused step's own simulator cannot, but
// it runs the same code).  Returns false when the step does not fit this path after all: the caller runs the full solve
// (the HBM panel, if inc_pre already moved it, is marked invalid); uniform over the workgroup.
// hand (k_step; or null): LDS that receives what the map stage reads next - est_pose [P][4] and, at hand + 4 hand_cap,
// pose_info [P][6] (as SlamCtx::back leaves them) - and the landmark estimates are left in x.thl for the same reason.
template <bool kLds, int kSNT = ISNT>
__device__ __forceinline__ bool inc_post(const DrlgxState &S, const IncCtx &x, int L, int M, const SimBox &box, int tid,
                                         double *hand = nullptr, int hand_cap = 0) {
  const int lane = tid & 63, wave = tid >> 6, lc = lane & 15, lr = lane >> 4;
  const int inst = x.inst, P = x.P, pn = x.pn, L0 = x.L0, M0 = x.M0, n1 = x.n1, n1p = x.n1p, a0 = x.a0;
  int *meta = inc_meta(S, inst);
  const int nf = M - M0, nn = L - L0;
  const int n = 3 * P + 2 * L, a = 3 + 2 * L;
  int *fre = x.fre, *fnew = x.fnew, *fslot = x.fslot, *ictl = x.ictl;
  double *recs = x.recs, *gs = x.gs, *vvec = x.vvec, *wks = x.wks, *thp = x.thp, *thl = x.thl, *dl = x.dl, *Dl = x.Dl, *Y = x.Y;
  double *gpan = S.jc + (size_t)inst * S.jc_stride;
  auto growp = [&](int q) -> double * { return gpan + (size_t)(q < 3 * P ? q : q + 3 * (S.P_max - P)) * S.jc_ld; };
  auto rowp = [&](int q) -> double * {
    if constexpr (kLds) return x.cwl + (size_t)q * x.ldw;
    else return gpan + (size_t)(q < 3 * P ? q : q + 3 * (S.P_max - P)) * S.jc_ld;
  };
  int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  double *d_pose = S.d_pose + (size_t)inst * S.P_max * 3;
  double *d_lm = S.d_lm + (size_t)inst * S.L_max * 2;
  double *jd = S.jd + (size_t)inst * S.P_max * 6;
  const bool feasible = nf >= 0 && nf <= INF && nn >= 0 && nn <= nf && L <= x.Lcap;
  // ---- 0. the new factors: structure, lists, linearisation at (theta of the new pose, theta of the landmark) ----
  const int *slot_src = box.br ? box.slot : S.meas_lm + (size_t)inst * S.M_max + M0;
  const double *br_src = box.br ? box.br : S.meas_br + ((size_t)inst * S.M_max + M0) * 2;
  const double *nl_src = box.br ? box.lm : S.th_lm + ((size_t)inst * S.L_max + L0) * 2;
  if (feasible && tid < 64) {  // (nf <= 64: one wave ranks them in factor order)
    const bool have = tid < nf;
    const int slot = have ? slot_src[tid] : -1;
    if (have) fslot[tid] = slot;
    bool bad = have && (slot < 0 || slot >= L);
    if (have && !box.br) bad |= S.meas_pose[(size_t)inst * S.M_max + M0 + tid] != pn;
    for (int t = 0; t < nf; ++t) {  // a landmark twice in one step: not this path
      const int st = __builtin_amdgcn_readlane(slot, t);
      bad |= have && t < tid && st == slot;
    }
    const bool re = have && slot < L0, nw = have && slot >= L0;
    const unsigned long long mre = __ballot(re), mnw = __ballot(nw);
    if (re) fre[__popcll(mre & ((1ull << lane) - 1ull))] = tid;
    if (nw && !bad) fnew[slot - L0] = tid;
    if ((__ballot(bad) || __popcll(mnw) != nn) && lane == 0) ictl[2] = 1;
    if (lane == 0) ictl[0] = __popcll(mre);
  }
  for (int e = tid - 64; e >= 0 && e < 2 * nn && feasible; e += kThreads - 64) thl[2 * L0 + e] = nl_src[e];
  __syncthreads();
  if (!feasible || ictl[2]) {
    if (!kLds && tid == 0) meta[0] = 0;
    return false;
  }
  const int n_re = ictl[0];
  if (tid < nf) linearize_br(thp + 4 * pn, thl + 2 * fslot[tid], br_src[2 * tid], br_src[2 * tid + 1], recs + (size_t)REC * tid);
  __syncthreads();
  DRLGX_PROF(S, 35);
  // ---- B. the re-observed landmarks, <= 8 at a time (independent measurement noise: sequential batches are exact) ----
  const double Rb = S.cfg.bearing_noise * S.cfg.bearing_noise, Rr = S.cfg.range_noise * S.cfg.range_noise;
  const int ntr = n1p >> 4, ntc = (a0 + 15) >> 4;
  // One batch of nb re-observed landmarks starting at list position b0.  kWide: up to 16 landmarks (k = 2 nb <= 32 columns, the
  // k x k system inverted by 2 x 2 blocks of 16 x 16 tiles) instead of up to 8 - one walk over the panel instead of two for the
  // steps that re-observe 9 .. 16 landmarks; needs the second half of Y (x.Yh) in LDS.
  auto batch = [&](int b0, int nb, auto wide_tag) {
    constexpr bool kWide = decltype(wide_tag)::value;
    const int k = 2 * nb, kD = kWide ? k - 16 : 0;  // (kWide: the first 16 columns are all live, kD of the second 16)
    double *Yh = x.Yh;
    // B1. Y = Sigma A^T, row by row: a thread serves ONE factor of the batch (its Jacobians in registers) for every 64th (32nd) row
    {
      constexpr int FS = kWide ? 16 : 8, RS = kThreads / FS;
      const int f = tid & (FS - 1);
      double j0 = 0, j1 = 0, j2 = 0, j3 = 0, j4 = 0, j5 = 0, l0c = 0, l1c = 0, l2c = 0, l3c = 0;
      int cl = 3;
      if (f < nb) {
        const int t = fre[b0 + f];
        const double *rc = recs + (size_t)REC * t;
        j0 = rc[0]; j1 = rc[1]; j2 = rc[2]; j3 = rc[3]; j4 = rc[4]; j5 = rc[5];
        l0c = rc[6]; l1c = rc[7]; l2c = rc[8]; l3c = rc[9];
        cl = 3 + 2 * fslot[t];
      }
      double *Yf = (kWide && f >= 8) ? Yh : Y;
      const int k0 = ks16(2 * (f & 7)), k1 = ks16(2 * (f & 7) + 1);
      // (four rows' loads in flight per thread: in the HBM / L2 form of the panel every load is a round trip)
      for (int q0 = tid / FS; q0 < n1; q0 += 4 * RS) {
        double c0[4], c1[4], c2[4], l0[4], l1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double *r = rowp(min(q0 + u * RS, n1 - 1));
          c0[u] = r[0]; c1[u] = r[1]; c2[u] = r[2]; l0[u] = r[cl]; l1[u] = r[cl + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + u * RS;
          if (q < n1) {  // (factors beyond the batch: all-zero Jacobians, zero columns)
            Yf[(size_t)q * IYS + k0] = c0[u] * j0 + c1[u] * j1 + c2[u] * j2 + l0[u] * l0c + l1[u] * l1c;
            Yf[(size_t)q * IYS + k1] = c0[u] * j3 + c1[u] * j4 + c2[u] * j5 + l0[u] * l2c + l1[u] * l3c;
          }
        }
      }
    }
    __syncthreads();
    if (b0 == 0) DRLGX_PROF(S, 36);
    // B2. T = R + A Y (symmetric by construction), v = -e - A delta, W' = -T^-1: one wave
    if (wave == 0) {
      // entry (i, j), i >= j, of T: row i = (factor i / 2, bearing or range), column j of Y (in the half j / 16)
      auto tent = [&](int i, int j) -> double {
        const int t = fre[b0 + (i >> 1)], u = i & 1;
        const double *rc = recs + (size_t)REC * t;
        const double *Yj = (kWide && j >= 16) ? Yh : Y;
        const int col = ks16(j & 15);
        const double *yp = Yj + (size_t)(3 * pn) * IYS + col, *yl = Yj + (size_t)(3 * P + 2 * fslot[t]) * IYS + col;
        double sv = rc[3 * u] * yp[0] + rc[3 * u + 1] * yp[IYS] + rc[3 * u + 2] * yp[2 * IYS] + rc[6 + 2 * u] * yl[0] + rc[7 + 2 * u] * yl[IYS];
        if (i == j) sv += u ? Rr : Rb;
        return sv;
      };
      if (lane < (kWide ? 32 : 16)) {
        double v = 0.0;
        if (lane < k) {
          const int t = fre[b0 + (lane >> 1)], u = lane & 1;
          const double *rc = recs + (size_t)REC * t;
          const double *dp = dl + 3 * pn, *dq = dl + 3 * P + 2 * fslot[t];
          v = -rc[10 + u] - (rc[3 * u] * dp[0] + rc[3 * u + 1] * dp[1] + rc[3 * u + 2] * dp[2] + rc[6 + 2 * u] * dq[0] + rc[7 + 2 * u] * dq[1]);
        }
        vvec[(lane & 16) + ks16(lane & 15)] = v;  // (ks16 order per half, like the rows of Y)
      }
      const int kA = kWide ? 16 : k;
      v4d d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lr + 4 * r, hi = max(i, lc), lo = min(i, lc);
        if (hi < kA) d[r] = tent(hi, lo);
      }
      const SweepCtx sx{0, lane, lc, lr, kA, 16, true, true, ictl + 1, nullptr};
      if constexpr (!kWide) {
        inv16_blk<true>(sx, kA, d);
        // image of W' for the matrix cores: row lc, columns lr + 4 r at ks16 positions 4 lr + r (W' is symmetric)
        double *o = wks + lc * IYS + 4 * lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (lc < k && lr + 4 * r < k) ? d[r] : 0.0;
      } else {
        // T = [A B; B^T D] in 16 x 16 tiles (accumulator layout: lane (lr, lc), register r = element (lr + 4 r, lc)):
        //   A' = -A^-1,  X' = A' B,  S = D + B^T X',  S' = -S^-1,   W' = -T^-1 = [A' + X' S' X'^T, X' S'; S' X'^T, S']
        // Accumulator registers used as the A operand are the TRANSPOSE of the tile, as the B operand the tile itself; the one
        // transposition that is not free (X'^T as a B operand) goes through LDS.
        v4d tb = {0.0, 0.0, 0.0, 0.0}, td = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = lr + 4 * r, hi = max(i, lc), lo = min(i, lc);
          if (lc < kD) tb[r] = tent(16 + lc, i);          // B[i][lc] = T[i][16 + lc]
          if (hi < kD) td[r] = tent(16 + hi, 16 + lo);    // D
        }
        inv16_blk<false>(sx, 16, d);  // A'
        double a4[4] = {d[0], d[1], d[2], d[3]}, b4[4] = {tb[0], tb[1], tb[2], tb[3]};
        v4d xp = {0.0, 0.0, 0.0, 0.0};
        xp = mfma4(a4, b4, xp);       // X' = A' B  (A' symmetric)
        double x4[4] = {xp[0], xp[1], xp[2], xp[3]};
        td = mfma4(b4, x4, td);       // S = D + B^T X'
        const SweepCtx sd{0, lane, lc, lr, kD, 16, true, true, ictl + 1, nullptr};
        inv16_blk<true>(sd, kD, td);  // S'
        double *tsc = wks + 4 * 16 * IYS;  // X' row-major (ks16 columns): read back as the image whose B-operand use is X'^T
#pragma unroll
        for (int r = 0; r < 4; ++r) tsc[(lr + 4 * r) * IYS + ks16(lc)] = xp[r];
        wave_lds_sync();
        double xt4[4], s4[4];
        ld4(tsc + lc * IYS + 4 * lr, xt4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s4[r] = (lc < kD && lr + 4 * r < kD) ? td[r] : 0.0;  // (S' with its inactive part zeroed)
        v4d w10 = {0.0, 0.0, 0.0, 0.0};
        w10 = mfma4(s4, xt4, w10);    // W'[1][0] = S' X'^T   (rows: second half, columns: first half)
        double w10a[4] = {w10[0], w10[1], w10[2], w10[3]};
        d = mfma4(w10a, xt4, d);      // W'[0][0] = A' + (S' X'^T)^T X'^T = A' + X' S' X'^T
        // operand images (row lc, ks16 columns): [0] W00, [1] W01 = W10^T, [2] W10, [3] W11
        double *o0 = wks + lc * IYS + 4 * lr, *o1 = o0 + 16 * IYS, *o3 = o0 + 3 * 16 * IYS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o0[r] = d[r];      // (symmetric: the registers are the image)
          o1[r] = w10[r];    // (the registers of a tile, stored as they are, are the image of its transpose)
          o3[r] = s4[r];
          wks[2 * 16 * IYS + (lr + 4 * r) * IYS + ks16(lc)] = w10[r];  // W10 itself: element (lr + 4 r, lc)
        }
      }
    }
    __syncthreads();
    if (b0 == 0) DRLGX_PROF(S, 37);
    // B3. U'^T = W' Y^T per 16-row tile; delta and the pose marginals; every column tile of the panel: C += U' Ya^T.
    // A wave owns the row tiles I = wave, wave + 8, ...; the columns are walked in PAIRS of 16-column tiles.  The wave's
    // (row tile, pair) units form ONE software pipeline: the loads of the next unit are issued before the products of the
    // current one (the panel may live in HBM / L2: a unit that loads only after the previous one's stores pays a full
    // round trip each time).  Loads and stores are unconditional: the rows up to the next multiple of 16 and the columns up
    // to the next multiple of 32 exist (the future landmarks' rows / columns), their operands are zero rows of Y: they are
    // written back unchanged.
    if (wave < ntr) {
      const int npr = (a0 + 31) >> 5;                  // column pairs
      const int nks = kWide ? 4 : (k + 3) >> 2;        // K steps with live columns (k = 2 nb; kWide: of the first half)
      const int nks1 = kWide ? (kD + 3) >> 2 : 0;      // ... of the second half
      const int nun = ((ntr - wave + kWaves - 1) / kWaves) * npr;
      // the two accumulator tiles of a unit, twice (current / next): element r = row 16 I + lr + 4 r, columns 32 p + lc and
      // 32 p + 16 + lc - loaded straight into the registers the matrix cores accumulate in (16-byte accesses of adjacent
      // column pairs were tried: the two halves belong to different accumulator tuples, the compiler copies them apart right
      // behind the load and waits for it there - no pipeline left)
      v4d aA0, aA1, aB0, aB1;
      double ua[4] = {0.0, 0.0, 0.0, 0.0}, ub[4] = {0.0, 0.0, 0.0, 0.0};  // U' of the row tile: columns 0 .. 15 / 16 .. 31
      auto loads = [&](int e, v4d &c0, v4d &c1) {
        const int it = e / npr, pr = e - it * npr, I = wave + kWaves * it;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#ifdef INC_EXP_NOLOAD
          c0[r] = 0.0;
          c1[r] = 0.0;
#else
          const double *rw = rowp(16 * I + lr + 4 * r) + 32 * pr + lc;
          c0[r] = rw[0];
          c1[r] = rw[16];
#endif
        }
      };
      auto unit = [&](int e, v4d &acc0, v4d &acc1) {
        const int it = e / npr, pr = e - it * npr, I = wave + kWaves * it;
        if (pr == 0) {
          // U' of this row tile: lane (lr, lc), register r: U'[16 I + lc][lr + 4 r]
          double w4[4], yI[4], yH[4];
          ld4(wks + lc * IYS + 4 * lr, w4);
          ld4(Y + (size_t)(16 * I + lc) * IYS + 4 * lr, yI);
          v4d ut = {0.0, 0.0, 0.0, 0.0};
          ut = mfma4(w4, yI, ut);
          if constexpr (kWide) {
            ld4(Yh + (size_t)(16 * I + lc) * IYS + 4 * lr, yH);
            ld4(wks + 16 * IYS + lc * IYS + 4 * lr, w4);
            ut = mfma4(w4, yH, ut);
            v4d u2 = {0.0, 0.0, 0.0, 0.0};
            ld4(wks + 2 * 16 * IYS + lc * IYS + 4 * lr, w4);
            u2 = mfma4(w4, yI, u2);
            ld4(wks + 3 * 16 * IYS + lc * IYS + 4 * lr, w4);
            u2 = mfma4(w4, yH, u2);
            ub[0] = u2[0]; ub[1] = u2[1]; ub[2] = u2[2]; ub[3] = u2[3];
          }
          ua[0] = ut[0]; ua[1] = ut[1]; ua[2] = ut[2]; ua[3] = ut[3];
          // this row's delta and - pose rows - its entries of the pose marginal, from the U' registers: a lane holds the columns
          // lr + 4 r of row q = 16 I + lc; the four lanes of a row (lc, lc + 16, lc + 32, lc + 48) are summed with the gfx950
          // permlane swaps (no trip through LDS):  delta' = delta - U' v  (U' = -Sigma A^T T^-1, v = -e - A delta),  D_i += U'_i Y_i^T
          const int q = 16 * I + lc;
          auto rowdot = [&](const double *vec, const double *vech) -> double {
            double y4[4];
            ld4(vec + 4 * lr, y4);
            double sdot = ua[0] * y4[0] + ua[1] * y4[1] + ua[2] * y4[2] + ua[3] * y4[3];
            if constexpr (kWide) {
              ld4(vech + 4 * lr, y4);
              sdot += ub[0] * y4[0] + ub[1] * y4[1] + ub[2] * y4[2] + ub[3] * y4[3];
            }
            sdot += rowgroup_xor<16>(sdot);
            sdot += rowgroup_xor<32>(sdot);
            return sdot;
          };
          const double sv = rowdot(vvec, vvec + 16);
          const bool prow = q < 3 * pn;
          const int qc = prow ? q : 0, pi = qc / 3, rp = qc - 3 * pi;
          const size_t o3 = (size_t)(3 * pi) * IYS;
          const double s0 = rowdot(Y + o3, Yh + o3), s1 = rowdot(Y + o3 + IYS, Yh + o3 + IYS), s2 = rowdot(Y + o3 + 2 * IYS, Yh + o3 + 2 * IYS);
          if (lr == 0 && q < n1) {
            dl[q] -= sv;
            if (prow) {
              double *D = Dl + 6 * pi + (rp * (rp + 1)) / 2;
              D[0] += s0;
              if (rp >= 1) D[1] += s1;
              if (rp >= 2) D[2] += s2;
            }
          }
        }
        {
          const int cA = 32 * pr + lc, cB = cA + 16;
          const int y0 = cA < 3 ? 3 * pn + cA : (cA < a0 ? 3 * P + cA - 3 : n1p);  // (pad columns: the zero row)
          const int y1 = cB < a0 ? 3 * P + cB - 3 : n1p;
          double yJ0[4], yJ1[4];
          ld4(Y + (size_t)y0 * IYS + 4 * lr, yJ0);
          ld4(Y + (size_t)y1 * IYS + 4 * lr, yJ1);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            if (ks < nks) {
              acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[ks], yJ0[ks], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[ks], yJ1[ks], acc1, 0, 0, 0);
            }
          if constexpr (kWide) {
            ld4(Yh + (size_t)y0 * IYS + 4 * lr, yJ0);
            ld4(Yh + (size_t)y1 * IYS + 4 * lr, yJ1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
              if (ks < nks1) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ub[ks], yJ0[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ub[ks], yJ1[ks], acc1, 0, 0, 0);
              }
          }
        }
#ifndef INC_EXP_NOSTORE
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double *rw = rowp(16 * I + lr + 4 * r) + 32 * pr + lc;
          rw[0] = acc0[r];
          rw[16] = acc1[r];
        }
#endif
      };
      // (straight-line body: every iteration issues the same loads - past the end a repeat of the last unit's tiles, never used -
      // so that the wait in front of a unit's products can count them.  With the loads under `if (e + 1 < nun)` the compiler
      // could not, waited for EVERYTHING in flight there, and the next unit's loads never ran under this unit's products.)
      int e = 0;
      loads(0, aA0, aA1);
      for (; e + 1 < nun; e += 2) {
        loads(e + 1, aB0, aB1);
        unit(e, aA0, aA1);
        loads(min(e + 2, nun - 1), aA0, aA1);
        unit(e + 1, aB0, aB1);
      }
      if (e < nun) unit(e, aA0, aA1);  // (an odd count: the last unit's tiles are the ones the loop requested last)
    }
    __syncthreads();
    if (b0 == 0) DRLGX_PROF(S, 38);
  };

  // ---- B'. the panel in HBM / L2: ONE walk over it per batch of up to 8 NT re-observed landmarks (k = 2 nb <= 16 NT columns) ----
  // In the form above every batch of 8 landmarks reads the panel twice (Y = Sigma A^T, then the tiles) and writes it once; at BASELINE
  // config 5 (110 poses, 92 landmarks: a 770 KB panel per instance, ~21 factors per step) that is three batches = 1.8 GB per
  // 256-instance launch, and the update ran at the memory system's speed: 370-420 us.  Here
  //   S1  Ya = (Sigma A^T) for the ACTIVE rows only (the current pose, the landmarks: 36 % of the rows), kept in LDS indexed by the
  //       panel's column - it is the B operand of every tile product and all that T = R + A Ya needs;
  //   S2  one wave: T (k x k, NT x NT tiles in registers, both triangles), W' = -T^-1 by block Gauss-Jordan on the matrix cores
  //       (pivot tiles by inv16_blk; with both triangles in registers every product is of the form X^T Z that accumulator
  //       registers feed directly), stored as operand images;
  //   S3  per row tile, once: the lanes form THEIR entries of Y^T (the B operand of U'^T = W' Y^T) straight from the rows in HBM /
  //       L2 - lines the tile loads behind them want anyway -, U' on the matrix cores, delta and the pose marginals from the U'
  //       registers and the neighbouring lanes' Y (DPP), then every column pair C += U' Ya^T as above.
  // Pose rows are dealt 15 to a tile (five poses: the three rows of a pose never straddle tiles, so the marginal's row products
  // stay inside the 16-lane rows), landmark rows 16.  One batch of all re-observed landmarks equals the sequence of batches of 8
  // in exact arithmetic (independent measurement noise).
  auto sbatch = [&](int b0, int nb, auto nt_tag) {
    constexpr int NT = decltype(nt_tag)::value;
    const int k = 2 * nb, npr = (a0 + 31) >> 5, yrows = x.yrows;
    double *Yt = x.yt, *jt = x.jt, *vv = x.svv;
    const size_t ytile = (size_t)yrows * IYS;
    // tile i of the NT x NT images of W' (and, before them, of the register dumps of T): the first five where the LDS form keeps its
    // W' images, the others behind Ya
    auto timg = [&](int i) -> double * { return i < 5 ? wks + (size_t)i * 16 * IYS : x.wim + (size_t)(i - 5) * 16 * IYS; };
    // S0 / S1. the batch's Jacobians in column order (kk = 2 f + u: Jx (3), Jl (2), the landmark's first column); Ya
    if (tid >= kThreads - 16 * NT) {
      const int kk = kThreads - 1 - tid, f = kk >> 1, u = kk & 1;
      double *o = jt + 8 * kk;
      int cl = 3;
      double j0 = 0, j1 = 0, j2 = 0, j3 = 0, j4 = 0;
      if (f < nb) {
        const int t = fre[b0 + f];
        const double *rc = recs + (size_t)REC * t;
        j0 = rc[3 * u]; j1 = rc[3 * u + 1]; j2 = rc[3 * u + 2]; j3 = rc[6 + 2 * u]; j4 = rc[7 + 2 * u];
        cl = 3 + 2 * fslot[t];
      }
      o[0] = j0; o[1] = j1; o[2] = j2; o[3] = j3; o[4] = j4;
      reinterpret_cast<int *>(o + 5)[0] = cl;
    }
    {
      constexpr int FS = 8 * NT, RS = kThreads / FS;
      const int f = tid % FS, rr = tid / FS;
      if (rr < RS) {
        double j0 = 0, j1 = 0, j2 = 0, j3 = 0, j4 = 0, j5 = 0, l0c = 0, l1c = 0, l2c = 0, l3c = 0;
        int cl = 3;
        if (f < nb) {
          const int t = fre[b0 + f];
          const double *rc = recs + (size_t)REC * t;
          j0 = rc[0]; j1 = rc[1]; j2 = rc[2]; j3 = rc[3]; j4 = rc[4]; j5 = rc[5];
          l0c = rc[6]; l1c = rc[7]; l2c = rc[8]; l3c = rc[9];
          cl = 3 + 2 * fslot[t];
        }
        double *Yf = Yt + (size_t)(f >> 3) * ytile;
        const int k0 = ks16(2 * (f & 7)), k1 = ks16(2 * (f & 7) + 1);
        for (int c0 = rr; c0 < a0; c0 += 4 * RS) {
          double p0[4], p1[4], p2[4], l0[4], l1[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = min(c0 + u * RS, a0 - 1);
            const double *r = rowp(c < 3 ? 3 * pn + c : 3 * P + c - 3);
            p0[u] = r[0]; p1[u] = r[1]; p2[u] = r[2]; l0[u] = r[cl]; l1[u] = r[cl + 1];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * RS;
            if (c < a0) {
              Yf[(size_t)c * IYS + k0] = p0[u] * j0 + p1[u] * j1 + p2[u] * j2 + l0[u] * l0c + l1[u] * l1c;
              Yf[(size_t)c * IYS + k1] = p0[u] * j3 + p1[u] * j4 + p2[u] * j5 + l0[u] * l2c + l1[u] * l3c;
            }
          }
        }
      }
      for (int e = tid; e < NT * IYS; e += kThreads) {  // the pad columns' zero row
        const int t = e / IYS, o = e - t * IYS;
        Yt[(size_t)t * ytile + (size_t)a0 * IYS + o] = 0.0;
      }
    }
    __syncthreads();
    if (b0 == 0) DRLGX_PROF(S, 36);
    // S2. T = R + A Ya and v = -e - A delta by everybody (T as the register dumps of its NT x NT tiles, where the images of W' go
    // afterwards); W' = -T^-1 by one wave.  (Assembled in registers by that wave alone the compiler requested every operand of every
    // entry at once: 160 loads in flight at NT = 2, and every belief kernel of this translation unit spilled.)
#ifndef INC_EXP_NO_S2
    {
      if (tid < 16 * NT) {
        double v = 0.0;
        if (tid < k) {
          const int t = fre[b0 + (tid >> 1)], u = tid & 1;
          const double *rc = recs + (size_t)REC * t;
          const double *dp = dl + 3 * pn, *dq = dl + 3 * P + 2 * fslot[t];
          v = -rc[10 + u] - (rc[3 * u] * dp[0] + rc[3 * u + 1] * dp[1] + rc[3 * u + 2] * dp[2] + rc[6 + 2 * u] * dq[0] + rc[7 + 2 * u] * dq[1]);
        }
        vv[(tid & ~15) + ks16(tid & 15)] = v;
      }
#pragma clang loop unroll(disable)
      for (int e = tid; e < NT * NT * 256; e += kThreads) {
        const int tile = e >> 8, rg = (e >> 6) & 3, ln = e & 63, ta = tile / NT, tb = tile - NT * ta;
        const int i = 16 * ta + (ln >> 4) + 4 * rg, j = 16 * tb + (ln & 15), hi = max(i, j), lo = min(i, j);
        double sv = i == j ? 1.0 : 0.0;
        if (hi < k) {  // row hi of A (factor hi / 2, bearing or range) against column lo of Ya
          const double *ji = jt + 8 * hi;
          const int cl = reinterpret_cast<const int *>(ji + 5)[0];
          const double *yc = Yt + (size_t)(lo >> 4) * ytile + ks16(lo & 15);
          sv = ji[0] * yc[0] + ji[1] * yc[IYS] + ji[2] * yc[2 * IYS] + ji[3] * yc[(size_t)cl * IYS] + ji[4] * yc[(size_t)(cl + 1) * IYS];
          if (hi == lo) sv += (hi & 1) ? Rr : Rb;
        }
        timg(tile)[e & 255] = sv;
      }
    }
    __syncthreads();
    if (wave == 0) {
      v4d T[NT][NT];
#pragma unroll
      for (int ta = 0; ta < NT; ++ta)
#pragma unroll
        for (int tb = 0; tb < NT; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) T[ta][tb][r] = timg(ta * NT + tb)[64 * r + lane];
      wave_lds_sync();  // (the images below overwrite the dumps)
      // block Gauss-Jordan, E = -D^-1 convention (as the sweep of k_slam.hip): after the last pivot tile T = -T^-1.
      // mfma4(X registers, Z registers) = X^T Z; T is symmetric, so tile (i, K) is the transpose of tile (K, i).
#pragma unroll
      for (int K = 0; K < NT; ++K) {
        if (16 * K < k) {
          const int nact = min(16, k - 16 * K);
          const SweepCtx sx{0, lane, lc, lr, nact, 16, true, true, ictl + 1, nullptr};
          inv16_blk<true>(sx, nact, T[K][K]);
          const double e4[4] = {T[K][K][0], T[K][K][1], T[K][K][2], T[K][K][3]};
          v4d V[NT], Wt[NT];
#pragma unroll
          for (int i = 0; i < NT; ++i)
            if (i != K) {
              const double tki[4] = {T[K][i][0], T[K][i][1], T[K][i][2], T[K][i][3]};
              const v4d z = {0.0, 0.0, 0.0, 0.0};
              V[i] = mfma4(e4, tki, z);   // E T_Ki
              Wt[i] = mfma4(tki, e4, z);  // T_iK E
            }
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if (j != K) {
              const double tkj[4] = {T[K][j][0], T[K][j][1], T[K][j][2], T[K][j][3]};
#pragma unroll
              for (int i = 0; i < NT; ++i)
                if (i != K) {
                  const double vi[4] = {V[i][0], V[i][1], V[i][2], V[i][3]};
                  T[j][i] = mfma4(tkj, vi, T[j][i]);  // T_ji += T_jK E T_Ki
                }
            }
#pragma unroll
          for (int i = 0; i < NT; ++i)
            if (i != K) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                T[K][i][r] = -V[i][r];
                T[i][K][r] = -Wt[i][r];
              }
            }
        }
      }
      // operand images: image (mt, t) row lc = W'[16 mt + lc][16 t + lr + 4 r] at ks16 position 4 lr + r = the registers of tile
      // (t, mt) as they are (W' is symmetric); the inactive part zeroed
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          double *o = timg(mt * NT + t) + lc * IYS + 4 * lr;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (16 * mt + lc < k && 16 * t + lr + 4 * r < k) ? T[t][mt][r] : 0.0;
        }
    }
#endif
    __syncthreads();
    if (b0 == 0) DRLGX_PROF(S, 37);
    else DRLGX_PROF(S, 39);  // (a second batch: start of its walk)
    // S3. the walk over the panel
    const int TP = (P + 4) / 5, ntr = TP + ((n1 - 3 * P + 15) >> 4);
    auto rowq = [&](int I, int j) -> int {  // panel row of tile I's row j, or -1
      if (I < TP) {
        const int q = 15 * I + j;
        return (j < 15 && q < 3 * P) ? q : -1;
      }
      const int q = 3 * P + 16 * (I - TP) + j;
      return q < n1 ? q : -1;
    };
#ifndef INC_EXP_NO_S3
    if (wave < ntr) {
      const int nun = ((ntr - wave + kWaves - 1) / kWaves) * npr;
      const int nksl = (k - 16 * (NT - 1) + 3) >> 2;  // K steps with live columns in the last factor tile
      v4d aA0, aA1, aB0, aB1;
      double ua[NT][4];
      // the operands of a row tile's Y^T entries, requested a whole row tile ahead: the first three columns of this lane's row and
      // its entries in the columns of the batch's landmarks (lines the tile loads behind them want anyway)
      double gp[3], gl0[NT][4], gl1[NT][4];
      auto gather = [&](int I) {
        const double *rw = rowp(max(rowq(I, lc), 0));
        gp[0] = rw[0]; gp[1] = rw[1]; gp[2] = rw[2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int ss = 0; ss < 4; ++ss) {
            const int cl = reinterpret_cast<const int *>(jt + 8 * (16 * t + lr + 4 * ss) + 5)[0];
            gl0[t][ss] = rw[cl];
            gl1[t][ss] = rw[cl + 1];
          }
      };
      auto loads = [&](int e, v4d &c0, v4d &c1) {
        const int it = e / npr, pr = e - it * npr, I = wave + kWaves * it;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double *rw = rowp(max(rowq(I, lr + 4 * r), 0)) + 32 * pr + lc;
          c0[r] = rw[0];
          c1[r] = rw[16];
        }
      };
      auto unit = [&](int e, v4d &acc0, v4d &acc1) {
        const int it = e / npr, pr = e - it * npr, I = wave + kWaves * it;
        if (pr == 0) {
          // this lane's entries of Y^T: row q = tile row lc, factor columns 16 t + lr + 4 s.  (Requesting these operands a row tile
          // ahead was built: 35 more doubles live across the tile products, 1.1 KB of scratch per thread in the kernels of the
          // pose-chain solver - dropped.)
          const int q = rowq(I, lc);
          const double *rw = rowp(max(q, 0));
          const double p0 = rw[0], p1 = rw[1], p2 = rw[2];
          double yb[NT][4], l0[NT][4], l1[NT][4];
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
              const int cl = reinterpret_cast<const int *>(jt + 8 * (16 * t + lr + 4 * ss) + 5)[0];
              l0[t][ss] = rw[cl];
              l1[t][ss] = rw[cl + 1];
            }
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
              const double *jj = jt + 8 * (16 * t + lr + 4 * ss);
              yb[t][ss] = p0 * jj[0] + p1 * jj[1] + p2 * jj[2] + l0[t][ss] * jj[3] + l1[t][ss] * jj[4];
            }
          // U'^T = W' Y^T: tile mt, register r at lane (lr, lc) = U'[row lc][16 mt + lr + 4 r] - the A operand of the tile products
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) {
            v4d ut = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              double w4[4];
              ld4(timg(mt * NT + t) + lc * IYS + 4 * lr, w4);
              ut = mfma4(w4, yb[t], ut);
            }
            ua[mt][0] = ut[0]; ua[mt][1] = ut[1]; ua[mt][2] = ut[2]; ua[mt][3] = ut[3];
          }
          // delta' = delta - U' v;  pose rows: D_i += U'_i Y_i^T - the Y of the pose's earlier rows sits one / two lanes below
          double sv = 0.0, so = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) {
            double v4[4];
            ld4(vv + 16 * mt + 4 * lr, v4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sv += ua[mt][r] * v4[r];
              so += ua[mt][r] * yb[mt][r];
              s1 += ua[mt][r] * row_shr_f64<1>(yb[mt][r]);
              s2 += ua[mt][r] * row_shr_f64<2>(yb[mt][r]);
            }
          }
          sv += rowgroup_xor<16>(sv); sv += rowgroup_xor<32>(sv);
          so += rowgroup_xor<16>(so); so += rowgroup_xor<32>(so);
          s1 += rowgroup_xor<16>(s1); s1 += rowgroup_xor<32>(s1);
          s2 += rowgroup_xor<16>(s2); s2 += rowgroup_xor<32>(s2);
          if (lr == 0 && q >= 0) {
            dl[q] -= sv;
            if (q < 3 * pn) {
              const int pi = q / 3, rp = q - 3 * pi;
              double *D = Dl + 6 * pi + (rp * (rp + 1)) / 2;
              D[rp] += so;
              if (rp >= 1) D[rp - 1] += s1;
              if (rp >= 2) D[0] += s2;
            }
          }
        }
        {
          const int cA = min(32 * pr + lc, a0), cB = min(32 * pr + 16 + lc, a0);  // (pad columns: the zero row)
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) {
            double yJ0[4], yJ1[4];
            ld4(Yt + (size_t)mt * ytile + (size_t)cA * IYS + 4 * lr, yJ0);
            ld4(Yt + (size_t)mt * ytile + (size_t)cB * IYS + 4 * lr, yJ1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
              if (mt < NT - 1 || ks < nksl) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[mt][ks], yJ0[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[mt][ks], yJ1[ks], acc1, 0, 0, 0);
              }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = rowq(I, lr + 4 * r);
          if (q >= 0) {
            double *rw = rowp(q) + 32 * pr + lc;
            rw[0] = acc0[r];
            rw[16] = acc1[r];
          }
        }
      };
      int e = 0;
      loads(0, aA0, aA1);
      for (; e + 1 < nun; e += 2) {
        loads(e + 1, aB0, aB1);
        unit(e, aA0, aA1);
        loads(min(e + 2, nun - 1), aA0, aA1);
        unit(e + 1, aB0, aB1);
      }
      if (e < nun) unit(e, aA0, aA1);
    }
#endif
    __syncthreads();
    if (b0 == 0) DRLGX_PROF(S, 38);
    else DRLGX_PROF(S, 47);  // (... and its end)
  };
  // Which form serves this update (the same decision in the fused step and in the stage kernels: plan quantities and list lengths
  // only): the streamed one where it exists (kSNT > 0: the kernels of the pose-chain solver; x.snt > 0: the panel in HBM / L2) unless
  // the two-walk form needs a single batch and the panel is small enough for L2 to serve its second walk (measured in the
  // 100-landmark world, two-walk / streamed: 41 / 53 us at 96 poses and 43 / 53 at 112, where one wide batch does; 74 / 64 at 128 and
  // 105 / 91 at 204, where the second half of Y no longer fits the LDS; profiles/r06_ab_streamed_update.txt).  One form per update:
  // Ya lies where the other form keeps Y.
  bool streamed = false;
  if constexpr (!kLds && kSNT > 0)
#ifndef INC_EXP_NOSTREAM
    streamed = x.snt > 0 && (n_re > (x.wide ? 16 : 8) || (size_t)n1 * a0 * 8 > (size_t)(INC_STREAM_KB << 10));
#endif
  // (a loop per form: in one loop the invariants of both forms were hoisted in front of it and lived side by side - 170 B of scratch)
  if (streamed) {
    if constexpr (!kLds && kSNT > 0) {
      for (int b0 = 0; b0 < n_re;) {
        const int left = n_re - b0, nt = min(x.snt, (left + 7) >> 3), nb = min(8 * nt, left);
        if (nt >= 4) sbatch(b0, nb, std::integral_constant<int, kSNT >= 4 ? 4 : 1>{});
        else if (nt == 3) sbatch(b0, nb, std::integral_constant<int, 3>{});
        else if (nt == 2) sbatch(b0, nb, std::integral_constant<int, 2>{});
        else sbatch(b0, nb, std::integral_constant<int, 1>{});
        b0 += nb;
      }
    }
  } else {
    for (int b0 = 0; b0 < n_re;) {
      const int left = n_re - b0;
      if (left > 8 && x.wide) {
        const int nb = min(16, left);
        batch(b0, nb, std::true_type{});
        b0 += nb;
      } else {
        const int nb = min(8, left);
        batch(b0, nb, std::false_type{});
        b0 += nb;
      }
    }
  }
  DRLGX_PROF(S, 3);
  // ---- C. landmarks seen for the first time ----
  if (nn > 0) {
    if (tid < nn) {
      const double *rc = recs + (size_t)REC * fnew[tid];
      const double det = rc[6] * rc[9] - rc[7] * rc[8], id = 1.0 / det;
      const double i00 = rc[9] * id, i01 = -rc[7] * id, i10 = -rc[8] * id, i11 = rc[6] * id;  // Jl^-1
      double *g = gs + 12 * tid;
      for (int c = 0; c < 3; ++c) {
        g[c] = -(i00 * rc[c] + i01 * rc[3 + c]);
        g[3 + c] = -(i10 * rc[c] + i11 * rc[3 + c]);
      }
      g[6] = -(i00 * rc[10] + i01 * rc[11]);
      g[7] = -(i10 * rc[10] + i11 * rc[11]);
      g[8] = i00 * Rb * i00 + i01 * Rr * i01;  // Jl^-1 R Jl^-T
      g[9] = i00 * Rb * i10 + i01 * Rr * i11;
      g[10] = i10 * Rb * i10 + i11 * Rr * i11;
    }
    __syncthreads();
    // C1. the new rows against the old columns; delta_l
    for (int e = tid; e < 2 * nn * a0; e += kThreads) {
      const int rr = e / a0, c = e - rr * a0;
      const double *g = gs + 12 * (rr >> 1) + 3 * (rr & 1);
      rowp(n1 + rr)[c] = g[0] * rowp(3 * pn)[c] + g[1] * rowp(3 * pn + 1)[c] + g[2] * rowp(3 * pn + 2)[c];
    }
    if (tid < 2 * nn) {
      const double *g = gs + 12 * (tid >> 1);
      dl[n1 + tid] = g[3 * (tid & 1)] * dl[3 * pn] + g[3 * (tid & 1) + 1] * dl[3 * pn + 1] + g[3 * (tid & 1) + 2] * dl[3 * pn + 2] + g[6 + (tid & 1)];
    }
    __syncthreads();
    // C2. the new columns of every row (Sigma[., l] = Sigma[., x'] G^T), + the landmark's own noise on its diagonal block
    for (int e = tid; e < n * 2 * nn; e += kThreads) {
      const int q = e / (2 * nn), cc = e - q * 2 * nn;
      const double *g = gs + 12 * (cc >> 1);
      const double *r = rowp(q);
      double v = r[0] * g[3 * (cc & 1)] + r[1] * g[3 * (cc & 1) + 1] + r[2] * g[3 * (cc & 1) + 2];
      if (q >= n1 && ((q - n1) >> 1) == (cc >> 1)) {
        const int er = (q - n1) & 1, ec = cc & 1;
        v += er == ec ? (er ? g[10] : g[8]) : g[9];
      }
      rowp(q)[a0 + cc] = v;
    }
    __syncthreads();
  }
  DRLGX_PROF(S, 4);
  // ---- D. outputs: estimates theta (+) delta, information = inverse(marginal covariance) (SLAM2D.cpp:395-417), the panel ----
  double *est_pose = S.est_pose + (size_t)inst * S.P_max * 4;
  double *pose_info = S.pose_info + (size_t)inst * S.P_max * 6;
  double *pose_tr = S.pose_tr + (size_t)inst * S.P_max;
  for (int i = tid; i < P; i += kThreads) {
    const Pose t{thp[4 * i], thp[4 * i + 1], thp[4 * i + 2], thp[4 * i + 3]};
    const double d0 = dl[3 * i], d1 = dl[3 * i + 1], d2 = dl[3 * i + 2];
    const Pose e = compose(t, make_pose(d0, d1, d2));
    est_pose[4 * i] = e.x; est_pose[4 * i + 1] = e.y; est_pose[4 * i + 2] = e.c; est_pose[4 * i + 3] = e.s;
    d_pose[3 * i] = d0; d_pose[3 * i + 1] = d1; d_pose[3 * i + 2] = d2;
    double c00, c10, c11, c20, c21, c22;
    if (i == pn) {  // the current pose's marginal is its block of the panel
      const double *r0 = rowp(3 * pn), *r1 = rowp(3 * pn + 1), *r2 = rowp(3 * pn + 2);
      c00 = r0[0]; c10 = 0.5 * (r1[0] + r0[1]); c11 = r1[1]; c20 = 0.5 * (r2[0] + r0[2]); c21 = 0.5 * (r2[1] + r1[2]); c22 = r2[2];
    } else {
      const double *D = Dl + 6 * i;
      c00 = D[0]; c10 = D[1]; c11 = D[2]; c20 = D[3]; c21 = D[4]; c22 = D[5];
    }
    double *go = jd + 6 * i;
    go[0] = c00; go[1] = c10; go[2] = c11; go[3] = c20; go[4] = c21; go[5] = c22;
    pose_tr[i] = c00 + c11 + c22;
    double info[6];
    inv3_sym_fast(c00, c10, c20, c11, c21, c22, info);
    for (int k = 0; k < 6; ++k) pose_info[6 * i + k] = info[k];
    if (hand) {
      hand[4 * i] = e.x; hand[4 * i + 1] = e.y; hand[4 * i + 2] = e.c; hand[4 * i + 3] = e.s;
      for (int k = 0; k < 6; ++k) hand[4 * hand_cap + 6 * i + k] = info[k];
    }
  }
  {
    double *est_lm = S.est_lm + (size_t)inst * S.L_max * 2;
    double *lm_info = S.lm_info + (size_t)inst * S.L_max * 3;
    double *lm_tr = S.lm_tr + (size_t)inst * S.L_max;
    for (int j = kThreads - 1 - tid; j < L; j += kThreads) {
      const double dx = dl[3 * P + 2 * j], dy = dl[3 * P + 2 * j + 1];
      d_lm[2 * j] = dx; d_lm[2 * j + 1] = dy;
      const double ex = thl[2 * j] + dx, ey = thl[2 * j + 1] + dy;
      est_lm[2 * j] = ex;
      est_lm[2 * j + 1] = ey;
      if (hand) {  // (the linearisation points are dead by now: the map stage reads the estimates here)
        thl[2 * j] = ex;
        thl[2 * j + 1] = ey;
      }
      const double *r0 = rowp(3 * P + 2 * j), *r1 = rowp(3 * P + 2 * j + 1);
      const double c00 = r0[3 + 2 * j], c11 = r1[3 + 2 * j + 1], cs = 0.5 * (r0[3 + 2 * j + 1] + r1[3 + 2 * j]);
      lm_tr[j] = c00 + c11;
      const double id = 1.0 / (c00 * c11 - cs * cs);  // marginalCovariance(l).inverse() (SLAM2D.cpp:417)
      lm_info[3 * j] = c11 * id;
      lm_info[3 * j + 1] = -cs * id;
      lm_info[3 * j + 2] = c00 * id;
    }
  }
  DRLGX_PROF(S, 5);
  if constexpr (kLds) {
    const int npair = (a + 1) >> 1;
    const int cp0 = tid & 31, r0 = tid >> 5;
    for (int cp = cp0; cp < npair; cp += 32)
      for (int q = r0; q < n; q += 16) reinterpret_cast<double2 *>(growp(q))[cp] = reinterpret_cast<const double2 *>(rowp(q))[cp];
  }
  if (tid == 0) {
    meta[0] = 1; meta[1] = P; meta[2] = L; meta[3] = M;
    cnt[C_ISAM] = cnt[C_ISAM] + 1;
    cnt[C_NEWP] = P;
    cnt[C_NEWL] = L;
    if (ictl[1]) atomicMin(S.status, DRLGX_E_NUMERIC);
    if (S.inc_stats) atomicAdd(S.inc_stats, 1ull);
  }
  DRLGX_PROF(S, 7);
  return true;
}

// The panel after a full (dense) solve: Sigma[:, active] from what SlamCtx::back leaves in LDS - A = -Sigma_pp (packed lower
// triangle), the per-factor G_m = Lambda_pl Lambda_ll^-1 blocks, the per-landmark factor lists:
//     Sigma_pl = -Sigma_pp G          (column block of landmark j: the sum over its factor list)
//     Sigma_ll = Lambda_ll^-1 + G^T Sigma_pp G = Lambda_ll^-1 - G^T Sigma_pl
// written to the HBM panel (the second stage reads the first one's rows back through L2).
struct __attribute__((aligned(8))) PanelPair {  // two adjacent panel entries (the landmark columns start at column 3: 8-byte aligned only)
  double x, y;
};
__device__ __forceinline__ void panel_from_dense(const DrlgxState &S, const SlamCtx &c, int tid) {
  if (!S.jc) return;
  const int inst = c.inst, P = c.P, L = c.L, pn = P - 1;
  double *gpan = S.jc + (size_t)inst * S.jc_stride;
  const int ldg = S.jc_ld;
  auto prow = [&](int q) -> double * { return gpan + (size_t)q * ldg; };                           // pose rows
  auto lrow = [&](int q) -> double * { return gpan + (size_t)(3 * S.P_max + q) * ldg; };           // landmark rows
  auto asym = [&](int i, int j) -> double { return c.A[c.AT(max(i, j), min(i, j))]; };
  double *jd = S.jd + (size_t)inst * S.P_max * 6;
  DRLGX_PROF(S, 44);
  if (tid == 0) c.bad[1] = 0;  // work counter of the two item loops below
  // pose rows: columns of the current pose, the marginal, the landmark blocks
  for (int e = tid; e < 3 * P * 3; e += kThreads) {
    const int q = e / 3, cc = e - 3 * q;
    prow(q)[cc] = -asym(q, 3 * pn + cc);
  }
  for (int e = tid; e < 6 * P; e += kThreads) {
    const int i = e / 6, t = e - 6 * i;
    const int r = t < 1 ? 0 : (t < 3 ? 1 : 2), cc = t - (r * (r + 1)) / 2;
    jd[e] = -c.A[c.AT(3 * i + r, 3 * i + cc)];
  }
  __syncthreads();
  // Sigma_pl: one work item = (landmark j, pose i); the lanes of a wave share j and take consecutive poses, so that the walk over
  // j's factor list is uniform (one trip count, the G block and the observing pose are wave-uniform loads: no divergence - with
  // consecutive LANDMARKS per lane every wave paid for the longest list, ~40 entries against ~13 on average, and this stage plus
  // the next cost more than the whole dense solve: 74 us at 40 poses).  Same sums in the same order per output as before.
  // The items are handed out through an LDS counter (c.bad[1], free after the sweep): list lengths are very uneven - landmarks
  // near the start are seen from most poses - and a static deal left some waves with twice the work of others.
  // (Requesting the next factor's list entry, pose and G block under the current one's products - a hand-made two-stage pipeline of
  // the four dependent LDS round trips per factor - measured SLOWER: 17.9 -> 21.9 us at 40 poses, 20.6 -> 24.5 at 50.)
  {
    const int lane = tid & 63, ib = (P + 63) >> 6;  // pose blocks of 64 per landmark
    while (true) {
      int w = 0;
      if (lane == 0) w = atomicAdd(c.bad + 1, 1);
      w = __builtin_amdgcn_readfirstlane(w);
      if (w >= L * ib) break;
      const int j = w / ib, i = (w - j * ib) * 64 + lane;
      if (i >= P) continue;
      int rb[3];  // packed row starts of this pose's three rows
#pragma unroll
      for (int r = 0; r < 3; ++r) rb[r] = ((3 * i + r) * (3 * i + r + 1)) >> 1;
      double b[6] = {0, 0, 0, 0, 0, 0};
      const int t1 = c.lstart[j + 1];
      for (int t = c.lstart[j]; t < t1; ++t) {
        const int m = c.lfac[t], p = c.mp[m];
        const double *g = c.rec + (size_t)REC * m;
        const double g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4], g5 = g[5];
        const int c0 = 3 * p, cb0 = (c0 * (c0 + 1)) >> 1, cb1 = cb0 + c0 + 1, cb2 = cb1 + c0 + 2;  // (wave-uniform)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int q = 3 * i + r;
          // asym(q, c) = A[max (max + 1) / 2 + min]
          const double a0 = c.A[q >= c0 ? rb[r] + c0 : cb0 + q], a1 = c.A[q >= c0 + 1 ? rb[r] + c0 + 1 : cb1 + q],
                       a2 = c.A[q >= c0 + 2 ? rb[r] + c0 + 2 : cb2 + q];
          b[2 * r] += a0 * g0 + a1 * g2 + a2 * g4;
          b[2 * r + 1] += a0 * g1 + a1 * g3 + a2 * g5;
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) *reinterpret_cast<PanelPair *>(prow(3 * i + r) + 3 + 2 * j) = PanelPair{b[2 * r], b[2 * r + 1]};
    }
  }
  __syncthreads();
  DRLGX_PROF(S, 45);
  if (tid == 0) c.bad[1] = 0;
  // landmark rows
  for (int e = tid; e < 2 * L * 3; e += kThreads) {
    const int q = e / 3, cc = e - 3 * q;
    lrow(q)[cc] = prow(3 * pn + cc)[3 + q];
  }
  __syncthreads();
  // Sigma_ll: one work item = (landmark j, landmark j2), the lanes of a wave share j (uniform list walk, uniform G) and take
  // consecutive j2: the rows of Sigma_pl come back from L2 as contiguous 16-byte pieces
  {
    const int lane = tid & 63, jb = (L + 63) >> 6;
    while (true) {
      int w = 0;
      if (lane == 0) w = atomicAdd(c.bad + 1, 1);
      w = __builtin_amdgcn_readfirstlane(w);
      if (w >= L * jb) break;
      const int j = w / jb, j2 = (w - j * jb) * 64 + lane;
      if (j2 >= L) continue;
      double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
      if (j == j2) {
        const double *lb = c.lamb + 8 * j;
        s00 = lb[3]; s01 = lb[4]; s10 = lb[4]; s11 = lb[5];
      }
      // (the rows come back through L2, ~1 us per dependent round trip: four list entries' loads are in flight together - the
      // longest list, ~40 entries, used to set this stage's time at one round trip per entry)
      const int t1 = c.lstart[j + 1];
      for (int t = c.lstart[j]; t < t1; t += 4) {
        const double *g[4];
        PanelPair x[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = c.lfac[min(t + u, t1 - 1)], p = c.mp[m];
          g[u] = c.rec + (size_t)REC * m;
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) x[u][kk] = *reinterpret_cast<const PanelPair *>(prow(3 * p + kk) + 3 + 2 * j2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (t + u >= t1) break;
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const double x0 = x[u][kk].x, x1 = x[u][kk].y;
            s00 -= g[u][2 * kk] * x0; s01 -= g[u][2 * kk] * x1;
            s10 -= g[u][2 * kk + 1] * x0; s11 -= g[u][2 * kk + 1] * x1;
          }
        }
      }
      *reinterpret_cast<PanelPair *>(lrow(2 * j) + 3 + 2 * j2) = PanelPair{s00, s01};
      *reinterpret_cast<PanelPair *>(lrow(2 * j + 1) + 3 + 2 * j2) = PanelPair{s10, s11};
    }
  }
  DRLGX_PROF(S, 46);
  if (tid == 0) {
    int *meta = inc_meta(S, inst);
    meta[0] = 1; meta[1] = P; meta[2] = L; meta[3] = c.M;
    if (S.inc_stats) atomicAdd(S.inc_stats + 1, 1ull);
  }
}

// an update by a solver call that leaves no panel (estimates-only solves; the pose-chain solver builds one when S.jc is set)
__device__ __forceinline__ void panel_invalidate(const DrlgxState &S, int inst, int tid) {
  if (S.jc && tid == 0) {
    inc_meta(S, inst)[0] = 0;
    if (S.inc_stats) atomicAdd(S.inc_stats + 1, 1ull);
  }
}

#pragma clang fp contract(fast)
