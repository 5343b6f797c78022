// Batched SLAM belief update: one 512-thread workgroup per instance, the whole problem on chip.
//
// Restates SLAM2D::optimize / copy_optimize (src/em_exploration/SLAM2D.cpp:374-488) — one iSAM2
// update (gtsam ISAM2::update, third-party; policy in SURVEY.md App. A.3) followed by the block
// marginals of FastMarginals (src/em_exploration/FastMarginals.cpp:130-186):
//   front end (SlamCtx::front; in k_step it runs beside the simulator wave, on the state before the step):
//   1. relinearisation policy (every 10th update, |delta|_inf >= 0.1), theta staged in LDS
//   2. every bearing-range factor is linearised ONCE by its own thread into a 12-double LDS record (linearize_br); landmark
//      2x2 blocks (thread per landmark, walking a bit mask of the observing poses), pose 3x3 blocks by roles: odometry
//      factors linearised once for both keys, own factors summed eight lanes per pose
//   back end (SlamCtx::back; after the simulator):
//   3. this step's factors; landmarks are eliminated analytically -> Schur complement S on the poses (3P x 3P, LDS)
//   4. symmetric Gauss-Jordan SWEEP of the augmented system [S rhs] on the fp64 matrix cores
//      (v_mfma_f64_16x16x4_f64): the lower triangle lives in 16 x 16 accumulator tiles in registers for the whole
//      factorisation; afterwards the triangle holds -S^-1 (every pose marginal and cross block) and the augmented row
//      holds delta_p.  Fast path (<= 42 poses, everything in LDS): sweep_packed_fast - 16-wide block pivots, one role per
//      wave (sweep_role), the next diagonal tile inverted in registers by an otherwise idle wave (inv16_blk).
//      sweep_regtiles / sweep_streamed are the same block steps for the landmark systems of k_slam_arrow.hip that do not
//      fit that scheme (tiles in registers with panels in LDS; everything in the HBM/L2 workspace).
//   5. landmark deltas and 2x2 landmark marginals by back-substitution through G = Lambda_pl Lambda_ll^-1
//   6. estimates theta (+) delta, information blocks (3x3 cofactor inverse / 2x2 inverse), traces
// LDS: the packed system + per-factor records when they fit; the records fall back to an HBM/L2 workspace otherwise.
// Trajectories beyond 42 poses: k_slam_arrow.hip (pose chain eliminated first).
#include "drlgx_dev.h"
#include <type_traits>
#pragma clang fp contract(fast)  // (the unity build k_step.hip is compiled with -ffp-contract=off)

namespace kslam {

constexpr int kThreads = 512;
constexpr int REC = 12;  // per-factor record: [0..5] Jx (2x3) -> later G (3x2); [6..9] Jl (2x2) -> later partial; [10..11] e
// (12 doubles put the 64-bit accesses of a half-wave that walks consecutive records on 8 banks; a stride of 13 is
// conflict-free and measured SLOWER - G 1.3 -> 2.3 us, Schur 7.8 -> 9.0, landmark marginals 4.4 -> 6.4: the records lose
// their 16-byte alignment and with it the 128-bit loads)

// (linearize_br, fast_rcp and inv16_blk are shared with the incremental update (k_inc.hip), whose fused and staged forms must
// round alike although they are inlined into differently shaped code: contraction decided in the front end for them)
#pragma clang fp contract(on)
// BearingRangeFactor linearised at (pose, landmark) (SLAM2D.cpp:91-124; gtsam BearingRangeFactor).  d = the landmark in the
// pose frame, n = |d|, (c, s) = d / n: the predicted bearing is atan2(s, c) and never needed as an angle - the error
// Rot2 Local(measured, predicted) is taken from (c, s) directly - and the predicted range is n; the range Jacobians are
// (-c, -s, 0) for the pose and R (c, s) for the landmark.  One square root, one division, one sincos, one atan2 (the
// composition of bearing_of / range_of - two atan2, four trigonometric calls, two roots, six divisions - took most of the
// 2.9 us the factor tables cost).
__device__ __forceinline__ void linearize_br(const double *tp, const double *tl, double bm, double rm, double *rec) {
  Pose ps{tp[0], tp[1], tp[2], tp[3]};
  P2 lm{tl[0], tl[1]};
  const P2 d = transform_to(ps, lm);
  const double d2 = d.x * d.x + d.y * d.y, n = sqrt(d2);
  double sm, cm;
  sincos(bm, &sm, &cm);
  if (n > 1e-5) {
    const double in = 1.0 / n;
    const double c = d.x * in, s = d.y * in;
    const double a = -s * in, b = c * in;  // -d.y / d2, d.x / d2
    rec[0] = -a;
    rec[1] = -b;
    rec[2] = a * d.y - b * d.x;
    rec[3] = -c;
    rec[4] = -s;
    rec[5] = 0.0;
    rec[6] = a * ps.c - b * ps.s;
    rec[7] = a * ps.s + b * ps.c;
    rec[8] = ps.c * c - ps.s * s;
    rec[9] = ps.s * c + ps.c * s;
    rec[10] = atan2(-sm * c + cm * s, cm * c + sm * s);
    rec[11] = n - rm;
  } else {  // (a landmark on top of the pose: the conventions of bearing_of / range_of)
    double Jx[6], Jl[4];
    (void)bearing_of<true>(ps, lm, Jx, Jl);
    const double rp = range_of<true>(ps, lm, Jx + 3, Jl + 2);
    for (int k = 0; k < 6; ++k) rec[k] = Jx[k];
    for (int k = 0; k < 4; ++k) rec[6 + k] = Jl[k];
    rec[10] = atan2(-sm, cm);
    rec[11] = rp - rm;
  }
}

__device__ __forceinline__ size_t up8(size_t b) { return (b + 7) & ~(size_t)7; }

// 1/x to double round-off: v_rcp_f64 + two Newton steps (the pivot inverse is on every thread's critical path)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}

#pragma clang fp contract(fast)
typedef double v4d __attribute__((ext_vector_type(4)));
// doubles of the LDS region of a packed N x N system swept by sweep_packed_fast: the packed lower triangle, or the sweep's
// panels (two pivot-column panels, two W panels, two E tiles, two diagonal-tile dumps) that alias it
// (+ 6 N + 64 behind the triangle: SlamCtx::front parks 18 doubles per pose there - up to N = 128 the panels' size covers it)
__host__ __device__ inline size_t sweep_region_doubles(size_t N) {
  const size_t a = N * (N + 1) / 2 + 6 * N + 64, b = 64 * N + 1024;
  return a > b ? a : b;
}
constexpr int kWaves = kThreads / 64;
// The landmark-first dense solve serves systems of up to kDenseTiles tile rows (N = 160: 53 poses).  Up to eight the sweep gives
// every wave ONE tile row (sweep_packed_fast); with nine and ten rows (43 .. 53 poses) two light rows share a wave (rows 1 + 2, and
// 3 + 4 with ten) - the same block-step loop with one barrier per step.  (Until round 6 such systems went through sweep_regtiles -
// the lower tiles dealt over seven waves, three barriers per step: 54-58 us against 26 at eight rows.)  Either keeps such updates off
// the pose-chain solver (k_slam_arrow.hip: ~230 us at 46 poses against ~65 us for the dense solve at 41).
constexpr int kDenseTiles = 10;

struct SweepCtx {
  int I, lane, lc, lr, np, N;
  bool live;   // this wave's tile row holds real rows (I < number of 16-row blocks in use)
  bool ewave;  // this wave inverts the diagonal tiles (an idle tile row if there is one, else tile row 0)
  int *bad;
  long long *tr;  // dev aid: 5 cycle stamps per wave for one step (armed through drlgx_debug_phase_clocks_host)
};

// ------------------------------------------------------------------------------------------------------------------
// 16-wide block Gauss-Jordan on lower tiles in MFMA accumulators (diagonal tiles kept fully symmetric).  Scalar branches
// cost ~20-30 cycles here and a block barrier ~50 plus the arrival skew, so a step pivots on a whole 16 x 16 tile column K
// (7 steps at 37 poses):
//   P  the pivot tile column is published to LDS: panel PAN[i][.] = A[i][16 K + .]
//   W  every wave: W_I = PAN_I E_K  (4 chained MFMAs, E_K = -D_K^-1 from the look-ahead below)
//   U  every wave: A_Iu += W_I PAN_u^T for its tiles u <= I (4 MFMAs each); tile column K <- -W_I; pivot rows <- -W_u^T,
//      pivot block <- E_K
//   look-ahead: the diagonal tile D_{K+1} = A_{K+1,K+1} + W_{K+1} PAN_{K+1}^T is formed and inverted inside ONE wave with
//      little or no matrix work while the others run U.
// The fast path (sweep_role, below) keeps the panels as MFMA operand images and needs one barrier per step; the helpers
// here (row-major "KS" panels: a 16-vector v is stored as v[(c & 3) * 4 + (c >> 2)] so that the 4 K-steps of an MFMA
// operand lane are one 32-byte read; tile_step16 / tile_replace16) serve sweep_regtiles and sweep_streamed.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ks16(int c) { return (c & 3) * 4 + (c >> 2); }

__device__ __forceinline__ v4d mfma4(const double (&a)[4], const double (&b)[4], v4d c) {
#pragma unroll
  for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ void ld4(const double *p, double (&o)[4]) {
  const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2);
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}

__device__ __forceinline__ double readlane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- in-wave 16 x 16 symmetric inversion: scalar Gauss-Jordan sweeps in registers ----
// d (accumulator layout: lane (lr, lc), reg r = D[lr + 4 r][lc], full symmetric tile K of the system) <- -D^-1; pivots
// 16 K + k >= np are skipped (identity).  Per pivot k: row k is broadcast to the four 16-lane rows with the gfx950
// permlane swaps, the column entries of a lane's rows and the pivot come from DPP row broadcasts: no LDS, no shuffles
// through memory - this dependent chain (16 reciprocals) is the critical path of the whole factorisation.
template <int kLane>
__device__ __forceinline__ double row_bcast_lane(double v) {  // value of lane kLane of each 16-lane row
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, 0x150 + kLane, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), 0x150 + kLane, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int kRow>
__device__ __forceinline__ double rowgroup_bcast(double v) {  // 16-lane row kRow (0..3) copied to all four rows
  const long long b = __double_as_longlong(v);
  unsigned w[2] = {(unsigned)b, (unsigned)(b >> 32)};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const auto p16 = __builtin_amdgcn_permlane16_swap(w[h], w[h], false, false);  // [x0 x0 x2 x2], [x1 x1 x3 x3]
    const unsigned y = (kRow & 1) ? p16[1] : p16[0];
    const auto p32 = __builtin_amdgcn_permlane32_swap(y, y, false, false);        // [A A A A], [B B B B]
    w[h] = (kRow & 2) ? p32[1] : p32[0];
  }
  return __longlong_as_double(((long long)w[1] << 32) | w[0]);
}
// One scalar pivot.  (xr, q) = (row k broadcast to every lane's column, 1 / D[k][k]) come from the previous step: the
// register that holds row k + 1 is updated first and the next pivot's broadcast + reciprocal chain is started from it,
// so that the rest of this pivot's update runs in the shadow of that chain.  ~50 VALU instructions per pivot at 4 cycles
// each is the floor of this formulation (wave64 on a 16-lane SIMD).
template <int k>
__device__ __forceinline__ void gj_update_reg(const SweepCtx &x, v4d &d, int r, double t, double q) {
  constexpr int rk = k >> 2, lk = k & 3;
  const bool colk = x.lc == k, rowk = x.lr == lk;
  const double c = row_bcast_lane<k>(d[r]);  // D[lr + 4 r][k]
  double v = fma(-c, t, d[r]);
  v = colk ? c * q : v;
  if (r == rk) v = rowk ? (colk ? -q : t) : v;
  d[r] = v;
}
template <int k>
__device__ __forceinline__ void gj_head(const SweepCtx &x, const v4d &d, double &xr, double &q) {
  constexpr int rk = k >> 2, lk = k & 3;
  xr = rowgroup_bcast<lk>(d[rk]);                    // D[k][lc]
  const double p = readlane_f64(d[rk], 16 * lk + k);  // D[k][k] (uniform; off the broadcast chain)
  q = fast_rcp(p);
  if (x.lane == 0 && !(p > 0)) x.bad[0] = 1;
}
template <int k, bool kChainNext>
__device__ __forceinline__ void gj_pivot(const SweepCtx &x, v4d &d, double &xr, double &q) {
  const double t = xr * q, qk = q;
  constexpr int rn = (k + 1 < 16) ? ((k + 1) >> 2) : 0;
  gj_update_reg<k>(x, d, rn, t, qk);
  if constexpr (kChainNext && k + 1 < 16) gj_head<k + 1>(x, d, xr, q);
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r != rn) gj_update_reg<k>(x, d, r, t, qk);
}
template <int k = 0>
__device__ __forceinline__ void inv16_masked(const SweepCtx &x, int K, v4d &d) {
  if constexpr (k < 16) {
    if (16 * K + k < x.np) {
      double xr, q;
      gj_head<k>(x, d, xr, q);
      gj_pivot<k, false>(x, d, xr, q);
      inv16_masked<k + 1>(x, K, d);
    }
  }
}
template <int k = 0>
__device__ __forceinline__ void inv16_full(const SweepCtx &x, v4d &d, double &xr, double &q) {
  if constexpr (k < 16) {
    gj_pivot<k, true>(x, d, xr, q);
    inv16_full<k + 1>(x, d, xr, q);
  }
}
__device__ __forceinline__ void inv16(const SweepCtx &x, int K, v4d &d) {
  // all 16 pivots active (every block but the last): one straight-line block, so that the scheduler can start pivot
  // k + 1's broadcast / reciprocal chain under the tail of pivot k's update
  if (16 * K + 16 <= x.np) {
    double xr, q;
    gj_head<0>(x, d, xr, q);
    inv16_full<0>(x, d, xr, q);
  } else {
    inv16_masked<0>(x, K, d);
  }
}

#pragma clang fp contract(on)
// ---- in-wave 16 x 16 SPD inversion by 4 x 4 BLOCK pivots on the fp64 matrix cores ----
// Same contract as inv16 (d: full symmetric tile in accumulator layout <- -D^-1 on the first `nact` pivots), four block
// steps instead of sixteen scalar ones.  Block step Kb (rows / columns 4 Kb .. 4 Kb + 3 = accumulator register Kb):
//   E4 = -(pivot block)^-1                  closed form (2 x 2 blocks, two reciprocals), from ten v_readlane values
//   W^T = E4 D[Kb rows, :]                  ONE MFMA: A operand = E4 (lanes lc < 4), B operand = register Kb as it is;
//                                           output register 0 at lane (lr, lc) = W[lc][lr] = the A operand of the update
//   D <- D + W (D[Kb rows, :] with the pivot columns replaced by -I)   ONE MFMA; with the pivot columns of the
//                                           accumulator input zeroed this leaves -W there, exactly
//   pivot rows <- -W^T, pivot block <- E4   selects
// The dependent chain per block is ~25 fp64 operations + two MFMAs instead of four scalar pivots of ~10 operations plus
// their permlane / DPP broadcasts (scripts/emul/inv16_blk_emul.py checks the index algebra against numpy).
struct Inv16Lane {  // lane constants of the block inversion
  bool lr1, c1, top, left, ua, ub, lc_lt4;
  double sel;  // -1 where (lc & 3) == lr, else 0: the "-I" of the pivot columns in the B operand
  __device__ __forceinline__ Inv16Lane(int lr, int lc) {
    lr1 = lr & 1; c1 = lc & 1; top = lr < 2; left = (lc & 3) < 2;
    ua = top ? c1 : lr1; ub = top ? lr1 : c1;
    lc_lt4 = lc < 4;
    sel = ((lc & 3) == lr) ? -1.0 : 0.0;
  }
};
template <int Kb>
__device__ __forceinline__ void inv16_blk_step(const SweepCtx &x, const Inv16Lane &q, v4d &d, bool &spd) {
  constexpr int c0 = 4 * Kb;
  const double a00 = readlane_f64(d[Kb], c0);
  const double a10 = readlane_f64(d[Kb], 16 + c0), a11 = readlane_f64(d[Kb], 16 + c0 + 1);
  const double a20 = readlane_f64(d[Kb], 32 + c0), a21 = readlane_f64(d[Kb], 32 + c0 + 1), a22 = readlane_f64(d[Kb], 32 + c0 + 2);
  const double a30 = readlane_f64(d[Kb], 48 + c0), a31 = readlane_f64(d[Kb], 48 + c0 + 1), a32 = readlane_f64(d[Kb], 48 + c0 + 2),
               a33 = readlane_f64(d[Kb], 48 + c0 + 3);
  // P = [a00 a10; a10 a11], Q = [a20 a21; a30 a31], R = [a22 a32; a32 a33]: block inverse through S = R - Q P^-1 Q^T.
  // (A variant that carries det P as a scale, so that the two reciprocals are not in sequence - dependent depth ~16
  // instead of ~29 operations - measured SLOWER, 3172 against 2988 cycles per tile: the step is bound by instruction
  // issue of the one wave that runs it, not by latency; scripts/micro/inv16_bench.hip.)
  const double detp = a00 * a11 - a10 * a10;
  const double ip = fast_rcp(detp);
  const double p00 = a11 * ip, p10 = -a10 * ip, p11 = a00 * ip;           // P^-1
  const double t00 = a20 * p00 + a21 * p10, t01 = a20 * p10 + a21 * p11;  // T = Q P^-1
  const double t10 = a30 * p00 + a31 * p10, t11 = a30 * p10 + a31 * p11;
  const double s00 = a22 - (t00 * a20 + t01 * a21);                       // S = R - T Q^T
  const double s10 = a32 - (t10 * a20 + t11 * a21);
  const double s11 = a33 - (t10 * a30 + t11 * a31);
  const double dets = s00 * s11 - s10 * s10;
  const double is = fast_rcp(dets);
  const double r00 = s11 * is, r10 = -s10 * is, r11 = s00 * is;           // S^-1
  const double u00 = r00 * t00 + r10 * t10, u01 = r00 * t01 + r10 * t11;  // U = S^-1 T
  const double u10 = r10 * t00 + r11 * t10, u11 = r10 * t01 + r11 * t11;
  spd = spd && (a00 > 0) && (detp > 0) && (s00 > 0) && (dets > 0);  // (tested once per tile: off the dependent chain)
  // this lane's entry E4[lr][lc & 3] of  E4 = -D^-1 = -[P^-1 + T^T U, -U^T; -U, S^-1]
  const double t0x = q.lr1 ? t01 : t00, t1x = q.lr1 ? t11 : t10;  // T[.][lr & 1]
  const double u0c = q.c1 ? u01 : u00, u1c = q.c1 ? u11 : u10;    // U[.][lc & 1]
  const double pI = (q.lr1 == q.c1) ? (q.lr1 ? p11 : p00) : p10;
  const double rI = (q.lr1 == q.c1) ? (q.lr1 ? r11 : r00) : r10;
  const double e_tl = -(pI + t0x * u0c + t1x * u1c);
  const double uo = q.ua ? (q.ub ? u11 : u10) : (q.ub ? u01 : u00);  // U[lr - 2][c] below the diagonal, U[c - 2][lr] above
  const double e_lane = (q.top == q.left) ? (q.top ? e_tl : -rI) : uo;
  const double eA = q.lc_lt4 ? e_lane : 0.0;
  const v4d z = {0.0, 0.0, 0.0, 0.0};
  const v4d wt4 = __builtin_amdgcn_mfma_f64_16x16x4f64(eA, d[Kb], z, 0, 0, 0);
  const double wt = wt4[0];  // lane (lr, lc): W[lc][lr],  W = D[:, Kb columns] E4
  const bool inK = (x.lc >> 2) == Kb;
  const double bop = inK ? q.sel : d[Kb];
  v4d cin;
#pragma unroll
  for (int r = 0; r < 4; ++r) cin[r] = inK ? 0.0 : d[r];
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(wt, bop, cin, 0, 0, 0);
  d[Kb] = inK ? e_lane : -wt;
}
// nact: number of pivots of this tile (1 .. 16); the rows / columns beyond are not pivots and their content afterwards is
// finite but meaningless (every user of E multiplies them by the zeroed panel columns or never reads them)
// kSkip: block steps whose four pivots are all inactive are left out (the incremental update's k x k systems, k <= 16: the
// inactive part is an identity block, decoupled from the rest)
template <bool kSkip = false>
__device__ __forceinline__ void inv16_blk(const SweepCtx &x, int nact, v4d &d) {
  if (nact < 16) {  // decouple the inactive rows / columns: identity there
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = x.lr + 4 * r;
      if (row >= nact || x.lc >= nact) d[r] = (row == x.lc) ? 1.0 : 0.0;
    }
  }
  const Inv16Lane q(x.lr, x.lc);
  bool spd = true;
  inv16_blk_step<0>(x, q, d, spd);
  if (!kSkip || nact > 4) inv16_blk_step<1>(x, q, d, spd);
  if (!kSkip || nact > 8) inv16_blk_step<2>(x, q, d, spd);
  if (!kSkip || nact > 12) inv16_blk_step<3>(x, q, d, spd);
  if (!spd && x.lane == 0) x.bad[0] = 1;
}

#pragma clang fp contract(fast)
// ---- the fast sweep as ONE runtime loop over the block steps ----
// (Round 2 instantiated a block step per tile column: 145 KB of straight-line code for seven steps - more than twice the
// 64 KB instruction cache, so the wave that inverts the diagonal tiles, alone on the critical path, ran its 16 pivots out
// of cold instruction fetches: 5.3 k cycles in the kernel against 3.8 k warm.)  One copy of the step now serves every
// tile column K; the accumulator tile "K" is selected by uniform branches over the statically indexed registers.
//
// LDS panels are stored as OPERAND IMAGES: a 16 x 16 block X is kept as the four MFMA operand registers of every lane,
//   img[(s >> 1) * 128 + 2 * lane + (s & 1)] = X[lc][4 s + lr],
// two lane-linear 16-byte halves (conflict-free ds_read_b128 / ds_write_b128; the row-major [16] KS layout of round 2 put
// every lane of a 16-lane group on two banks).  The same registers serve as the A operand of X . and as the B operand of
// . X^T.  With that, products are formed TRANSPOSED so that an MFMA result is directly the next MFMA's operand:
//   W_I^T = E_K PAN_I^T   (A = image of E_K (symmetric), B = image of PAN_I)   -> registers = image of W_I
//   A_Iu += W_I PAN_u^T   (A = those registers, B = image of PAN_u)
// and the E-wave's look-ahead  D_{K+1} += W_{K+1} PAN_{K+1}^T needs no LDS round trip any more.  A wave needs the W of
// other waves only for the pivot rows (A_Ku <- -W_u^T); that replacement is deferred until after the next step's barrier
// (W images double buffered), which leaves ONE workgroup barrier per block step instead of two.
struct SwL {  // every buffer twice (index = block step & 1); address arithmetic, no pointer tables (they would go to scratch)
  double *base;  // pan[2][16 N] operand images of the pivot tile column, wt[2][16 N] images of W_I = PAN_I E_K,
  int n16;       // einv[2][256] image of E_K = -D_K^-1 (= the accumulator registers of the inverting wave),
                 // dscr[2][256] accumulator registers of the next diagonal tile (lane-linear)
  __device__ __forceinline__ double *pan(int b) const { return base + b * n16; }
  __device__ __forceinline__ double *wt(int b) const { return base + (2 + b) * n16; }
  __device__ __forceinline__ double *einv(int b) const { return base + 4 * n16 + b * 256; }
  __device__ __forceinline__ double *dscr(int b) const { return base + 4 * n16 + 512 + b * 256; }
};
__device__ __forceinline__ void ld_op(const double *img, int lane, double (&o)[4]) {
  const double2 a = *reinterpret_cast<const double2 *>(img + 2 * lane), b = *reinterpret_cast<const double2 *>(img + 128 + 2 * lane);
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}
__device__ __forceinline__ void st_op(double *img, int lane, double v0, double v1, double v2, double v3) {
  *reinterpret_cast<double2 *>(img + 2 * lane) = make_double2(v0, v1);
  *reinterpret_cast<double2 *>(img + 128 + 2 * lane) = make_double2(v2, v3);
}
// offset inside an operand image of the element (row lr + 4 r, column lc) that a lane holds in accumulator layout
__device__ __forceinline__ int acc_off(int lr, int lc, int r) {
  return (lc >> 3) * 128 + 2 * (16 * (lc & 3) + lr + 4 * r) + ((lc >> 2) & 1);
}

// ONE TILE ROW of the sweep: the tiles (R, 0 .. R) in accumulator registers and what the block steps do to them.  R is a
// compile-time constant (R = -1: no row) - every register index except "tile column K" is static and the tile loops have no
// branches.  A role (below) owns one row or two.
template <int R>
struct SweepRow {
  static constexpr int NT = R >= 0 ? R + 1 : 1;
  v4d acc[NT];
  static __device__ __forceinline__ int AT(int i, int j) { return i * (i + 1) / 2 + j; }

  __device__ __forceinline__ void load(const double *A, int N, int lr, int lc) {
    if constexpr (R >= 0) {
#pragma unroll
      for (int u = 0; u <= R; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * R + lr + 4 * r, j = 16 * u + lc;
          acc[u][r] = (i < N && j < N) ? A[AT(max(i, j), min(i, j))] : 0.0;
        }
    }
  }
  // the first diagonal tile, for E_0
  __device__ __forceinline__ void dump_d0(const SwL &L, int lane) {
    if constexpr (R == 0) st_op(L.dscr(0), lane, acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
  }
  // ---- P: publish the pivot tile column (masked columns / rows as zeros) ----
  __device__ __forceinline__ void publish(const SwL &L, int K, int np, bool have_next, int lane, int lr, int lc, const int (&aoff)[4]) {
    if constexpr (R >= 0) {
      const int kb = 16 * K;
      double *pan = L.pan(K & 1);
      if (K <= R) {
        const bool colact = kb + lc < np;
        double *pI = pan + 256 * R;
#pragma unroll
        for (int u = 0; u <= R; ++u)  // (a ladder of scalar branches selects the statically indexed tile K)
          if (u == K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pI[aoff[r]] = colact ? acc[u][r] : 0.0;
          }
        if (K == R) {  // the transposed tiles (R, u < R): accumulator registers = operand image of PAN_u
#pragma unroll
          for (int u = 0; u < R; ++u)
            st_op(pan + 256 * u, lane, (kb + lr < np) ? acc[u][0] : 0.0, (kb + lr + 4 < np) ? acc[u][1] : 0.0,
                  (kb + lr + 8 < np) ? acc[u][2] : 0.0, (kb + lr + 12 < np) ? acc[u][3] : 0.0);
        }
      }
      if (have_next && K + 1 == R)  // current values of the next diagonal tile, for the look-ahead
        st_op(L.dscr((K + 1) & 1), lane, acc[NT - 1][0], acc[NT - 1][1], acc[NT - 1][2], acc[NT - 1][3]);
    }
  }
  // ---- deferred from step K - 1: its pivot rows A_{K-1,u} <- -(W_u)^T (all rows active: only the last block is masked) ----
  __device__ __forceinline__ void deferred(const SwL &L, int K, int lane) {
    if constexpr (R >= 1) {
      if (K == R + 1) {
        const double *wp = L.wt((K - 1) & 1);
#pragma unroll
        for (int u = 0; u < R; ++u) {
          double t[4];
          ld_op(wp + 256 * u, lane, t);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u][r] = -t[r];
        }
      }
    }
  }
  // ---- W, U ----
  __device__ __forceinline__ void update(const SwL &L, int K, int np, int lane, int lr, int lc, const int (&aoff)[4]) {
    if constexpr (R >= 0) {
      const int kb = 16 * K, b = K & 1;
      const bool has_mask = np < kb + 16;  // the last block holds the rhs row / pads: they are not pivots
      const double *pan = L.pan(b);
      double aP[4], eB[4];
      ld_op(pan + 256 * R, lane, aP);
      ld_op(L.einv(b), lane, eB);
      v4d wv = {0.0, 0.0, 0.0, 0.0};
      wv = mfma4(eB, aP, wv);  // image of W_R
      double *wI = L.wt(b) + 256 * R;
      st_op(wI, lane, wv[0], wv[1], wv[2], wv[3]);
      const double aW[4] = {wv[0], wv[1], wv[2], wv[3]};
      if (K != R || has_mask) {
        // A_Ru += W_R PAN_u^T (tile column K is replaced below, except in wave K whose masked rows keep the update); the
        // next tile's operand is loaded while this tile's MFMAs run
        // two tiles at a time: their MFMA chains are independent, so the matrix pipe is issued back to back (a chain on
        // ONE accumulator waits ~20 cycles per link for the previous result)
        double bP[2][2][4];
        ld_op(pan, lane, bP[0][0]);
        if (R >= 1) ld_op(pan + 256, lane, bP[0][1]);
#pragma unroll
        for (int u = 0; u <= R; u += 2) {
          constexpr int R1 = R >= 0 ? R : 0;
          const int h = (u >> 1) & 1, u1 = u + 1 <= R1 ? u + 1 : u;
          if (u + 2 <= R) ld_op(pan + 256 * (u + 2), lane, bP[h ^ 1][0]);
          if (u + 3 <= R) ld_op(pan + 256 * (u + 3), lane, bP[h ^ 1][1]);
          const bool d0 = u != K || K == R, d1 = u + 1 <= R && (u + 1 != K || K == R);
          if (d0 && d1) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
              acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[s2], bP[h][0][s2], acc[u], 0, 0, 0);
              acc[u1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aW[s2], bP[h][1][s2], acc[u1], 0, 0, 0);
            }
          } else if (d0) {
            acc[u] = mfma4(aW, bP[h][0], acc[u]);
          } else if (d1) {
            acc[u1] = mfma4(aW, bP[h][1], acc[u1]);
          }
        }
      }
      if (K <= R) {
        wave_lds_sync();  // own image -> accumulator layout
        double w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = wI[aoff[r]];
        if (K < R) {
#pragma unroll
          for (int u = 0; u < R; ++u)
            if (u == K) {
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[u][r] = -w[r];  // A_RK <- A_RK D^-1 (masked columns: W = 0)
            }
        } else {
          // pivot block <- E_K; rows >= np (rhs, pads) keep the regular update, their pivot columns take -W like any other
          // row; the pivot rows of the tiles (K, u < K) follow after the next barrier
          const bool colact = kb + lc < np;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool rowact = kb + lr + 4 * r < np;
            acc[NT - 1][r] = rowact ? (colact ? eB[r] : -aW[r]) : (colact ? -w[r] : acc[NT - 1][r]);
          }
        }
      }
    }
  }
  // the pivot rows of the last block (masked: rows >= np keep their regular update)
  __device__ __forceinline__ void last_rows(const SwL &L, int nK, int np, int lane, int lr) {
    if constexpr (R >= 1) {
      if (R == nK - 1) {
        const int kb = 16 * R;
        const double *wp = L.wt(R & 1);
#pragma unroll
        for (int u = 0; u < R; ++u) {
          double t[4];
          ld_op(wp + 256 * u, lane, t);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[u][r] = (kb + lr + 4 * r < np) ? -t[r] : acc[u][r];
        }
      }
    }
  }
  __device__ __forceinline__ void store(double *A, int N, int lr, int lc) const {
    if constexpr (R >= 0) {
#pragma unroll
      for (int u = 0; u <= R; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * R + lr + 4 * r, j = 16 * u + lc;
          if (j <= i && i < N) A[AT(i, j)] = acc[u][r];
        }
    }
  }
};

// The sweep of ONE ROLE: the wave that owns tile row I (I = -1: none), a second one I2 (the nine- and ten-row systems of 43 .. 53
// poses: two light rows share a wave, so that every wave still runs ONE block-step loop with ONE barrier per step) and, with kE,
// inverts the diagonal tiles.  The block steps are a runtime loop, so each wave runs a few KB of code that stays in the
// instruction cache.
// A: the packed lower triangle (LDS); the panels alias it once the tiles are in registers.  Every role executes the same
// sequence of workgroup barriers.
// have_e0 (kE only): e0 = E_0 = -D_0^-1 as the caller inverted it already (SlamCtx::back does, under the Schur phase); by
// value - a pointer to it would put it into scratch memory
template <int I, bool kE, int I2 = -1>
__device__ __forceinline__ void sweep_role(const DrlgxState &S, const SweepCtx &x, double *A, int N, bool have_e0 = false,
                                           v4d e0 = v4d{0.0, 0.0, 0.0, 0.0}) {
  const int lane = x.lane, lc = x.lc, lr = x.lr, np = x.np;
  const int nK = (np + 15) >> 4;
  SweepRow<I> r1;
  SweepRow<I2> r2;
  r1.load(A, N, lr, lc);
  r2.load(A, N, lr, lc);
  __syncthreads();  // every tile is in registers: the LDS region of A now holds the sweep panels
  const SwL L{A, 16 * N};
  r1.dump_d0(L, lane);
  r2.dump_d0(L, lane);
  __syncthreads();
  if constexpr (kE) {  // E_0
    if (have_e0) {
      st_op(L.einv(0), lane, e0[0], e0[1], e0[2], e0[3]);
    } else {
      double t[4];
      ld_op(L.dscr(0), lane, t);
      v4d d = {t[0], t[1], t[2], t[3]};
      inv16_blk(x, min(16, np), d);
      st_op(L.einv(0), lane, d[0], d[1], d[2], d[3]);
    }
  }
  const int aoff[4] = {acc_off(lr, lc, 0), acc_off(lr, lc, 1), acc_off(lr, lc, 2), acc_off(lr, lc, 3)};
#pragma clang loop unroll(disable)
  for (int K = 0; K < nK; ++K) {
    const int kb = 16 * K, b = K & 1;
    const bool have_next = kb + 16 < np;
    const bool trg = x.tr && K == 3;
    if (trg) x.tr[0] = clock64();
    r1.publish(L, K, np, have_next, lane, lr, lc, aoff);
    r2.publish(L, K, np, have_next, lane, lr, lc, aoff);
    if (trg) x.tr[1] = clock64();
    __syncthreads();  // panels of step K, E_K, the W images of step K - 1
    const double *pan = L.pan(b);
    r1.deferred(L, K, lane);
    r2.deferred(L, K, lane);
    // ---- look-ahead (critical path): E_{K+1} = -(D_{K+1} + W_{K+1} PAN_{K+1}^T)^-1 ----
    if constexpr (kE) {
      if (have_next) {
        // (this chain is the critical path of the whole sweep: it outranks the SIMD partner's update work)
        __builtin_amdgcn_s_setprio(3);
        double aP[4], eB[4], t[4];
        ld_op(pan + 256 * (K + 1), lane, aP);
        ld_op(L.einv(b), lane, eB);
        ld_op(L.dscr((K + 1) & 1), lane, t);
        v4d w1 = {0.0, 0.0, 0.0, 0.0};
        w1 = mfma4(eB, aP, w1);  // image of W_{K+1}
        const double aW[4] = {w1[0], w1[1], w1[2], w1[3]};
        v4d dn = {t[0], t[1], t[2], t[3]};
        dn = mfma4(aW, aP, dn);
        if (trg) x.tr[2] = clock64();
        inv16_blk(x, min(16, np - kb - 16), dn);
        st_op(L.einv((K + 1) & 1), lane, dn[0], dn[1], dn[2], dn[3]);
        __builtin_amdgcn_s_setprio(0);
        if (trg) x.tr[3] = clock64();
      }
    }
    r1.update(L, K, np, lane, lr, lc, aoff);
    r2.update(L, K, np, lane, lr, lc, aoff);
    if (trg) x.tr[4] = clock64();
  }
  __syncthreads();
  r1.last_rows(L, nK, np, lane, lr);
  r2.last_rows(L, nK, np, lane, lr);
  __syncthreads();
  r1.store(A, N, lr, lc);
  r2.store(A, N, lr, lc);
}

// iterate the poses p (ascending) whose bit is set in the W-word mask at `mk`
#define FOR_EACH_OBSERVING_POSE(mk, W, p)                                      \
  for (int _w = 0; _w < (W); ++_w)                                            \
    for (unsigned long long _m = (mk)[_w]; _m; _m &= _m - 1)                  \
      if (const int p = 64 * _w + __ffsll((long long)_m) - 1; true)

// the part of a block step that overwrites instead of updating: tile column K takes -W, the pivot rows -(W)^T, the pivot
// block E_K (masked rows / columns excepted)
__device__ __forceinline__ void tile_replace16(int I, int J, int K, int np, int lc, int lr, const double *wt, const double *einv,
                                               v4d &acc) {
  const int kb = 16 * K;
  if (I > K && J == K) {  // A_IK <- A_IK D^-1 (masked columns: W = 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = -wt[(16 * I + lr + 4 * r) * 16 + ks16(lc)];
  }
  if (I == K) {
    if (J < K) {  // pivot rows: A_KJ <- -(W_J)^T; rows >= np (rhs, pads) keep the regular update
      double tt[4];
      ld4(wt + (16 * J + lc) * 16 + lr * 4, tt);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = (kb + lr + 4 * r < np) ? -tt[r] : acc[r];
    } else {  // pivot block <- E_K; masked rows / columns take -W like any other row
      double ee[4], tt[4];
      ld4(einv + lc * 16 + lr * 4, ee);
      ld4(wt + (16 * K + lc) * 16 + lr * 4, tt);
      const bool colact = kb + lc < np;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool rowact = kb + lr + 4 * r < np;
        const double wv = wt[(16 * K + lr + 4 * r) * 16 + ks16(lc)];
        acc[r] = rowact ? (colact ? ee[r] : -tt[r]) : (colact ? -wv : acc[r]);
      }
    }
  }
}

// One lower tile (I, J) of block step K of the 16-wide symmetric sweep, for the variants that do not keep a whole tile row
// per wave: the update / replacement rules above, from the LDS panels PAN (pivot column), WT = PAN E_K
// and E_K.  acc: accumulator layout (row lr + 4 r, column lc of the tile; diagonal tiles fully symmetric).
__device__ __forceinline__ void tile_step16(int I, int J, int K, int np, int lc, int lr, const double *pan, const double *wt,
                                            const double *einv, v4d &acc) {
  const int kb = 16 * K;
  const bool has_mask = np < kb + 16;
  if ((I != K || has_mask) && (J != K || I == K)) {
    double aW[4], bP[4];
    ld4(wt + (16 * I + lc) * 16 + lr * 4, aW);
    ld4(pan + (16 * J + lc) * 16 + lr * 4, bP);
    acc = mfma4(aW, bP, acc);
  }
  tile_replace16(I, J, K, np, lc, lr, wt, einv, acc);
}
// Pose i's diagonal block B (3x3, full) and gradient g of the prior / odometry (odo[i] = measured odometry between the
// poses i and i + 1: x, y, cos, sin) / own bearing-range factors linearised at thp, and - for i + 1 < P - the block O = (i + 1, i) of the odometry factor i (SLAM2D.cpp:44-89; records: linearize_br)
__device__ __forceinline__ void pose_block(const DrlgxState &S, int inst, const double *thp, const double *odo, const double *rec, const int *mstart,
                                           int i, int P, double wb, double wr, double *B, double *g, double *O) {
  const drlgx_config &cfg = S.cfg;
  for (int k = 0; k < 9; ++k) B[k] = 0.0;
  for (int k = 0; k < 3; ++k) g[k] = 0.0;
  const Pose ti{thp[4 * i], thp[4 * i + 1], thp[4 * i + 2], thp[4 * i + 3]};
  if (i == 0) {  // prior (SLAM2D.cpp:44-57): e = Local(prior, x0), J = diag(R_h^T, 1), W = information
    const double *pr = S.prior + (size_t)inst * DRLGX_PRIOR_STRIDE;
    const Pose h = between(Pose{pr[0], pr[1], pr[2], pr[3]}, ti, nullptr);
    const double e[3] = {h.x, h.y, theta_of(h)};
    const double J[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
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
  if (i > 0) {  // odometry factor i-1 seen from its second key: J2 = Hlocal (SLAM2D.cpp:59-89)
    const double *oo = odo + 4 * (i - 1);
    const Pose tm{thp[4 * (i - 1)], thp[4 * (i - 1) + 1], thp[4 * (i - 1) + 2], thp[4 * (i - 1) + 3]};
    const Pose hx = between(tm, ti, nullptr);
    const Pose h = between(Pose{oo[0], oo[1], oo[2], oo[3]}, hx, nullptr);
    const double e[3] = {h.x, h.y, theta_of(h)};
    const double J2[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        B[r * 3 + c] += J2[r] * wo[0] * J2[c] + J2[3 + r] * wo[1] * J2[3 + c] + J2[6 + r] * wo[2] * J2[6 + c];
      g[r] += J2[r] * wo[0] * e[0] + J2[3 + r] * wo[1] * e[1] + J2[6 + r] * wo[2] * e[2];
    }
  }
  if (i + 1 < P) {  // odometry factor i from its first key: J1 = Hlocal * H1; also block (i+1, i) = J2^T W J1
    const double *oo = odo + 4 * i;
    const Pose tn{thp[4 * (i + 1)], thp[4 * (i + 1) + 1], thp[4 * (i + 1) + 2], thp[4 * (i + 1) + 3]};
    double H1[9];
    const Pose hx = between(ti, tn, H1);
    const Pose h = between(Pose{oo[0], oo[1], oo[2], oo[3]}, hx, nullptr);
    const double e[3] = {h.x, h.y, theta_of(h)};
    const double Hl[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
    double J1[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) J1[r * 3 + c] = Hl[r * 3] * H1[c] + Hl[r * 3 + 1] * H1[3 + c] + Hl[r * 3 + 2] * H1[6 + c];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        B[r * 3 + c] += J1[r] * wo[0] * J1[c] + J1[3 + r] * wo[1] * J1[3 + c] + J1[6 + r] * wo[2] * J1[6 + c];
      g[r] += J1[r] * wo[0] * e[0] + J1[3 + r] * wo[1] * e[1] + J1[6 + r] * wo[2] * e[2];
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        O[r * 3 + c] = Hl[r] * wo[0] * J1[c] + Hl[3 + r] * wo[1] * J1[3 + c] + Hl[6 + r] * wo[2] * J1[6 + c];
  }
  for (int m = mstart[i]; m < mstart[i + 1]; ++m) {  // own bearing-range factors
    const double *l = rec + (size_t)REC * m;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) B[r * 3 + c] += l[r] * wb * l[c] + l[3 + r] * wr * l[3 + c];
      g[r] += l[r] * wb * l[10] + l[3 + r] * wr * l[11];
    }
  }
}

// ---- pose_block in pieces, for the LDS-resident solver's front end (one thread per pose ran the whole of it: ~3 us of
// serial fp64 per pose - two atan2 among it - plus ~0.3 us per own factor, in ONE wave, while the others idled) ----
// lower triangle (xx, yx, yy, tx, ty, tt) + gradient of one bearing-range factor seen from its pose, added to B6 / g
__device__ __forceinline__ void own_factor_add(const double *l, double wb, double wr, double *B6, double *g) {
  for (int r = 0, q = 0; r < 3; ++r) {
    for (int c = 0; c <= r; ++c, ++q) B6[q] += l[r] * wb * l[c] + l[3 + r] * wr * l[3 + c];
    g[r] += l[r] * wb * l[10] + l[3 + r] * wr * l[11];
  }
}
// the prior on pose 0 (SLAM2D.cpp:44-57): its block (lower triangle) and gradient
__device__ __forceinline__ void prior_factor(const DrlgxState &S, int inst, const double *thp, double *B6, double *g) {
  const Pose t0{thp[0], thp[1], thp[2], thp[3]};
  const double *pr = S.prior + (size_t)inst * DRLGX_PRIOR_STRIDE;
  const Pose h = between(Pose{pr[0], pr[1], pr[2], pr[3]}, t0, nullptr);
  const double e[3] = {h.x, h.y, theta_of(h)};
  const double J[9] = {h.c, h.s, 0, -h.s, h.c, 0, 0, 0, 1};
  const double *W = pr + 4;
  double WJ[9], We[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) WJ[r * 3 + c] = W[r * 3] * J[c] + W[r * 3 + 1] * J[3 + c] + W[r * 3 + 2] * J[6 + c];
    We[r] = W[r * 3] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2];
  }
  for (int r = 0, q = 0; r < 3; ++r) {
    for (int c = 0; c <= r; ++c, ++q) B6[q] = J[r] * WJ[c] + J[3 + r] * WJ[3 + c] + J[6 + r] * WJ[6 + c];
    g[r] = J[r] * We[0] + J[3 + r] * We[1] + J[6 + r] * We[2];
  }
}
// odometry factor i (poses i, i + 1; SLAM2D.cpp:59-89) linearised ONCE: what it adds to the block / gradient of its first
// key (C1, g1) and of its second key (C2, g2), and the off-diagonal block O = (i + 1, i)
__device__ __forceinline__ void odo_factor(const DrlgxState &S, const double *thp, const double *odo, int i, double *C1, double *g1,
                                           double *C2, double *g2, double *O) {
  const double wt = S.w_trans, wr = S.w_rot;  // W = diag(wt, wt, wr)
  const Pose ti{thp[4 * i], thp[4 * i + 1], thp[4 * i + 2], thp[4 * i + 3]};
  const Pose tn{thp[4 * (i + 1)], thp[4 * (i + 1) + 1], thp[4 * (i + 1) + 2], thp[4 * (i + 1) + 3]};
  const double *oo = odo + 4 * i;
  double H1[9];
  const Pose hx = between(ti, tn, H1);
  const Pose h = between(Pose{oo[0], oo[1], oo[2], oo[3]}, hx, nullptr);
  const double e0 = h.x, e1 = h.y, e2 = theta_of(h);
  // Jacobians: second key Hl = [h.c h.s 0; -h.s h.c 0; 0 0 1], first key J1 = Hl H1 with H1 = [. . .; . . .; 0 0 -1] - the
  // zero / unit entries are written out (the generic 3x3 products spend two thirds of their operations on them)
  const double j00 = h.c * H1[0] + h.s * H1[3], j01 = h.c * H1[1] + h.s * H1[4], j02 = h.c * H1[2] + h.s * H1[5];
  const double j10 = h.c * H1[3] - h.s * H1[0], j11 = h.c * H1[4] - h.s * H1[1], j12 = h.c * H1[5] - h.s * H1[2];
  // C1 = J1^T W J1 (lower: xx yx yy tx ty tt), g1 = J1^T W e;  row 2 of J1 = (0, 0, -1)
  C1[0] = wt * (j00 * j00 + j10 * j10);
  C1[1] = wt * (j01 * j00 + j11 * j10);
  C1[2] = wt * (j01 * j01 + j11 * j11);
  C1[3] = wt * (j02 * j00 + j12 * j10);
  C1[4] = wt * (j02 * j01 + j12 * j11);
  C1[5] = wt * (j02 * j02 + j12 * j12) + wr;
  g1[0] = wt * (j00 * e0 + j10 * e1);
  g1[1] = wt * (j01 * e0 + j11 * e1);
  g1[2] = wt * (j02 * e0 + j12 * e1) - wr * e2;
  // C2 = Hl^T W Hl, g2 = Hl^T W e
  const double n2 = h.c * h.c + h.s * h.s;
  C2[0] = wt * n2; C2[1] = 0.0; C2[2] = wt * n2; C2[3] = 0.0; C2[4] = 0.0; C2[5] = wr;
  g2[0] = wt * (h.c * e0 - h.s * e1);
  g2[1] = wt * (h.s * e0 + h.c * e1);
  g2[2] = wr * e2;
  // O = Hl^T W J1 = block (i + 1, i)
  O[0] = wt * (h.c * j00 - h.s * j10); O[1] = wt * (h.c * j01 - h.s * j11); O[2] = wt * (h.c * j02 - h.s * j12);
  O[3] = wt * (h.s * j00 + h.c * j10); O[4] = wt * (h.s * j01 + h.c * j11); O[5] = wt * (h.s * j02 + h.c * j12);
  O[6] = 0.0; O[7] = 0.0; O[8] = -wr;
}
// sum over the aligned groups of 8 lanes, in the lane with (lane & 7) == 7: DPP row shifts, no LDS traffic
template <int kCtrl>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, kCtrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), kCtrl, 0xf, 0xf, true);
  return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// the value of another lane of the same quad: kCtrl = DPP quad_perm (0xB1: lane ^ 1, 0x4E: lane ^ 2)
template <int kCtrl>
__device__ __forceinline__ double dpp_quad_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, kCtrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), kCtrl, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double sum8_lane7(double v) {
  v = dpp_add_f64<0x111>(v);  // row_shr:1
  v = dpp_add_f64<0x112>(v);  // row_shr:2
  v = dpp_add_f64<0x114>(v);  // row_shr:4
  return v;
}

// Symmetric Gauss-Jordan sweep of the packed lower triangle `A` (LDS, row i at i (i + 1) / 2, N = 16 Tn <= 16 FT rows; the
// region must hold sweep_region_doubles(N) doubles: the sweep panels alias it while the tiles are in registers) on
// the pivots [0, np); rows >= np (the rhs row np, pads) are carried along.  Afterwards A holds -A_pp^-1 and row np the
// solution.  All kThreads threads of the workgroup call it (block barriers inside).
template <int FT>
__device__ __forceinline__ void sweep_packed_fast(const DrlgxState &S, double *A, int np, int N, int Tn, int *bad, int tid, bool have_e0 = false,
                                                  v4d e0 = v4d{0.0, 0.0, 0.0, 0.0}) {
  static_assert(FT == 8, "one role per wave of the 512-thread workgroup");
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (Tn > FT) {
    // Nine or ten tile rows (43 .. 53 poses): the wave of row 0 inverts the diagonal tiles, rows 1 + 2 share a wave (five tiles), with
    // ten rows 3 + 4 too (nine); waves w and w + 4 share a SIMD - light next to heavy:
    //   ten rows:  SIMD 0: {E, 0} + {9}   1: {1, 2} + {8}   2: {3, 4} + {5}   3: {6} + {7}      (11 / 14 / 15 / 15 tiles)
    //   nine rows: SIMD 0: {E, 0} + {8}   1: {1, 2} + {7}   2: {3} + {6}      3: {4} + {5}      (10 / 13 / 11 / 11)
    const SweepCtx x{0, lane, lane & 15, lane >> 4, np, N, true, wv == 0, bad,
                     (S.prof && blockIdx.x == S.prof_block && lane == 0) ? S.prof + 64 + 5 * wv : nullptr};
    const bool ten = Tn == 10;
    switch (wv) {
      case 0: sweep_role<0, true>(S, x, A, N); break;
      case 1: sweep_role<1, false, 2>(S, x, A, N); break;
      case 2:
        if (ten) sweep_role<3, false, 4>(S, x, A, N);
        else sweep_role<3, false>(S, x, A, N);
        break;
      case 3:
        if (ten) sweep_role<6, false>(S, x, A, N);
        else sweep_role<4, false>(S, x, A, N);
        break;
      case 4:
        if (ten) sweep_role<9, false>(S, x, A, N);
        else sweep_role<8, false>(S, x, A, N);
        break;
      case 5:
        if (ten) sweep_role<8, false>(S, x, A, N);
        else sweep_role<7, false>(S, x, A, N);
        break;
      case 6:
        if (ten) sweep_role<5, false>(S, x, A, N);
        else sweep_role<6, false>(S, x, A, N);
        break;
      default:
        if (ten) sweep_role<7, false>(S, x, A, N);
        else sweep_role<5, false>(S, x, A, N);
        break;
    }
    return;
  }
  // tile rows r and FT-1-r share a SIMD (waves w and w+4): lower-triangle MFMA work is balanced across the SIMDs
  int trow = wv < FT / 2 ? wv : (FT - 1) - (wv - FT / 2);
  if (Tn == FT) {
    // no idle tile row: the wave of row 0 (one tile of update work) also inverts the diagonal tiles, and its SIMD
    // partner takes the next lightest row, so that the inversion chain competes with the fewest MFMAs:
    // SIMD pairs (0, 1), (2, FT-1), (3, FT-2), ...
    trow = wv == 0 ? 0 : wv == FT / 2 ? 1 : wv < FT / 2 ? wv + 1 : FT + FT / 2 - wv;
  }
  const bool live = trow < Tn, ewave = trow == (Tn < FT ? FT - 1 : 0);
  const SweepCtx x{trow, lane, lane & 15, lane >> 4, np, N, live, ewave, bad,
                   (S.prof && blockIdx.x == S.prof_block && lane == 0) ? S.prof + 64 + 5 * wv : nullptr};
  if (!live) {
    if (ewave) sweep_role<-1, true>(S, x, A, N, have_e0, e0);
    else sweep_role<-1, false>(S, x, A, N);
    return;
  }
  switch (trow) {
    case 0:
      if (ewave) sweep_role<0, true>(S, x, A, N);
      else sweep_role<0, false>(S, x, A, N);
      break;
    case 1: sweep_role<1, false>(S, x, A, N); break;
    case 2: sweep_role<2, false>(S, x, A, N); break;
    case 3: sweep_role<3, false>(S, x, A, N); break;
    case 4: sweep_role<4, false>(S, x, A, N); break;
    case 5: sweep_role<5, false>(S, x, A, N); break;
    case 6: sweep_role<6, false>(S, x, A, N); break;
    default: sweep_role<7, false>(S, x, A, N); break;
  }
}

// 16-wide block steps with the lower tiles in registers: `A` is the packed lower triangle in LDS
// (kPackedA, the panels `pbase` may alias it: every tile is in registers before the first panel is written) or the square
// matrix with leading dimension N in the HBM/L2 workspace (panels `pbase` in LDS, 32 N + 1280 doubles).
template <bool kPackedA, int NTW>
__device__ __forceinline__ void sweep_regtiles(double *A, double *pbase, int np, int N, int Tn, int ntiles, int *bad, int tid) {
  const int ld = N;
  auto AT = [&](int i, int j) -> int { return kPackedA ? i * (i + 1) / 2 + j : i * ld + j; };
  constexpr bool kLds = kPackedA;
  {
    constexpr int TW = kWaves - 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lc = lane & 15, lr = lane >> 4;
    const bool ewave = wave == TW;
    double *pan = pbase, *wt = pbase + 16 * N, *einv0 = pbase + 32 * N, *dscr = einv0 + 512, *es = dscr + 256;
    const SweepCtx x{0, lane, lc, lr, np, N, true, ewave, bad, nullptr};
    v4d acc[NTW];
    int tI[NTW], tJ[NTW];
    bool live[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const int t = wave + TW * u;
      live[u] = !ewave && t < ntiles;
      int ib = 0, jb = 0;
      if (live[u]) {
        ib = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
        while ((ib + 1) * (ib + 2) / 2 <= t) ++ib;
        while (ib * (ib + 1) / 2 > t) --ib;
        jb = t - ib * (ib + 1) / 2;
      }
      tI[u] = ib;
      tJ[u] = jb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ib + lr + 4 * r, j = 16 * jb + lc;
        acc[u][r] = live[u] ? A[AT(max(i, j), min(i, j))] : 0.0;
      }
    }
    if (kLds) __syncthreads();  // every tile is in registers before the panels overwrite the matrix region
    // E_0 from the first diagonal tile (tile 0 = wave 0, slot 0)
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dscr[4 * lane + r] = acc[0][r];
    }
    __syncthreads();
    if (ewave) {
      double t4[4];
      ld4(dscr + 4 * lane, t4);
      v4d d = {t4[0], t4[1], t4[2], t4[3]};
      inv16(x, 0, d);
#pragma unroll
      for (int r = 0; r < 4; ++r) einv0[(lr + 4 * r) * 16 + ks16(lc)] = d[r];
    }
    __syncthreads();
    for (int K = 0; 16 * K < np; ++K) {
      const int kb = 16 * K;
      const bool have_next = kb + 16 < np;
      double *einv = einv0 + 256 * (K & 1), *enext = einv0 + 256 * ((K + 1) & 1);
      // P: publish the pivot column panel from the tiles of column K and (transposed) of row K; the owner of the next
      // diagonal tile dumps its current values for the look-ahead
#pragma unroll
      for (int u = 0; u < NTW; ++u) {
        if (!live[u]) continue;
        if (tJ[u] == K) {
          const bool colact = kb + lc < np;
#pragma unroll
          for (int r = 0; r < 4; ++r) pan[(16 * tI[u] + lr + 4 * r) * 16 + ks16(lc)] = colact ? acc[u][r] : 0.0;
        } else if (tI[u] == K) {  // tJ < K: PAN[16 J + lc][c = lr + 4 r] = A[kb + lr + 4 r][16 J + lc]
          double *o = pan + (16 * tJ[u] + lc) * 16 + lr * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (kb + lr + 4 * r < np) ? acc[u][r] : 0.0;
        }
        if (have_next && tI[u] == K + 1 && tJ[u] == K + 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dscr[4 * lane + r] = acc[u][r];
        }
      }
      __syncthreads();
      v4d dn = {0.0, 0.0, 0.0, 0.0};
      if (!ewave) {
        for (int I = wave; I < Tn; I += TW) {  // W_I = PAN_I E_K
          double aP[4], eB[4];
          ld4(pan + (16 * I + lc) * 16 + lr * 4, aP);
          ld4(einv + lc * 16 + lr * 4, eB);
          v4d w = {0.0, 0.0, 0.0, 0.0};
          w = mfma4(aP, eB, w);
#pragma unroll
          for (int r = 0; r < 4; ++r) wt[(16 * I + lr + 4 * r) * 16 + ks16(lc)] = w[r];
        }
      } else if (have_next) {  // look-ahead: D'_{K+1} (the E-wave forms W_{K+1} itself)
        double aP[4], eB[4], aW[4], t4[4];
        ld4(pan + (16 * (K + 1) + lc) * 16 + lr * 4, aP);  // also the B operand of the update (PAN_{K+1}^T)
        ld4(einv + lc * 16 + lr * 4, eB);
        v4d w1 = {0.0, 0.0, 0.0, 0.0};
        w1 = mfma4(aP, eB, w1);
#pragma unroll
        for (int r = 0; r < 4; ++r) es[(lr + 4 * r) * 16 + ks16(lc)] = w1[r];  // accumulator -> A-operand layout
        wave_lds_sync();
        ld4(es + lc * 16 + lr * 4, aW);
        ld4(dscr + 4 * lane, t4);
        dn = v4d{t4[0], t4[1], t4[2], t4[3]};
        dn = mfma4(aW, aP, dn);
      }
      __syncthreads();
      if (ewave) {
        if (have_next) {
          inv16(x, K + 1, dn);
#pragma unroll
          for (int r = 0; r < 4; ++r) enext[(lr + 4 * r) * 16 + ks16(lc)] = dn[r];
        }
      } else {
#pragma unroll
        for (int u = 0; u < NTW; ++u)
          if (live[u]) tile_step16(tI[u], tJ[u], K, np, lc, lr, pan, wt, einv, acc[u]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      if (!live[u]) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * tI[u] + lr + 4 * r, j = 16 * tJ[u] + lc;
        if (j <= i) A[AT(i, j)] = acc[u][r];
      }
    }
  }
}

// 16-wide block steps with EVERYTHING in the HBM / L2 workspace: the square matrix A (leading dimension N, lower triangle
// valid), the pivot-column panel and W (pws: 32 N doubles); LDS holds the two E tiles and the look-ahead scratch only
// (lds_s: 1280 doubles).  This is the landmark system of k_slam_arrow beyond what its register-tile sweep holds (> 127
// landmarks - the reference has no cap, SLAM2D.cpp:103-124): every lower tile is read, updated on the matrix cores and
// written back once per block step (8 bytes x N^2 / 2 of traffic per step - a few hundred MB per update at 500 landmarks;
// this path exists so that such worlds RUN, not to be fast).  Same update / replacement rules as sweep_regtiles.
__device__ __forceinline__ void sweep_streamed(double *A, double *pws, double *lds_s, int np, int N, int Tn, int *bad, int tid) {
  const int ld = N;
  constexpr int TW = kWaves - 1;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lc = lane & 15, lr = lane >> 4;
  const bool ewave = wave == TW;
  double *pan = pws, *wt = pws + 16 * (size_t)N;
  double *einv0 = lds_s, *dscr = lds_s + 512, *es = dscr + 256;
  const SweepCtx x{0, lane, lc, lr, np, N, true, ewave, bad, nullptr};
  const int ntiles = Tn * (Tn + 1) / 2;
  // element (i, j) of the symmetric matrix from its stored lower triangle
  auto sym = [&](int i, int j) -> double { return A[(size_t)max(i, j) * ld + min(i, j)]; };
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dscr[4 * lane + r] = sym(lr + 4 * r, lc);
  }
  __syncthreads();
  if (ewave) {
    double t4[4];
    ld4(dscr + 4 * lane, t4);
    v4d d = {t4[0], t4[1], t4[2], t4[3]};
    inv16(x, 0, d);
#pragma unroll
    for (int r = 0; r < 4; ++r) einv0[(lr + 4 * r) * 16 + ks16(lc)] = d[r];
  }
  __syncthreads();
  for (int K = 0; 16 * K < np; ++K) {
    const int kb = 16 * K;
    const bool have_next = kb + 16 < np;
    double *einv = einv0 + 256 * (K & 1), *enext = einv0 + 256 * ((K + 1) & 1);
    // P: the pivot column panel PAN[i][.] = A[i][16 K + .] (masked rows / columns as zeros) from the tiles of column K and,
    // transposed, of row K
    if (!ewave) {
      for (int I = wave; I < Tn; I += TW) {
        if (I >= K) {
          const bool colact = kb + lc < np;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + lr + 4 * r;
            pan[(size_t)i * 16 + ks16(lc)] = colact ? sym(i, kb + lc) : 0.0;
          }
        } else {  // PAN[16 I + lc][c = lr + 4 r] = A[kb + lr + 4 r][16 I + lc]
          double *o = pan + (size_t)(16 * I + lc) * 16 + lr * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (kb + lr + 4 * r < np) ? A[(size_t)(kb + lr + 4 * r) * ld + 16 * I + lc] : 0.0;
        }
      }
    } else if (have_next) {  // current values of the next diagonal tile, for the look-ahead
#pragma unroll
      for (int r = 0; r < 4; ++r) dscr[4 * lane + r] = sym(kb + 16 + lr + 4 * r, kb + 16 + lc);
    }
    __syncthreads();
    v4d dn = {0.0, 0.0, 0.0, 0.0};
    if (!ewave) {
      for (int I = wave; I < Tn; I += TW) {  // W_I = PAN_I E_K
        double aP[4], eB[4];
        ld4(pan + (size_t)(16 * I + lc) * 16 + lr * 4, aP);
        ld4(einv + lc * 16 + lr * 4, eB);
        v4d w = {0.0, 0.0, 0.0, 0.0};
        w = mfma4(aP, eB, w);
#pragma unroll
        for (int r = 0; r < 4; ++r) wt[(size_t)(16 * I + lr + 4 * r) * 16 + ks16(lc)] = w[r];
      }
    } else if (have_next) {  // look-ahead: D'_{K+1} (the E-wave forms W_{K+1} itself)
      double aP[4], eB[4], aW[4], t4[4];
      ld4(pan + (size_t)(16 * (K + 1) + lc) * 16 + lr * 4, aP);  // also the B operand of the update (PAN_{K+1}^T)
      ld4(einv + lc * 16 + lr * 4, eB);
      v4d w1 = {0.0, 0.0, 0.0, 0.0};
      w1 = mfma4(aP, eB, w1);
#pragma unroll
      for (int r = 0; r < 4; ++r) es[(lr + 4 * r) * 16 + ks16(lc)] = w1[r];  // accumulator -> A-operand layout
      wave_lds_sync();
      ld4(es + lc * 16 + lr * 4, aW);
      ld4(dscr + 4 * lane, t4);
      dn = v4d{t4[0], t4[1], t4[2], t4[3]};
      dn = mfma4(aW, aP, dn);
    }
    __syncthreads();
    if (ewave) {
      if (have_next) {
        inv16(x, K + 1, dn);
#pragma unroll
        for (int r = 0; r < 4; ++r) enext[(lr + 4 * r) * 16 + ks16(lc)] = dn[r];
      }
    } else {
      for (int t = wave; t < ntiles; t += TW) {
        int ib = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
        while ((ib + 1) * (ib + 2) / 2 <= t) ++ib;
        while (ib * (ib + 1) / 2 > t) --ib;
        const int jb = t - ib * (ib + 1) / 2;
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ib + lr + 4 * r, j = 16 * jb + lc;
          acc[r] = sym(i, j);  // (diagonal tiles are kept fully symmetric in registers)
        }
        tile_step16(ib, jb, K, np, lc, lr, pan, wt, einv, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ib + lr + 4 * r, j = 16 * jb + lc;
          if (j <= i) A[(size_t)i * ld + j] = acc[r];
        }
      }
    }
    __syncthreads();
  }
}

// ---- software barrier among the waves that run the SLAM front end beside the simulator wave (k_step) ----
// A monotonic LDS counter: every participating wave adds one and waits (lane 0, s_sleep) until all have arrived.  The
// hardware barrier cannot be used there: the simulator wave does not take part.
struct SubBarrier {
  int *cnt;
  int nwaves, phase;
  __device__ __forceinline__ void sync(int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    ++phase;
    if (lane == 0) {
      __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nwaves * phase) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};

// What the simulator wave of k_step appended in this step, left in LDS (ksim::measure): factor M0 + r = (newest pose,
// landmark slot[r], bearing br[2 r], range br[2 r + 1]), new landmark L0 + r at lm[2 r], lm[2 r + 1].  br == null: not there.
struct SimBox {
  const double *br;
  const int *slot;
  const double *lm;
};

// The fast path: <= 42 poses (N <= 16 FT = 128), the whole problem in LDS.  Longer trajectories: arrow_body (k_slam_arrow.hip).
//
// The update is split in two so that the fused step kernel can run the first part BESIDE the simulator wave:
//   front  everything that does not depend on this step's measurements: relinearisation policy and theta staging,
//          linearisation of the factors that existed before the step, the pose blocks (the new pose's initial guess and
//          its odometry factor only depend on the commanded odometry and the previous estimate), the landmark sums over
//          the old factors.  In k_step it runs on 7 waves with software barriers (front<true>), in k_slam on all 8.
//   back   the new factors (linearisation, their terms appended to the sums in factor order - the same order of additions
//          as a single pass over all factors, so both kernels produce identical bits), landmark elimination, sweep, outputs.
struct SlamCtx {
  int inst, P, L, M;     // poses of the system; landmarks / factors known so far (front) or final (back)
  int Lb, Mb;            // carve bounds of the per-landmark / per-factor arrays
  int np, N, Tn, MW;
  int n_old_p, n_old_l, count;
  bool relin;
  // LDS
  double *thp, *odl, *thl, *lamb, *A, *rec;
  int *mstart, *bad, *lstart, *pstart;
  unsigned short *mp, *ml, *lfac, *obs, *pairlm;
  unsigned long long *lmask;

  __device__ __forceinline__ int AT(int i, int j) const { return i * (i + 1) / 2 + j; }

  // LDS carve from byte offset `off` of the dynamic shared memory: small arrays first, then the dense system, then - kBigLds -
  // the factor records and the observation table, which otherwise go to the HBM workspace
  __device__ __forceinline__ static size_t small_bytes(int P, int Lb, int Mb) {
    const size_t MW = (size_t)(P + 63) >> 6;
    return up8((size_t)P * 32) * 2 + up8((size_t)Lb * 16) + up8((size_t)Lb * 64) + up8((size_t)(P + 2) * 4) + up8((size_t)(Lb + 2) * 4) * 2 +
           up8((size_t)Mb * 2) * 3 + up8((size_t)(Mb / 2 + Lb + 2) * 2) + 8 + up8((size_t)Lb * MW * 8) + 32;
  }
  __device__ __forceinline__ static bool big_fits(size_t off, int lds_bytes, int P, int Lb, int Mb) {
    const size_t N = 16 * (((size_t)3 * P + 1 + 15) / 16);
    return off + small_bytes(P, Lb, Mb) + sweep_region_doubles(N) * 8 + (size_t)Mb * REC * 8 + up8((size_t)Lb * P * 2) <= (size_t)lds_bytes;
  }
  template <bool kBigLds>
  __device__ __forceinline__ void setup(const DrlgxState &S, unsigned char *smem_raw, size_t off, int lds_bytes, int inst_, int P_, int Lb_, int Mb_) {
    inst = inst_; P = P_; Lb = Lb_; Mb = Mb_;
    np = 3 * P;
    // padded to 16x16 MFMA tiles; row np holds the rhs (its column and all pad rows / columns stay zero).  Only the lower
    // triangle is ever addressed and it is stored packed (row i at i (i + 1) / 2)
    Tn = (np + 1 + 15) / 16; N = 16 * Tn;
    MW = (P + 63) >> 6;
    auto take = [&](size_t bytes) { unsigned char *q = smem_raw + off; off += up8(bytes); return q; };
    thp = reinterpret_cast<double *>(take((size_t)P * 4 * 8));
    odl = reinterpret_cast<double *>(take((size_t)P * 4 * 8));
    thl = reinterpret_cast<double *>(take((size_t)Lb * 2 * 8));
    lamb = reinterpret_cast<double *>(take((size_t)Lb * 8 * 8));
    mstart = reinterpret_cast<int *>(take((size_t)(P + 2) * 4));
    lstart = reinterpret_cast<int *>(take((size_t)(Lb + 2) * 4));
    pstart = reinterpret_cast<int *>(take((size_t)(Lb + 2) * 4));
    mp = reinterpret_cast<unsigned short *>(take((size_t)Mb * 2));
    ml = reinterpret_cast<unsigned short *>(take((size_t)Mb * 2));
    lfac = reinterpret_cast<unsigned short *>(take((size_t)Mb * 2));
    pairlm = reinterpret_cast<unsigned short *>(take((size_t)(Mb / 2 + Lb + 2) * 2));
    bad = reinterpret_cast<int *>(take(8));
    // poses observing each landmark as bit masks (MW 64-bit words): the per-landmark loops visit only those poses
    lmask = reinterpret_cast<unsigned long long *>(take((size_t)Lb * MW * 8));
    off = (off + 31) & ~(size_t)31;
    A = reinterpret_cast<double *>(smem_raw + off); off += sweep_region_doubles(N) * 8;  // (reused for the sweep panels)
    if constexpr (kBigLds) {
      rec = reinterpret_cast<double *>(smem_raw + off); off += (size_t)Mb * REC * 8;
      obs = reinterpret_cast<unsigned short *>(smem_raw + off);
    } else {
      double *wsd = S.slam_ws + (size_t)inst * S.slam_ws_stride;
      rec = wsd; wsd += (size_t)S.M_max * REC;
      obs = reinterpret_cast<unsigned short *>(wsd);
    }
    // Both are used through `flat` instructions whatever their placement: with pointers the compiler knows to be LDS the
    // record-walking phases measured SLOWER (landmark marginals 3.9 -> 6.6 us, the loads are scheduled as short-latency
    // ones), so the address space is hidden from it
    asm volatile("" : "+v"(rec));
    asm volatile("" : "+v"(obs));
  }

  // tables + the (expensive) linearisation of the factors [m0, m1), one thread each
  __device__ __forceinline__ void factor_tables(const DrlgxState &S, int m0, int m1, int t, int nt, const SimBox &box = SimBox{nullptr, nullptr, nullptr}) const {
    const int *meas_pose = S.meas_pose + (size_t)inst * S.M_max;
    const int *meas_lm = S.meas_lm + (size_t)inst * S.M_max;
    const double *meas_br = S.meas_br + (size_t)inst * S.M_max * 2;
    if (box.br) {  // this step's factors from the simulator wave's LDS: all of them observed from the newest pose
      const int p = P - 1;
      for (int m = m0 + t; m < m1; m += nt) {
        const int j = box.slot[m - m0];
        mp[m] = (unsigned short)p;
        ml[m] = (unsigned short)j;
        if (m == m0) mstart[p] = m;
        obs[j * P + p] = (unsigned short)(m + 1);
        atomicOr(&lmask[MW * j + (p >> 6)], 1ull << (p & 63));
        linearize_br(thp + 4 * p, thl + 2 * j, box.br[2 * (m - m0)], box.br[2 * (m - m0) + 1], rec + (size_t)REC * m);
      }
      return;
    }
    for (int m = m0 + t; m < m1; m += nt) {
      const int p = meas_pose[m], j = meas_lm[m];
      mp[m] = (unsigned short)p;
      ml[m] = (unsigned short)j;
      if (m == 0 || meas_pose[m - 1] != p) mstart[p] = m;
      obs[j * P + p] = (unsigned short)(m + 1);
      atomicOr(&lmask[MW * j + (p >> 6)], 1ull << (p & 63));
      linearize_br(thp + 4 * p, thl + 2 * j, meas_br[2 * m], meas_br[2 * m + 1], rec + (size_t)REC * m);
    }
  }

  // kSub: called by the threads 64 .. kThreads-1 (ft = tid - 64) while wave 0 simulates; Pf / L / M are the counts before
  // the step, the new pose (index Pf) comes from `odomP`.  Otherwise by all threads with the final counts (Pf = P).
  template <bool kSub>
  __device__ __forceinline__ void front(const DrlgxState &S, int tid, int Pf, int Lf, int Mf, int n_old_p_, int n_old_l_, int count_,
                                        bool refresh, const double *odom3, SubBarrier sb) {
    const drlgx_config &cfg = S.cfg;
    const int ft = kSub ? tid - 64 : tid, fn = kSub ? kThreads - 64 : kThreads, lane = tid & 63;
    auto bar = [&]() {
      if constexpr (kSub) sb.sync(lane);
      else __syncthreads();
    };
    L = Lf; M = Mf; n_old_p = n_old_p_; n_old_l = n_old_l_; count = count_;
    double *th_pose = S.th_pose + (size_t)inst * S.P_max * 4;
    double *d_pose = S.d_pose + (size_t)inst * S.P_max * 3;
    double *th_lm = S.th_lm + (size_t)inst * S.L_max * 2;
    double *d_lm = S.d_lm + (size_t)inst * S.L_max * 2;
    // ---- 1. relinearisation policy (gtsam ISAM2: relinearizeSkip 10, relinearizeThreshold 0.1);
    //         theta (+ folded delta) is staged in LDS ----
    relin = !refresh && (count % 10 == 0);
    for (int i = ft; i < Pf; i += fn) {
      Pose t{th_pose[4 * i], th_pose[4 * i + 1], th_pose[4 * i + 2], th_pose[4 * i + 3]};
      if (relin && i < n_old_p) {
        const double a = fabs(d_pose[3 * i]), b = fabs(d_pose[3 * i + 1]), c = fabs(d_pose[3 * i + 2]);
        if (fmax(a, fmax(b, c)) >= 0.1) {
          t = compose(t, make_pose(d_pose[3 * i], d_pose[3 * i + 1], d_pose[3 * i + 2]));
          th_pose[4 * i] = t.x; th_pose[4 * i + 1] = t.y; th_pose[4 * i + 2] = t.c; th_pose[4 * i + 3] = t.s;
        }
      }
      thp[4 * i] = t.x; thp[4 * i + 1] = t.y; thp[4 * i + 2] = t.c; thp[4 * i + 3] = t.s;
      if (i + 1 < Pf) {  // measured odometry between pose i and i + 1
        const double *oo = S.odo + ((size_t)inst * S.P_max + i) * 4;
        odl[4 * i] = oo[0]; odl[4 * i + 1] = oo[1]; odl[4 * i + 2] = oo[2]; odl[4 * i + 3] = oo[3];
      }
    }
    if constexpr (kSub) {
      // the pose this step appends: SLAM2D::addOdometry's initial guess = last estimate * odom (SLAM2D.cpp:70-89), the same
      // expressions as the simulator wave evaluates (k_sim.hip sim_step_body), which stores them to HBM
      if (ft == fn - 1) {
        const Pose odomP = make_pose(odom3[0], odom3[1], odom3[2]);
        const double *ep = S.est_pose + ((size_t)inst * S.P_max + (Pf - 1)) * 4;
        const Pose p2 = compose(Pose{ep[0], ep[1], ep[2], ep[3]}, odomP);
        thp[4 * Pf] = p2.x; thp[4 * Pf + 1] = p2.y; thp[4 * Pf + 2] = p2.c; thp[4 * Pf + 3] = p2.s;
        odl[4 * (Pf - 1)] = odomP.x; odl[4 * (Pf - 1) + 1] = odomP.y; odl[4 * (Pf - 1) + 2] = odomP.c; odl[4 * (Pf - 1) + 3] = odomP.s;
      }
    }
    for (int j = ft; j < Lf; j += fn) {
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
    // ---- 2. clear the system; factor tables (factors are appended in pose order: contiguous ranges) ----
    {
      double2 *A2 = reinterpret_cast<double2 *>(A);
      const int n2 = (int)((size_t)N * (N + 1) / 2 / 2);  // (N is a multiple of 16: even)
      for (int e = ft; e < n2; e += fn) A2[e] = make_double2(0.0, 0.0);
    }
    for (int e = ft; e < Lb * P; e += fn) obs[e] = 0;
    for (int e = ft; e < MW * Lb; e += fn) lmask[e] = 0ull;
    for (int e = ft; e <= P; e += fn) mstart[e] = 0x7fffffff;
    if (ft == 0) {
      bad[0] = 0;
      bad[1] = 0;  // (pairs of tile (0, 0) subtracted so far: back(), Schur phase)
    }
    bar();
    if (S.prof && blockIdx.x == S.prof_block && ft == 0) S.prof[33] = wall_clock64();
    factor_tables(S, 0, Mf, ft, fn);
    bar();
    if (S.prof && blockIdx.x == S.prof_block && ft == 0) S.prof[34] = wall_clock64();
    // poses without factors get the empty range [next pose's start, same): the first assigned start at or after p, i.e.
    // the suffix minimum of the raw starts (they increase with the pose); one wave, top chunk first
    if (ft < 64) {
      int carry = 0x7fffffff;
      for (int base = (P >> 6) << 6; base >= 0; base -= 64) {
        const int q = base + ft;
        int v = q < P ? mstart[q] : (q == P ? Mf : 0x7fffffff);  // (the end of the list = the start of this step's factors)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int w = __shfl_down(v, o);
          if (ft + o < 64) v = min(v, w);
        }
        v = min(v, carry);
        if (q <= P) mstart[q] = v;
        carry = __shfl(v, 0);
      }
    }
    bar();
    if (S.prof && blockIdx.x == S.prof_block && ft == 0) S.prof[35] = wall_clock64();
    // ---- 3. block assembly.  first waves: one thread per landmark; following waves: one thread per pose ----
    const double wb = S.w_bear, wr = S.w_range;
    for (int j = ft; j < Lf; j += fn) {
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
      lb[0] = a; lb[1] = b; lb[2] = d; lb[3] = g0; lb[4] = g1;  // (sums so far; back() appends this step's terms)
    }
    // Pose blocks (pose_block in pieces).  The LAST wave: lane i linearises odometry factor i once - for both of its keys -
    // and lane P - 1, which has none, the prior; the contributions to the second key travel through LDS (`c2buf`, 9 doubles
    // per pose; slot 0 = the prior).  The threads below it, eight per pose from the top down (the landmark loop above
    // occupies the first ones): the own bearing-range factors, every eighth factor per lane, summed over the eight lanes
    // (back() adds the newest pose's the same way when this front end ran before they existed: both ways round alike).
    double *c2buf = A + (size_t)N * (N + 1) / 2;  // (the sweep region is larger than the packed triangle: >= 18 P doubles)
    double *ownsum = c2buf + 9 * P;
    const int la = ft - (fn - 64);
    double B6[6] = {0, 0, 0, 0, 0, 0}, g3[3] = {0, 0, 0};
    if (la >= 0) {
      if (la + 1 < P) {
        double C2[6], g2[3], O[9];
        odo_factor(S, thp, odl, la, B6, g3, C2, g2, O);
        double *o2 = c2buf + 9 * (la + 1);
        for (int q = 0; q < 6; ++q) o2[q] = C2[q];
        for (int r = 0; r < 3; ++r) o2[6 + r] = g2[r];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) A[AT((3 * (la + 1) + r), 3 * la + c)] = O[r * 3 + c];
      } else if (la == P - 1) {
        double PB[6], pg[3];
        prior_factor(S, inst, thp, PB, pg);
        for (int q = 0; q < 6; ++q) c2buf[q] = PB[q];
        for (int r = 0; r < 3; ++r) c2buf[6 + r] = pg[r];
      }
    } else {
      const int idx = fn - 65 - ft, grp = idx >> 3, part = idx & 7, ngrp = (fn - 64) >> 3;
      for (int i0 = 0; i0 < P; i0 += ngrp) {  // (uniform trip count: the lane sums run on whole waves)
        const int i = i0 + grp;
        double s6[6] = {0, 0, 0, 0, 0, 0}, sg[3] = {0, 0, 0};
        if (i < P)
          for (int m = mstart[i] + part; m < mstart[i + 1]; m += 8) own_factor_add(rec + (size_t)REC * m, wb, wr, s6, sg);
        for (int q = 0; q < 6; ++q) s6[q] = sum8_lane7(s6[q]);
        for (int r = 0; r < 3; ++r) sg[r] = sum8_lane7(sg[r]);
        if (i < P && (lane & 7) == 7) {
          double *o = ownsum + 9 * i;
          for (int q = 0; q < 6; ++q) o[q] = s6[q];
          for (int r = 0; r < 3; ++r) o[6 + r] = sg[r];
        }
      }
    }
    bar();
    if (la >= 0 && la < P) {
      // prior / second key, first key, own factors: the order pose_block adds them in
      const double *c2 = c2buf + 9 * la, *own = ownsum + 9 * la;
      double B[6], g[3];
      for (int q = 0; q < 6; ++q) B[q] = (c2[q] + B6[q]) + own[q];
      for (int r = 0; r < 3; ++r) g[r] = (c2[6 + r] + g3[r]) + own[6 + r];
      for (int r = 0, q = 0; r < 3; ++r) {
        for (int c = 0; c <= r; ++c, ++q) A[AT((3 * la + r), 3 * la + c)] = B[q];
        A[AT(np, 3 * la + r)] = -g[r];  // rhs lives in the augmented row
      }
    }
    if (S.prof && blockIdx.x == S.prof_block) {  // (dev aid: end of the front end, first thread and per wave)
      if (ft == 0) S.prof[14] = wall_clock64();
      if (lane == 0) S.prof[24 + (tid >> 6)] = wall_clock64();
    }
  }

  // everything after the simulator: all kThreads threads, hardware barriers.  Lfin / Mfin: the final counts (>= the front's).
  // hand: LDS (or null) that receives what the map stage of k_step reads next - est_pose [P][4] and, behind it at
  // hand + 4 P_max, pose_info [P][6] - so that it does not fetch them back from HBM; the landmark estimates are left in
  // `thl` for the same reason (the linearisation points are dead by then)
  template <int FT>
  // hand_cap: the pose capacity of those tables (the launch's pose bound, LaunchSel::cap)
  __device__ __forceinline__ void back(const DrlgxState &S, int tid, int Lfin, int Mfin, bool full, bool refresh, double *hand = nullptr,
                                       const SimBox &box = SimBox{nullptr, nullptr, nullptr}, int hand_cap = 0) {
    int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
    double *d_pose = S.d_pose + (size_t)inst * S.P_max * 3;
    double *d_lm = S.d_lm + (size_t)inst * S.L_max * 2;
    const drlgx_config &cfg = S.cfg;
    const double wb = S.w_bear, wr = S.w_range;
    double *th_lm = S.th_lm + (size_t)inst * S.L_max * 2;
    const int L0 = L, M0 = M;
    L = Lfin; M = Mfin;
    // ---- this step's landmarks and factors (all of them observed from the newest pose) ----
    for (int j = L0 + tid; j < L; j += kThreads) {
      thl[2 * j] = box.br ? box.lm[2 * (j - L0)] : th_lm[2 * j];
      thl[2 * j + 1] = box.br ? box.lm[2 * (j - L0) + 1] : th_lm[2 * j + 1];
      double *lb = lamb + 8 * j;
      lb[0] = lb[1] = lb[2] = lb[3] = lb[4] = 0.0;
    }
    if (tid == 0) mstart[P] = M;
    __syncthreads();
    factor_tables(S, M0, M, tid, kThreads, box);
    __syncthreads();
    DRLGX_PROF(S, 1);
    // ---- landmark blocks: this step's term (at most one per landmark) closes the sum, then Lambda_jj^-1 and eta_j;
    //      CSR offsets of the per-landmark factor lists (one wave); the newest pose's own factors close its block ----
    for (int j = tid; j < L; j += kThreads) {
      double *lb = lamb + 8 * j;
      double a = lb[0], b = lb[1], d = lb[2], g0 = lb[3], g1 = lb[4];
      const int m1 = M > M0 ? obs[j * P + (P - 1)] : 0;
      if (m1 > M0) {
        const double *r = rec + (size_t)REC * (m1 - 1);
        a += r[6] * wb * r[6] + r[8] * wr * r[8];
        b += r[6] * wb * r[7] + r[8] * wr * r[9];
        d += r[7] * wb * r[7] + r[9] * wr * r[9];
        g0 += r[6] * wb * r[10] + r[8] * wr * r[11];
        g1 += r[7] * wb * r[10] + r[9] * wr * r[11];
      }
      const double id = 1.0 / (a * d - b * b);
      lb[0] = a; lb[1] = b; lb[2] = d;
      lb[3] = d * id; lb[4] = -b * id; lb[5] = a * id;  // Lambda_jj^-1
      lb[6] = -g0; lb[7] = -g1;                           // eta_j
    }
    if (tid >= kThreads - 64) {  // exclusive scan of the observation counts -> lstart[0 .. L]
      const int ln = tid - (kThreads - 64);
      unsigned carry = 0;
      for (int base = 0; base < L; base += 64) {
        const int j = base + ln;
        int k = 0;
        if (j < L)
          for (int w = 0; w < MW; ++w) k += __popcll(lmask[MW * j + w]);
        // (two scans in one: observations in the low half, pairs of observations - ceil(k / 2) - in the high half)
        const int k2 = (k + 1) >> 1;
        unsigned v = (unsigned)k | ((unsigned)k2 << 16);
        // inclusive scan over the wave on DPP: row shifts inside the 16-lane rows, then the two row broadcasts
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1, 3
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2, 3
        if (j < L) {
          lstart[j] = (int)((carry & 0xffffu) + (v & 0xffffu)) - k;
          pstart[j] = (int)((carry >> 16) + (v >> 16)) - k2;
        }
        carry += (unsigned)__builtin_amdgcn_readlane((int)v, 63);
      }
      if (ln == 0) {
        lstart[L] = (int)(carry & 0xffffu);
        pstart[L] = (int)(carry >> 16);
      }
    }
    if (M > M0 && (tid >> 3) == kThreads / 16) {
      // the own factors of the newest pose, appended after the front ran: eight lanes, every eighth factor each, like the
      // front end sums the own factors of every pose (so that (front + this) == the front alone when it runs after them)
      const int i = P - 1, part = 7 - (tid & 7);
      double s6[6] = {0, 0, 0, 0, 0, 0}, sg[3] = {0, 0, 0};
      for (int m = M0 + part; m < M; m += 8) own_factor_add(rec + (size_t)REC * m, wb, wr, s6, sg);
      for (int q = 0; q < 6; ++q) s6[q] = sum8_lane7(s6[q]);
      for (int r = 0; r < 3; ++r) sg[r] = sum8_lane7(sg[r]);
      if ((tid & 7) == 7) {
        for (int r = 0, q = 0; r < 3; ++r) {
          for (int c = 0; c <= r; ++c, ++q) A[AT(3 * i + r, 3 * i + c)] += s6[q];
          A[AT(np, 3 * i + r)] = -(-A[AT(np, 3 * i + r)] + sg[r]);
        }
      }
    }
    __syncthreads();
    DRLGX_PROF(S, 2);
    // ---- 4. landmark elimination: rec[0..5] <- G_m = Lambda_pl Lambda_ll^-1 (3x2), rec[6..11] <- H_m = G_m Lambda_jj
    //         (= Lambda_pl; the Jacobian of the landmark and the residual are not needed any more);
    //         per-landmark factor lists lfac[lstart[j] ..] in pose order, and the pair table of phase 6 ----
    for (int m = tid; m < M; m += kThreads) {
      double *l = rec + (size_t)REC * m;
      const double *lb = lamb + 8 * ml[m];
      double g[6], h[6];
      for (int r = 0; r < 3; ++r) {
        const double b0 = l[r] * wb * l[6] + l[3 + r] * wr * l[8];
        const double b1 = l[r] * wb * l[7] + l[3 + r] * wr * l[9];
        g[r * 2 + 0] = b0 * lb[3] + b1 * lb[4];
        g[r * 2 + 1] = b0 * lb[4] + b1 * lb[5];
      }
      for (int r = 0; r < 3; ++r) {
        h[r * 2 + 0] = g[r * 2] * lb[0] + g[r * 2 + 1] * lb[1];
        h[r * 2 + 1] = g[r * 2] * lb[1] + g[r * 2 + 1] * lb[2];
      }
      for (int k = 0; k < 6; ++k) l[k] = g[k];
      for (int k = 0; k < 6; ++k) l[6 + k] = h[k];
      // the factor's place in its landmark's list = the rank of its pose among the observers (no list walk)
      const int j = ml[m], p = mp[m];
      int rank = 0;
      for (int w = 0; w < (p >> 6); ++w) rank += __popcll(lmask[MW * j + w]);
      rank += __popcll(lmask[MW * j + (p >> 6)] & ((1ull << (p & 63)) - 1ull));
      lfac[lstart[j] + rank] = (unsigned short)m;
      if (rank < pstart[j + 1] - pstart[j]) pairlm[pstart[j] + rank] = (unsigned short)j;
    }

    __syncthreads();
    DRLGX_PROF(S, 3);
    //      Schur complement: S_pq -= sum_j G_m Lambda_jj G_mq^T   (Lambda_pl = G Lambda_jj = H)
    // While the other seven waves do that, the wave that inverts the diagonal tiles in the sweep (an idle tile row's wave
    // when the system has fewer than FT tile rows: <= 37 poses) already inverts the FIRST one: tile (0, 0) is complete as
    // soon as the pairs of the poses 0..5 are subtracted - the first 21 pairs, counted in an LDS flag by their threads -
    // and its inversion (2.2 us) used to run after this phase with every other wave waiting at a barrier.
    const bool pre_e0 = Tn < FT;
    const int ewave_first = 64 * (FT / 2);  // first thread of that wave (sweep_packed_fast: wave FT / 2 owns tile row FT - 1)
    const bool is_ewave = pre_e0 && tid >= ewave_first && tid < ewave_first + 64;
    v4d e0 = {0.0, 0.0, 0.0, 0.0};
    if (is_ewave) {
      const int p6 = min(P, 6), need = p6 * (p6 + 1) / 2;
      while (__hip_atomic_load(bad + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      const int lane = tid & 63, lc = lane & 15, lr = lane >> 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = lr + 4 * r;
        e0[r] = A[AT(max(i, lc), min(i, lc))];  // (N >= 16: the whole tile is inside the triangle)
      }
      const SweepCtx x{FT - 1, lane, lc, lr, np, N, false, true, bad, nullptr};
      inv16_blk(x, min(16, np), e0);
    } else {
      const int npairs = P * (P + 1) / 2;
      const int sidx = (pre_e0 && tid >= ewave_first) ? tid - 64 : tid, sn = pre_e0 ? kThreads - 64 : kThreads;
      for (int e = sidx; e < npairs; e += sn) {
        int p = (int)((sqrtf(8.0f * e + 1.0f) - 1.0f) * 0.5f);
        while ((p + 1) * (p + 2) / 2 <= e) ++p;
        while (p * (p + 1) / 2 > e) --p;
        const int q = e - p * (p + 1) / 2;
        // Row-major pairs: the lanes of a wave share p (one factor list, one trip count).  Measured and dropped - none
        // faster than this plain loop (5.2 us; 9.1 us for the instances with the most factors, of which 2.1 / 4.3 us are
        // the look-ups and the rest the terms): four look-ups per round issued together; look-ups software-pipelined one
        // factor ahead; the landmarks common to both poses from per-pose bit masks; diagonal-major pair order (lanes of
        // similar hit counts, but different factor lists: 6.2 / 9.8 us); two lanes per pair on the even / odd factors with a
        // DPP sum (three rounds of half the length instead of 1.3 of the full one: 6.9 / 9.9 us - the per-pair overhead,
        // index decode and the nine read-modify-writes, is worth ~2.5 loop iterations).
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        bool any = false;
        for (int m = mstart[p]; m < mstart[p + 1]; ++m) {
          const int mq1 = obs[ml[m] * P + q];
          if (!mq1) continue;
          any = true;
          const double *h = rec + (size_t)REC * m + 6, *gq = rec + (size_t)REC * (mq1 - 1);
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) acc[r * 3 + c] += h[r * 2] * gq[c * 2] + h[r * 2 + 1] * gq[c * 2 + 1];
        }
        if (any)
          for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
              if (p == q && c > r) continue;
              A[AT((3 * p + r), 3 * q + c)] -= acc[r * 3 + c];
            }
        if (pre_e0 && p < 6) {  // a pair of tile (0, 0): done (release: the subtractions above are visible before the count)
          __threadfence_block();
          atomicAdd(bad + 1, 1);
        }
      }
      for (int p = (tid + kThreads - 64 * (FT / 2 + 1)) % kThreads; p < P; p += kThreads) {  // rhs_p -= sum_m G_m eta_j (idle waves)
        double s0 = 0, s1 = 0, s2 = 0;
        for (int m = mstart[p]; m < mstart[p + 1]; ++m) {
          const double *g = rec + (size_t)REC * m, *lb = lamb + 8 * ml[m];
          s0 += g[0] * lb[6] + g[1] * lb[7];
          s1 += g[2] * lb[6] + g[3] * lb[7];
          s2 += g[4] * lb[6] + g[5] * lb[7];
        }
        A[AT(np, 3 * p + 0)] -= s0;
        A[AT(np, 3 * p + 1)] -= s1;
        A[AT(np, 3 * p + 2)] -= s2;
      }
    }
    __syncthreads();
    DRLGX_PROF(S, 4);
    // ---- 5. sweep: one tile row per wave (sweep_packed_fast); 9 - 10 tile rows: the tiles dealt over seven waves ----
#ifdef DRLGX_SWEEP_REGTILES  // (A/B: nine and ten tile rows as tiles dealt over seven waves, three barriers per block step)
    if (Tn <= FT) sweep_packed_fast<FT>(S, A, np, N, Tn, bad, tid, pre_e0, e0);
    else sweep_regtiles<true, 8>(A, A, np, N, Tn, Tn * (Tn + 1) / 2, bad, tid);
#else
    sweep_packed_fast<FT>(S, A, np, N, Tn, bad, tid, pre_e0, e0);  // (nine / ten rows: two light rows share a wave)
#endif
    __syncthreads();
    DRLGX_PROF(S, 5);
    for (int k = tid; k < np; k += kThreads) d_pose[k] = A[AT(np, k)];
    // ---- 6. landmark marginals: Sigma_jj = Lambda_jj^-1 + sum_{a, b} G_a^T Sigma[p_a][p_b] G_b over the landmark's factor list
    //         (ascending poses).  By symmetry only b <= a is evaluated: factor a gets
    //             rec[6..9] <- Y_a + X_a + X_a^T,   Y_a = G_a^T Sigma_aa G_a,   X_a = sum_{b < a} G_a^T Sigma_ab G_b,
    //         whose sum over the list is the full double sum.  One work item = the list entries a and k-1-a of a landmark
    //         (a + (k-1-a) = k-1 block products whatever a: balanced), split over S6 adjacent lanes and combined by a
    //         butterfly (a fixed tree: deterministic).  The longest list sets the latency of this phase: (k-1) / S6 rounds. ----
    if (full) {
      const int NP = pstart[L];
      const int sh6 = 4 * NP <= kThreads ? 2 : 2 * NP <= kThreads ? 1 : 0, S6 = 1 << sh6;
      const int per_pass = kThreads >> sh6;
      for (int pid0 = 0; pid0 < NP; pid0 += per_pass) {
        const int pid = pid0 + (tid >> sh6), s6 = tid & (S6 - 1);
        const bool work = pid < NP;
        double X[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};  // X_a of the two entries (row major 2 x 2)
        int ma[2] = {0, 0};
        bool two = false;
        if (work) {
          const int j = pairlm[pid], t0 = lstart[j], k = lstart[j + 1] - t0;
          const int a0 = pid - pstart[j], a1 = k - 1 - a0;  // a0 <= a1
          two = a1 > a0;
          ma[0] = lfac[t0 + a0];
          ma[1] = lfac[t0 + a1];
          const int pa0 = mp[ma[0]], pa1 = mp[ma[1]];
          double W[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
          // the earlier entries b of the list: Sigma[p_a][p_b] = -(swept block), stored as rows of the later pose p_a
          for (int b = s6; b < a1; b += S6) {
            const int mb = lfac[t0 + b], pb = mp[mb];
            const double *gb = rec + (size_t)REC * mb;
            const double g0 = gb[0], g1 = gb[1], g2 = gb[2], g3 = gb[3], g4 = gb[4], g5 = gb[5];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              if (e == 0 && (b >= a0 || !two)) continue;
              const int pa = e ? pa1 : pa0;
#pragma unroll
              for (int r = 0; r < 3; ++r) {
                const int row = AT(3 * pa + r, 3 * pb);
                const double t0v = A[row], t1v = A[row + 1], t2v = A[row + 2];
                W[e][r * 2] -= t0v * g0 + t1v * g2 + t2v * g4;
                W[e][r * 2 + 1] -= t0v * g1 + t1v * g3 + t2v * g5;
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const double *ga = rec + (size_t)REC * ma[e];
            X[e][0] = ga[0] * W[e][0] + ga[2] * W[e][2] + ga[4] * W[e][4];
            X[e][1] = ga[0] * W[e][1] + ga[2] * W[e][3] + ga[4] * W[e][5];
            X[e][2] = ga[1] * W[e][0] + ga[3] * W[e][2] + ga[5] * W[e][4];
            X[e][3] = ga[1] * W[e][1] + ga[3] * W[e][3] + ga[5] * W[e][5];
          }
        }
        // (butterfly over the S6 <= 4 adjacent lanes of an item: DPP quad permutes, no trip through the LDS crossbar)
        if (S6 >= 4)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) X[e][q4] += dpp_quad_f64<0x4E>(X[e][q4]);  // lane ^ 2
        if (S6 >= 2)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) X[e][q4] += dpp_quad_f64<0xB1>(X[e][q4]);  // lane ^ 1
        if (work && s6 == 0) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (e == 0 && !two) continue;  // (a0 == a1: the middle entry of an odd list, handled as e = 1)
            double *g = rec + (size_t)REC * ma[e];
            const int pa = mp[ma[e]];
            // Y = G^T Sigma_aa G with the symmetric diagonal block
            const double s00 = -A[AT(3 * pa, 3 * pa)], s10 = -A[AT(3 * pa + 1, 3 * pa)], s11 = -A[AT(3 * pa + 1, 3 * pa + 1)];
            const double s20 = -A[AT(3 * pa + 2, 3 * pa)], s21 = -A[AT(3 * pa + 2, 3 * pa + 1)], s22 = -A[AT(3 * pa + 2, 3 * pa + 2)];
            const double d00 = s00 * g[0] + s10 * g[2] + s20 * g[4], d01 = s00 * g[1] + s10 * g[3] + s20 * g[5];
            const double d10 = s10 * g[0] + s11 * g[2] + s21 * g[4], d11 = s10 * g[1] + s11 * g[3] + s21 * g[5];
            const double d20 = s20 * g[0] + s21 * g[2] + s22 * g[4], d21 = s20 * g[1] + s21 * g[3] + s22 * g[5];
            const double y00 = g[0] * d00 + g[2] * d10 + g[4] * d20, y01 = g[0] * d01 + g[2] * d11 + g[4] * d21;
            const double y10 = g[1] * d00 + g[3] * d10 + g[5] * d20, y11 = g[1] * d01 + g[3] * d11 + g[5] * d21;
            g[6] = y00 + 2.0 * X[e][0];
            g[7] = y01 + (X[e][1] + X[e][2]);
            g[8] = y10 + (X[e][1] + X[e][2]);
            g[9] = y11 + 2.0 * X[e][3];
          }
        }
      }
    }
    __syncthreads();
    DRLGX_PROF(S, 6);
    double *est_lm = S.est_lm + (size_t)inst * S.L_max * 2;
    double *lm_info = S.lm_info + (size_t)inst * S.L_max * 3;
    double *lm_tr = S.lm_tr + (size_t)inst * S.L_max;
    // (a landmark's list is split over S7 adjacent lanes - the first half of the workgroup; the poses below take the second -
    // and combined by a butterfly)
    const int sh7 = 4 * L <= kThreads / 2 ? 2 : 2 * L <= kThreads / 2 ? 1 : 0, S7 = 1 << sh7;
    for (int j0 = 0; j0 < L; j0 += kThreads >> sh7) {
      const int j = j0 + (tid >> sh7), s7 = tid & (S7 - 1);
      const bool lwork = j < L;
      double c00 = 0, c01 = 0, c10 = 0, c11 = 0, dx = 0, dy = 0;
      if (lwork) {
        for (int t = lstart[j] + s7; t < lstart[j + 1]; t += S7) {
          const int mq = lfac[t], p = mp[mq];
          const double *g = rec + (size_t)REC * mq;
          c00 += g[6]; c01 += g[7]; c10 += g[8]; c11 += g[9];
          const double dp0 = A[AT(np, 3 * p)], dp1 = A[AT(np, 3 * p + 1)], dp2 = A[AT(np, 3 * p + 2)];
          dx -= g[0] * dp0 + g[2] * dp1 + g[4] * dp2;
          dy -= g[1] * dp0 + g[3] * dp1 + g[5] * dp2;
        }
      }
      if (S7 >= 4) {
        c00 += dpp_quad_f64<0x4E>(c00); c01 += dpp_quad_f64<0x4E>(c01); c10 += dpp_quad_f64<0x4E>(c10); c11 += dpp_quad_f64<0x4E>(c11);
        dx += dpp_quad_f64<0x4E>(dx); dy += dpp_quad_f64<0x4E>(dy);
      }
      if (S7 >= 2) {
        c00 += dpp_quad_f64<0xB1>(c00); c01 += dpp_quad_f64<0xB1>(c01); c10 += dpp_quad_f64<0xB1>(c10); c11 += dpp_quad_f64<0xB1>(c11);
        dx += dpp_quad_f64<0xB1>(dx); dy += dpp_quad_f64<0xB1>(dy);
      }
      if (!lwork || s7 != 0) continue;
      const double *lb = lamb + 8 * j;
      c00 += lb[3]; c01 += lb[4]; c10 += lb[4]; c11 += lb[5];
      // delta_j = Lambda^-1 eta_j - sum_m G_m^T delta_p
      dx += lb[3] * lb[6] + lb[4] * lb[7];
      dy += lb[4] * lb[6] + lb[5] * lb[7];
      d_lm[2 * j] = dx;
      d_lm[2 * j + 1] = dy;
      est_lm[2 * j] = thl[2 * j] + dx;
      est_lm[2 * j + 1] = thl[2 * j + 1] + dy;
      if (hand) {
        thl[2 * j] = thl[2 * j] + dx;
        thl[2 * j + 1] = thl[2 * j + 1] + dy;
      }
      if (!full) continue;
      const double cs = 0.5 * (c01 + c10);
      lm_tr[j] = c00 + c11;
      const double id = 1.0 / (c00 * c11 - cs * cs);  // marginalCovariance(l).inverse() (SLAM2D.cpp:417)
      lm_info[3 * j] = c11 * id;
      lm_info[3 * j + 1] = -cs * id;
      lm_info[3 * j + 2] = c00 * id;
    }
    // ---- 7. pose estimates, information = inverse(covariance) by LLT (SLAM2D.cpp:395-408) ----
    double *est_pose = S.est_pose + (size_t)inst * S.P_max * 4;
    double *pose_info = S.pose_info + (size_t)inst * S.P_max * 6;
    double *pose_tr = S.pose_tr + (size_t)inst * S.P_max;
    for (int i = (tid + kThreads / 2) % kThreads; i < P; i += kThreads) {
      const int k0 = 3 * i;
      const Pose t{thp[4 * i], thp[4 * i + 1], thp[4 * i + 2], thp[4 * i + 3]};
      const Pose e = compose(t, make_pose(A[AT(np, k0)], A[AT(np, k0 + 1)], A[AT(np, k0 + 2)]));
      est_pose[4 * i] = e.x; est_pose[4 * i + 1] = e.y; est_pose[4 * i + 2] = e.c; est_pose[4 * i + 3] = e.s;
      if (hand) {
        hand[4 * i] = e.x; hand[4 * i + 1] = e.y; hand[4 * i + 2] = e.c; hand[4 * i + 3] = e.s;
      }
      if (!full) continue;
      const double c00 = -A[AT(k0, k0)], c10 = -A[AT((k0 + 1), k0)], c11 = -A[AT((k0 + 1), k0 + 1)];
      const double c20 = -A[AT((k0 + 2), k0)], c21 = -A[AT((k0 + 2), k0 + 1)], c22 = -A[AT((k0 + 2), k0 + 2)];
      pose_tr[i] = c00 + c11 + c22;
      double info[6];
      inv3_sym_fast(c00, c10, c20, c11, c21, c22, info);
      for (int k = 0; k < 6; ++k) pose_info[6 * i + k] = info[k];
      if (hand)
        for (int k = 0; k < 6; ++k) hand[4 * hand_cap + 6 * i + k] = info[k];
    }
    DRLGX_PROF(S, 7);
    if (tid == 0) {
      if (!refresh) {
        cnt[C_ISAM] = count;
        cnt[C_NEWP] = P;
        cnt[C_NEWL] = L;
      }
      if (bad[0]) atomicMin(S.status, DRLGX_E_NUMERIC);
    }
  }
};

#include "k_inc.hip"

// The incremental update as a stage (k_slam / k_slam_arrow / k_step_arrow, after the simulator): true when it served the
// instance - the caller then skips its solver.  Every thread of the workgroup calls it.
template <int kSNT>
__device__ __forceinline__ bool inc_stage(const DrlgxState &S, const LaunchSel &sel, int lds_bytes, size_t smem_off) {
  const int tid = drlgx_tid(), bi = drlgx_bid();
  if (!S.jc || !sel.on(bi)) return false;
  const int inst = sel.base + bi;
  const int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  if (cnt[C_FLAG]) return false;  // (a rejected move appended nothing)
  const int P = cnt[C_P], L = cnt[C_L], M = cnt[C_M];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (!inc_precheck(S, inst, P, tid, reinterpret_cast<int *>(smem_raw + smem_off))) return false;
  IncCtx x;
  bool lds_panel = false;
  if (!inc_plan(S, inst, P, lds_bytes, smem_off, x, lds_panel, sel.cap(S.P_max))) return false;
  const SimBox nobox{nullptr, nullptr, nullptr};
  bool done;
  if (lds_panel) {
    inc_pre<true, false>(S, x, tid, nullptr, SubBarrier{nullptr, 0, 0});
    __syncthreads();
    done = inc_post<true, kSNT>(S, x, L, M, nobox, tid);
  } else {
    inc_pre<false, false>(S, x, tid, nullptr, SubBarrier{nullptr, 0, 0});
    __syncthreads();
    done = inc_post<false, kSNT>(S, x, L, M, nobox, tid);
  }
  if (!done) __syncthreads();  // (the full solve reuses the LDS)
  return done;
}

// The SLAM stage after the simulator.  `pre` (have_pre): the context whose front() already ran beside the simulator (k_step)
// for the counts before the step (records in LDS).  smem_off: first byte of the dynamic LDS the stage may use.
template <int FT>
__device__ __forceinline__ void slam_finish(const DrlgxState &S, const LaunchSel &sel, int lds_bytes, size_t smem_off, const SlamCtx &pre, bool have_pre,
                                            const int *mail = nullptr, double *hand = nullptr, const double **lm_out = nullptr,
                                            SimBox box = SimBox{nullptr, nullptr, nullptr}, int hand_cap = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = drlgx_tid();
  const int bi = drlgx_bid();
  if (!sel.on(bi)) return;
  const int inst = sel.base + bi;
  int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  // `full`: the block marginals are wanted.  Look-ahead rollouts only need them at their last action (the virtual map is
  // rebuilt there and nowhere else): the steps before it solve for the estimates only.  If that last action was rejected
  // (nothing was appended) the marginals of the unchanged system are recomputed without counting as an update.
  const bool full = sel.map_on(bi);
  // (mail: the counts after the step as the simulator wave left them in LDS - no round trip to HBM; -1: it appended nothing)
  const bool mailed = mail && mail[0] >= 0;
  const bool refresh = mailed ? false : cnt[C_FLAG] != 0;
  if (refresh && !(sel.map_last_only && sel.n_act && full)) return;
  const int P = mailed ? mail[0] : cnt[C_P], L = mailed ? mail[1] : cnt[C_L], M = mailed ? mail[2] : cnt[C_M];
  if ((3 * P + 1 + 15) / 16 > kDenseTiles) {
    // more poses than this kernel was launched for (the host's bound was wrong): flag it, touch nothing
    if (tid == 0) atomicMin(S.status, DRLGX_E_CAPACITY);
    return;
  }
  DRLGX_PROF(S, 0);
  SlamCtx c;
  const bool from_pre = have_pre && !refresh && pre.P == P && L <= pre.Lb && M <= pre.Mb;
  if (from_pre) {
    c = pre;
  } else {
    // stand-alone kernel, a rejected move, or more new landmarks / factors than the front reserved room for: everything now
    const int n_old_p = cnt[C_NEWP], n_old_l = cnt[C_NEWL], count = cnt[C_ISAM] + (refresh ? 0 : 1);
    if (SlamCtx::big_fits(smem_off, lds_bytes, P, L, M)) c.setup<true>(S, smem_raw, smem_off, lds_bytes, inst, P, L, M);
    else c.setup<false>(S, smem_raw, smem_off, lds_bytes, inst, P, L, M);
    c.front<false>(S, tid, P, L, M, n_old_p, n_old_l, count, refresh, nullptr, SubBarrier{nullptr, 0, 0});
    __syncthreads();
  }
  c.back<FT>(S, tid, L, M, full, refresh, hand, (from_pre && mailed) ? box : SimBox{nullptr, nullptr, nullptr}, hand_cap);
  if (lm_out) *lm_out = c.thl;
  if (S.jc && !refresh) {  // the covariance panel the incremental updates continue from (k_inc.hip)
    __syncthreads();
    panel_from_dense(S, c, tid);
  }
}

template <int FT>
__device__ __forceinline__ void slam_body(const DrlgxState &S, const LaunchSel &sel, int lds_bytes) {
  if (inc_stage<0>(S, sel, lds_bytes, 0)) return;
  SlamCtx none;
  slam_finish<FT>(S, sel, lds_bytes, 0, none, false);
}

template <int FT>
__global__ __launch_bounds__(kThreads) void k_slam(DRLGX_KS_PARAM, LaunchSel sel, int lds_bytes) {
  const DrlgxState &S = DRLGX_KS_REF;
  slam_body<FT>(S, sel, lds_bytes);
}

constexpr int kLdsBudget = 160 * 1024;
constexpr int kFastTiles = 8;       // fast path: N = 128 (<= 42 poses), system + panels in LDS
constexpr int kFastTilesArrow = 8;  // arrow path: landmark system of <= 63 landmarks (N <= 128) packed in LDS
static_assert(kFastTilesArrow == 8, "inc_plan (k_inc.hip) spells the reach of k_step_arrow out as 16 * 8");
constexpr int kArrowRegTiles = 20;  // ... beyond: up to 20 register tiles per wave (N <= 256, <= 127 landmarks)

#include "k_slam_arrow.hip"

// LDS needed by the always-resident small arrays of the fast path
size_t slam_dim(int P_max) { return 16 * (((size_t)3 * P_max + 1 + 15) / 16); }
size_t slam_small_bytes(int P_max, int L_max, int M_max) {  // (SlamCtx::setup)
  return (size_t)P_max * 64 + (size_t)L_max * 16 + (size_t)L_max * 64 + (size_t)L_max * 8 * ((P_max + 63) / 64) +
         (size_t)(P_max + 2) * 4 + (size_t)(L_max + 2) * 8 + (size_t)M_max * 7 + (size_t)L_max * 2 + 224;
}
// LDS the arrow path cannot do without at full capacity: tables + the packed landmark system or the panels of the
// workspace variant (factor records and the observation table overflow to the workspace)
size_t arrow_lds_bytes(int P_max, int L_max, int M_max) {
  const size_t N = 16 * (((size_t)2 * L_max + 1 + 15) / 16);
  const size_t Tn = N / 16;
  // packed in LDS; register tiles + panels in LDS; or everything streamed from the workspace (E tiles + scratch in LDS)
  const size_t sys = N <= 16 * kFastTilesArrow ? sweep_region_doubles(N)
                     : Tn * (Tn + 1) / 2 <= (size_t)kArrowRegTiles * (kWaves - 1) ? 32 * N + 1280 : 1280;
  return arrow_small_bytes(P_max, L_max, M_max) + sys * 8 + 64;
}

}  // namespace kslam

// true when the fused LDS-resident kernel applies to trajectories of up to P_max poses
bool drlgx_slam_in_lds(int P_max, int L_max, int M_max) {
  const size_t n = kslam::slam_dim(P_max), nf = std::max<size_t>(n, 16 * kslam::kFastTiles);
  return n <= (size_t)16 * kslam::kDenseTiles &&
         kslam::slam_small_bytes(P_max, L_max, M_max) + kslam::sweep_region_doubles(nf) * 8 <= (size_t)kslam::kLdsBudget;
}
// capacities the SLAM kernels can serve at all (checked by drlgx_create)
bool drlgx_slam_capacity_ok(int P_max, int L_max, int M_max) {
  // (any number of landmarks: beyond the register-tile sweep the landmark system is streamed from the workspace)
  return kslam::arrow_lds_bytes(P_max, L_max, M_max) <= (size_t)kslam::kLdsBudget;
}
// doubles of HBM workspace per instance: X (3 P x (2 L + 1), row stride rounded up to 4), the selected-inverse blocks of the
// chain (6 + 9 + 9 per pose), the leaf -> right-separator rhs scratch, the square landmark system of the workspace variant, the factor records and the observation
// table when they do not fit the LDS
size_t drlgx_slam_ws_doubles(int P_max, int L_max, int M_max) {
  const size_t ldx = (size_t)((2 * L_max + 1 + 3 + 3) & ~3);  // (+ the three unit columns of the newest pose: arrow_body)
  const size_t n = (size_t)3 * P_max * ldx + (size_t)24 * P_max + (size_t)(P_max / kslam::kSeg + 2) * 3 * ldx + (size_t)(2 * L_max + 17) * (2 * L_max + 17) + (size_t)32 * (2 * L_max + 17) +
                   (size_t)M_max * kslam::REC + ((size_t)L_max * P_max * 2 + 7) / 8 + 16;
  return (n + 31) & ~(size_t)31;  // instances stay 256-byte aligned: 32-byte row loads of X
}

void drlgx_launch_slam(const DrlgxState &S, hipStream_t st, LaunchSel sel, int p_bound) {
  const int Pb = p_bound < S.P_max ? p_bound : S.P_max;
  if (sel.pcap <= 0) sel.pcap = Pb;
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kslam::k_slam<kslam::kFastTiles>),
                       reinterpret_cast<const void *>(&kslam::k_slam_arrow<0>),
                       reinterpret_cast<const void *>(&kslam::k_slam_arrow<kslam::kArrowRegTiles>)};
  drlgx_ensure_lds_attr(attr_set, fns, 3, kslam::kLdsBudget);
  const dim3 grid(sel.n), block(kslam::kThreads);
  // (the whole LDS is requested: what the tables and the system leave free holds the factor records and the observation table)
  if (drlgx_slam_in_lds(Pb, S.L_max, S.M_max))
    hipLaunchKernelGGL((kslam::k_slam<kslam::kFastTiles>), grid, block, kslam::kLdsBudget, st, DRLGX_KS_ARG(S), sel, kslam::kLdsBudget);
  else if (2 * S.L_max + 1 <= 16 * kslam::kFastTilesArrow)  // landmark system always packed in LDS: no register-tile sweep compiled in
    hipLaunchKernelGGL((kslam::k_slam_arrow<0>), grid, block, kslam::kLdsBudget, st, DRLGX_KS_ARG(S), sel, kslam::kLdsBudget);
  else
    hipLaunchKernelGGL((kslam::k_slam_arrow<kslam::kArrowRegTiles>), grid, block, kslam::kLdsBudget, st, DRLGX_KS_ARG(S), sel, kslam::kLdsBudget);
}
#pragma clang fp contract(off)
