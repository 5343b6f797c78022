// Virtual-map kernel: occupancy rebuild + covariance propagation (EKF push-through of every core
// pose onto the virtual-landmark grid, fused by covariance intersection) + utility reductions.
// One 512-thread workgroup per instance: ~100 KB of LDS at the bench state, i.e. ONE workgroup per CU (2 waves per SIMD).  By
// the SQ counters (profiles/r04_ab_map_two_workgroups_per_cu.txt) the VALU is busy 37 % of the time and 57 % of the wave-cycles
// are spent parked on waitcnt / barriers: the waves wait for each other at the phase barriers and along the dependent fusion
// chains, the kernel is neither HBM- nor issue-bound.  k_map_c below is the two-workgroups-per-CU form (compact carve, <= 128
// VGPRs): launched when there are more instances than CUs.  (Compiled without machine-level loop-invariant code motion - see the
// Makefile - both forms need ~125 VGPRs; with it this one took 211 and k_map_c spilled.)
//
// Reference: src/em_exploration/OccupancyMap.cpp:55-138 (log-odds ladder, bbox sector sweep),
// src/em_exploration/VirtualMap.cpp:47-84 (explored, updateProbability), :213-229
// (predictVirtualLandmark), :256-316 (updateInformation), :364-378 (covarianceIntersection2D),
// src/em_exploration/Planner2D.cpp:321-366 (calculateUncertainty / calculateUtility).
//
// Data flow (per instance): poses (x,y,c,s + 3x3 information, its LLT factor) are staged in LDS once.
//   phase A (pose-centric): the (pose, window cell) pairs of the W x W cell windows around the poses are spread evenly
//            over the threads: range + field-of-view test, EKF push-through of the pose covariance (3x3 LLT solve, 2x2
//            algebra in registers); the 2x2 information goes to an LDS stage and the pose's bit is set (LDS atomics) in
//            the cell's "updates me" mask and, for the occupancy model, in its "sees me" mask.
//   phase C (cell-centric, one pass): covariance-intersection fusion over the set bits in trajectory order; occupancy
//            ladder over the "sees me" bits as a host-built state machine (DrlgxState::lo_tr); probability, trace,
//            and the five utility sums.  Every cell is written once per belief update and never read back.
//   phase R: block reduction of the sums (wave shuffles + LDS).
// fp64 instruction count is what every phase pays, so (1) only (cell, pose) pairs that interact are visited, (2) divisions and
// square roots whose results are only compared against a tolerance use v_rcp/v_rsq + Newton (rcp_n / rsqrt_n), never the
// ones that feed a decision, and
// (3) two exact shortcuts avoid fp64 transcendentals:
//   * the field-of-view test needs atan2 only inside a thin wedge around the sensor's blind ray; cells
//     that are provably inside the FOV (d.x >= 0, or |d.y| > tan(blind half-angle + 1 mrad) |d.x|) skip it;
//   * when the 3-degree sector sweep covers the whole circle its bounding box provably contains every
//     in-range cell, so the sweep (120 sincos per pose) and the box tests are skipped.
// Compiled with -ffp-contract=off (thresholded decisions must round like the CPU reference).
#include "drlgx_dev.h"
#include <stdlib.h>

namespace kmap {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;

__device__ __forceinline__ double logodds2prob(double l) { return exp(l) / (1.0 + exp(l)); }

// BearingRangeSensorModel::check / checkWithoutMinRange on the bearing only (Simulator2D.cpp:100-111)
__device__ __forceinline__ bool in_fov(const DrlgxState &S, const Pose &ps, const P2 &pt) {
  const drlgx_config &cfg = S.cfg;
  if (S.fov_fast) {
    const P2 d = transform_to(ps, pt);
    if (d.x >= 0.0 || fabs(d.y) > S.fov_tan * fabs(d.x)) return true;  // provably inside: no atan2 needed
  }
  const double bearing = bearing_of<false>(ps, pt, nullptr, nullptr);
  return bearing < cfg.max_bearing && bearing > cfg.min_bearing;
}

// The algebra of predictVirtualLandmark behind its decisions: tolerance-only (the values are compared at 1e-7, no decision depends
// on them), so multiply-adds may contract - which this translation unit otherwise forbids for the decisions' sake.
__device__ __forceinline__ void predict_info(const DrlgxState &S, const Pose &ps, const double *pi, const P2 &pt, const P2 &d, double d2,
                                             double gx, double gy, double g2, double &oxx, double &oxy, double &oyy) {
#pragma clang fp contract(fast)
  const drlgx_config &cfg = S.cfg;
  const double rrange = rsqrt_n1(g2);  // 1 / range
  double Hbx[3], Hbl[2], Hrx[3], Hrl[2];
  if (d2 > 1e-10) {  // |d| > 1e-5
    const double rd2 = rcp_n1(d2);
    const double a = -d.y * rd2, b = d.x * rd2;
    Hbx[0] = a * -1.0;
    Hbx[1] = b * -1.0;
    Hbx[2] = a * d.y + b * -d.x;
    Hbl[0] = a * ps.c + b * -ps.s;
    Hbl[1] = a * ps.s + b * ps.c;
  } else {
    Hbx[0] = Hbx[1] = Hbx[2] = 0;
    Hbl[0] = Hbl[1] = 0;
  }
  {
    const double ux = gx * rrange, uy = gy * rrange;
    Hrx[0] = ux * -ps.c + uy * -ps.s;
    Hrx[1] = ux * ps.s + uy * -ps.c;
    Hrx[2] = 0;
    Hrl[0] = ux;
    Hrl[1] = uy;
  }
  const double R0 = cfg.bearing_noise * cfg.bearing_noise, R3 = cfg.range_noise * cfg.range_noise;
  const double Hl0 = Hbl[0], Hl1 = Hbl[1], Hl2 = Hrl[0], Hl3 = Hrl[1];
  // S = R + Hx * info.llt().solve(Hx^T); the LLT factor of the pose information (and the reciprocals of its
  // diagonal) was computed once per pose: pi = . l10 . l20 l21 . r00 r11 r22
  double xb0, xb1, xb2, xr0, xr1, xr2;
  {
    const double l10 = pi[1], l20 = pi[3], l21 = pi[4], r00 = pi[6], r11 = pi[7], r22 = pi[8];
    double y0 = Hbx[0] * r00, y1 = (Hbx[1] - l10 * y0) * r11, y2 = (Hbx[2] - l20 * y0 - l21 * y1) * r22;
    xb2 = y2 * r22; xb1 = (y1 - l21 * xb2) * r11; xb0 = (y0 - l10 * xb1 - l20 * xb2) * r00;
    y0 = Hrx[0] * r00; y1 = (Hrx[1] - l10 * y0) * r11;
    y2 = (0.0 - l20 * y0 - l21 * y1) * r22;  // Hrx[2] = 0
    xr2 = y2 * r22; xr1 = (y1 - l21 * xr2) * r11; xr0 = (y0 - l10 * xr1 - l20 * xr2) * r00;
  }
  const double s00 = R0 + (Hbx[0] * xb0 + Hbx[1] * xb1 + Hbx[2] * xb2);
  const double s01 = Hbx[0] * xr0 + Hbx[1] * xr1 + Hbx[2] * xr2;
  const double s11 = R3 + (Hrx[0] * xr0 + Hrx[1] * xr1);
  // The reference forms cov = Hp S Hp^T with Hp = (Hl^T Hl)^-1 Hl^T = Hl^-1 (Hl is square) and inverts it; the same
  // matrix without the two inversions:  information = cov^-1 = Hl^T S^-1 Hl  (tolerance-only algebra: the values
  // are compared at 1e-7, no decision depends on them)
  const double rdet = rcp_n1(s00 * s11 - s01 * s01);
  const double q00 = s11 * rdet, q01 = -s01 * rdet, q11 = s00 * rdet;
  const double u0 = q00 * Hl0 + q01 * Hl2, u1 = q00 * Hl1 + q01 * Hl3;  // (S^-1 Hl) row 0
  const double v0 = q01 * Hl0 + q11 * Hl2, v1 = q01 * Hl1 + q11 * Hl3;  // row 1
  oxx = Hl0 * u0 + Hl2 * v0;
  oxy = Hl0 * u1 + Hl2 * v1;
  oyy = Hl1 * u1 + Hl3 * v1;
}


// VirtualMap::predictVirtualLandmark (VirtualMap.cpp:213-229). info: symmetric xx xy xt yy yt tt.
template <bool kCheckFov>
__device__ __forceinline__ bool predict_cell(const DrlgxState &S, const Pose &ps, const double *pi, const P2 &pt,
                                             double &oxx, double &oxy, double &oyy) {
  // Jacobians of bearing (Pose2::bearing) and range (Pose2::range); the bearing VALUE is only needed for the
  // FOV check, which in_fov() answers without atan2 for almost every cell
  const P2 d = transform_to(ps, pt);
  const double d2 = d.x * d.x + d.y * d.y;
  const double gx = pt.x - ps.x, gy = pt.y - ps.y;
  const double g2 = gx * gx + gy * gy;
  // range < max_range && range > min_range, decided exactly on the squared distance (host-computed thresholds)
  if (!(g2 < S.r2_max_lt && g2 > S.r2_min_gt)) return false;
  if (kCheckFov && !in_fov(S, ps, pt)) return false;
  predict_info(S, ps, pi, pt, d, d2, gx, gy, g2, oxx, oxy, oyy);
  return true;
}

// VirtualMap::covarianceIntersection2D (VirtualMap.cpp:364-378), symmetric storage.  The LLT solve is
// written with two reciprocals instead of eight divisions (fp64 division is ~30 instructions here).
__device__ __forceinline__ void ci_fuse(double &axx, double &axy, double &ayy, double bxx, double bxy, double byy) {
  // (tolerance-only algebra: contraction allowed.  The fusion is ONE dependent chain per cell, walked ~30 times for the cells on the
  // trajectory; its depth is what the cell pass costs: c (3 operations behind the cell's current information), d, the reciprocal with
  // one Newton step folded into the product with the numerator, the clamps, one multiply-add per component.)
#pragma clang fp contract(fast)
  const double a = axx * ayy - axy * axy;
  const double b = bxx * byy - bxy * bxy;
  // c = a * m1.llt().solve(m2).trace() = det(m1) trace(m1^-1 m2) = trace(adj(m1) m2): no factorisation needed
  // (w is continuous across its clamps)
  const double c = (ayy * bxx - axy * bxy) + (axx * byy - axy * bxy);
  const double d = a + b - c;
  const double dxx = axx - bxx, dxy = axy - bxy, dyy = ayy - byy;  // (off the chain)
  const double r0 = __builtin_amdgcn_rcp(d);
  double w = ((b - 0.5 * c) * r0) * (2.0 - d * r0);  // 0.5 (2 b - c) / d
  {  // the reference's two clamps, as selects (no branches on the chain)
    const bool lo = w < 0, hi = w > 1, dn = d < 0, dp = d > 0;
    const bool zero = (lo & dn) | (hi & dp), one = (lo & dp) | (hi & dn);
    w = one ? 1.0 : w;
    w = zero ? 0.0 : w;
  }
  // w m1 + (1 - w) m2
  axx = w * dxx + bxx;
  axy = w * dxy + bxy;
  ayy = w * dyy + byy;
}

// Sum over the 64 lanes of a wave, result in lane 63: DPP row shifts within the 16-lane rows, then the two gfx9
// row broadcasts (15 -> next row, 31 -> upper half) - no LDS traffic, ~20 cycles per step.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_add(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, kCtrl, kRowMask, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), kCtrl, kRowMask, 0xf, true);
  return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum63(double v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row's sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
  return v;
}

// handed (k_step): est_pose / pose_info of this instance already are in the LDS where this stage keeps them (sp, si) and
// lm_lds holds the landmark estimates (LDS of the SLAM stage that this stage does not overwrite before it has read them):
// nothing is fetched back from HBM
// lo: entry threadIdx.x of the ladder tables (k_step fetches them before the SLAM stage: one HBM round trip less here)
// P >= 0: the instance's counts and rejected-move flag as k_step knows them from the simulator wave (else read from S.cnt)
struct LadderEntry {
  bool have;
  double pv;
  uint32_t tr;
  int P, L, flag;
};
// kCompact: the LDS-lean form that lets TWO workgroups share a CU (the stand-alone kernel when there are more instances than
// CUs: look-ahead rollouts, many envs per GPU).  It needs the full-circle sensor (bbox_noop: the poses that update a cell are
// then exactly the poses that see it, up to cells inside min_range - ONE mask per cell, those rare pairs carry a sentinel
// in their stage entry) and keeps the stage compact: entry k of the in-range pair list instead of a slot per (pose, window
// cell), found through a 16-bit index table.  Same arithmetic in the same order: bit-equal results.
#ifndef KMAPC_WAVES
#define KMAPC_WAVES 4
#endif
constexpr int kPairsPerPose = 40;  // in-range cells per pose the compact stage has room for (the disc of radius max_range
                                   // holds ~28 cell centres of the 7 x 7 window, never more than 32)
template <bool kCompact = false>
__device__ __forceinline__ void map_body(const DrlgxState &S, const LaunchSel &sel, int rebuild, int chunk, bool handed = false,
                                         const double *lm_lds = nullptr, LadderEntry lo = LadderEntry{false, 0.0, 0u, -1, 0, 0}) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = drlgx_tid(), lane = tid & 63, wave = tid >> 6;
  const int bi = drlgx_bid();
  if (!sel.on(bi) || !sel.map_on(bi)) return;
  const int inst = sel.base + bi;
  const int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  // a rejected move leaves the belief as it was: nothing to rebuild, unless this is the one rebuild of a rollout
  if ((lo.P >= 0 ? lo.flag : cnt[C_FLAG]) && !sel.map_last_only) return;
  const drlgx_config &cfg = S.cfg;
  const int P = lo.P >= 0 ? lo.P : cnt[C_P], L = lo.P >= 0 ? lo.L : cnt[C_L];
  const int V = S.V, cols = S.cols, rows = S.rows, W = S.win;
  // LDS carve; the per-pose tables hold pc poses: the launch's pose bound (LaunchSel::pcap), not the engine's capacity
  const int pc = sel.cap(S.P_max);
  if (rebuild && P > pc) {  // (the host's bound was wrong: flag it, touch nothing)
    if (tid == 0) atomicMin(S.status, DRLGX_E_CAPACITY);
    return;
  }
  double *sp = smem;                   // [pc][4]
  double *si = sp + (size_t)pc * 4;    // [pc][6]
  double *sl = si + (size_t)pc * 6;    // [pc][9] LLT factor of the pose information + reciprocals
  double *stage = sl + (size_t)pc * 9;  // [chunk][64][3]; kCompact: [chunk * kPairsPerPose][3]
  unsigned long long *mask = reinterpret_cast<unsigned long long *>(stage + (size_t)chunk * (kCompact ? kPairsPerPose : 64) * 3);  // [V]
  unsigned long long *omask = kCompact ? mask : mask + V;  // [V] poses that see the cell (occupancy ladder); kCompact: the same mask
  double *scratch = reinterpret_cast<double *>(omask + V);  // [kWaves]
  int *bbox = reinterpret_cast<int *>(scratch + kWaves + DRLGX_LO_TAB);  // [pc][4] min_row max_row min_col max_col
  int *worg = bbox + (size_t)pc * 4;                  // [pc][2] window origin row, col
  int *pskip = worg + (size_t)pc * 2;                 // [pc]
  int *lmc = pskip + pc;                              // [V] estimated landmarks per cell
  uint8_t *ltr = reinterpret_cast<uint8_t *>(lmc + V);  // [DRLGX_LO_TAB][4] ladder transitions
  double *lpv = scratch + kWaves;                      // [DRLGX_LO_TAB] ladder state -> cell probability
  int *pcount = reinterpret_cast<int *>(ltr + 4 * DRLGX_LO_TAB);  // number of (pose, cell) pairs in range (phase A)
  unsigned short *plist = reinterpret_cast<unsigned short *>(pcount + 1);  // [chunk * 64] their pair indices
  unsigned short *sidx = plist + (size_t)chunk * 64;  // kCompact: [chunk][64] stage entry of (pose, window slot)
  double *prob = S.vm_prob + (size_t)inst * V;
  double *ixx = S.vm_info + ((size_t)inst * 3 + 0) * V, *ixy = S.vm_info + ((size_t)inst * 3 + 1) * V,
         *iyy = S.vm_info + ((size_t)inst * 3 + 2) * V;
  uint8_t *upd = S.vm_upd + (size_t)inst * S.Vu;
  double *vtr = S.vm_tr + (size_t)inst * V;

  double utr = 0, known = 0, expl = 0, udet = 0, uwtr = 0;
  DRLGX_PROF(S, 16);
  if (rebuild) {
    const double *ep = S.est_pose + (size_t)inst * S.P_max * 4;
    const double *pin = S.pose_info + (size_t)inst * S.P_max * 6;
    if (!handed) {
      for (int e = tid; e < P * 4; e += kThreads) sp[e] = ep[e];
      for (int e = tid; e < P * 6; e += kThreads) si[e] = pin[e];
    }
    const double *el = (handed && lm_lds) ? lm_lds : S.est_lm + (size_t)inst * S.L_max * 2;
    for (int v = tid; v < V; v += kThreads) {  // (the first chunk's masks too, while the loads above are in flight)
      lmc[v] = 0;
      mask[v] = 0ull;
      if (!kCompact) omask[v] = 0ull;
    }
    if (tid == 0) *pcount = 0;
    if (lo.have) {
      if (tid < S.lo_ntab) {
        lpv[tid] = lo.pv;
        reinterpret_cast<uint32_t *>(ltr)[tid] = lo.tr;
      }
    } else {
      for (int t = tid; t < S.lo_ntab; t += kThreads) {
        lpv[t] = S.lo_pv[t];
        reinterpret_cast<uint32_t *>(ltr)[t] = reinterpret_cast<const uint32_t *>(S.lo_tr)[t];
      }
    }
    __syncthreads();
    DRLGX_PROF(S, 40);
    // landmarks on the last threads, poses on the first ones: both in the same barrier interval
    for (int j = kThreads - 1 - tid; j < L; j += kThreads) {
      // OccupancyMap::update(map): landmark cell (OccupancyMap.cpp:127-131); every landmark in a cell is one occupied update
      int r = (int)floor((el[2 * j + 1] - cfg.map_min_y) / cfg.resolution);
      int c = (int)floor((el[2 * j] - cfg.map_min_x) / cfg.resolution);
      if (!(r >= rows || r < 0 || c >= cols || c < 0)) atomicAdd(&lmc[r * cols + c], 1);
    }
    for (int p = tid; p < P; p += kThreads) {
      const double x = sp[4 * p], y = sp[4 * p + 1];
      int orow = (int)floor((y - cfg.map_min_y) / cfg.resolution);
      int ocol = (int)floor((x - cfg.map_min_x) / cfg.resolution);
      orow = min(max(0, orow), rows - 1);
      ocol = min(max(0, ocol), cols - 1);
      bbox[4 * p + 0] = orow; bbox[4 * p + 1] = orow; bbox[4 * p + 2] = ocol; bbox[4 * p + 3] = ocol;
      // window of candidate cells for the information update: one cell wider than the tightest
      // (open) interval so that no cell the reference's radius query accepts can fall outside
      worg[2 * p + 0] = (int)floor((y - cfg.max_range - cfg.map_min_y) / cfg.resolution - 0.5);
      worg[2 * p + 1] = (int)floor((x - cfg.max_range - cfg.map_min_x) / cfg.resolution - 0.5);
      const double *pi = si + 6 * p;
      pskip[p] = det3s(pi[0], pi[1], pi[2], pi[3], pi[4], pi[5]) < 1e-10 ? 1 : 0;  // VirtualMap.cpp:293-294
      // state.information.llt(): factored once per pose; the push-through needs the off-diagonal entries and the
      // reciprocals of the diagonal only (tolerance-only algebra: reciprocal square roots, no sqrt / division)
      double *o = sl + 9 * p;
      const double r00 = rsqrt_n1(pi[0]);
      const double l10 = pi[1] * r00, l20 = pi[2] * r00;
      const double r11 = rsqrt_n1(pi[3] - l10 * l10);
      const double l21 = (pi[4] - l20 * l10) * r11;
      const double r22 = rsqrt_n1(pi[5] - l20 * l20 - l21 * l21);
      o[1] = l10; o[3] = l20; o[4] = l21;
      o[6] = r00; o[7] = r11; o[8] = r22;
    }
    __syncthreads();
    DRLGX_PROF(S, 41);
    const bool use_bbox = !S.bbox_noop;
    if (use_bbox) {
      // bbox of the 3-degree sector sweep (OccupancyMap.cpp:79-96): (pose, sample) pairs in parallel
      for (int e = tid; e < P * S.n_sweep; e += kThreads) {
        const int p = e / S.n_sweep, k = e - p * S.n_sweep;
        const Pose ps{sp[4 * p], sp[4 * p + 1], sp[4 * p + 2], sp[4 * p + 3]};
        const double th0 = theta_of(ps), b = S.sweep_b[k];
        const double x = ps.x + cfg.max_range * cos(th0 + b);
        const double y = ps.y + cfg.max_range * sin(th0 + b);
        int row = (int)floor((y - cfg.map_min_y) / cfg.resolution);
        int col = (int)floor((x - cfg.map_min_x) / cfg.resolution);
        row = min(max(0, row), rows - 1);
        col = min(max(0, col), cols - 1);
        atomicMin(&bbox[4 * p + 0], row);
        atomicMax(&bbox[4 * p + 1], row);
        atomicMin(&bbox[4 * p + 2], col);
        atomicMax(&bbox[4 * p + 3], col);
      }
      __syncthreads();
    }
    DRLGX_PROF(S, 17);
    // ---- phases A / C, `chunk` poses at a time (one chunk unless P > 64 or the stage does not fit the LDS) ----
    // A (pose-centric): one wave per pose, one lane per cell of the W x W window around the pose: EKF push-through of the
    //    pose covariance to the cell (predict_cell); the 2x2 information goes to the LDS stage and the pose's bit is set in
    //    the cell's 64-bit mask (LDS atomic), so that
    // C (cell-centric) visits exactly the poses that update a cell, in trajectory order (ascending bits), for the
    //    covariance-intersection fusion - instead of testing every (cell, pose) pair.  The last chunk's C pass also runs
    //    the occupancy ladder (branch-free over the poses), writes the cell and accumulates the reductions: every cell is
    //    read and written once per belief update.
    const double i0 = S.vm_i0;
    const int extg = 20;
    // the ladder's transition table packed into registers when it has <= 16 states (4 bits per next state, 2 per flag,
    // packed on the host): a cell's walk over its sees-me bits is then pure ALU instead of one dependent LDS byte load
    // per pose
    const bool fsm_reg = S.lo_ntab > 0 && S.lo_ntab <= 16;
    const unsigned long long t_occ = S.lo_tocc, t_free = S.lo_tfree;
    const unsigned int t_flag = S.lo_tflag;
    for (int c0 = 0; c0 < P; c0 += chunk) {
      const int nc = min(chunk, P - c0);
      const bool last = c0 + nc >= P;
      if (c0 > 0) {  // (the first chunk's masks were cleared while the pose tables were loading)
        for (int v = tid; v < V; v += kThreads) {
          mask[v] = 0ull;
          if (!kCompact) omask[v] = 0ull;
        }
        if (tid == 0) *pcount = 0;
        __syncthreads();
      }
      if (c0 == 0) DRLGX_PROF(S, 42);
      // (pose, window cell) pairs: a cheap pass keeps the ones in range and in the field of view (~45 % of the window)
      // in a compact list, so that the EKF push-through below runs on full waves
      // (candidate e = 64 pl + 8 wr + wc: an 8 x 8 slot grid per pose whatever the window width W <= 8 - no integer
      // divisions; a wave tests one pose's window per round).  Returns 1: accepted, 0: rejected, 2: in range but the
      // field of view needs the exact bearing (thin wedge around the blind ray, or a narrow sensor).
      auto pair_test = [&](int e) -> int {
        const int pl = e >> 6, widx = e & 63;
        const int p = c0 + pl;
        const int wr = widx >> 3, wc = widx & 7;
        const int row = worg[2 * p] + wr, col = worg[2 * p + 1] + wc;
        const Pose ps{sp[4 * p], sp[4 * p + 1], sp[4 * p + 2], sp[4 * p + 3]};
        const P2 pt{(col + 0.5) * cfg.resolution + cfg.map_min_x, (row + 0.5) * cfg.resolution + cfg.map_min_y};
        const double dx = ps.x - pt.x, dy = ps.y - pt.y;
        // KDTreeR2::queryRadiusNeighbors / OccupancyMap range test: sqrt(d2) < max_range, exactly
        const bool inr = !pskip[p] && wr < W && wc < W && row >= 0 && row < rows && col >= 0 && col < cols &&
                         dx * dx + dy * dy < S.r2_max_lt;
        const P2 d = transform_to(ps, pt);
        const bool sure = S.fov_fast && (d.x >= 0.0 || fabs(d.y) > S.fov_tan * fabs(d.x));  // provably inside
        return inr ? (sure ? 1 : 2) : 0;
      };
      auto pair_exact = [&](int e) -> bool {  // BearingRangeSensorModel::check on the bearing itself
        const int pl = e >> 6, widx = e & 63;
        const int p = c0 + pl;
        const int row = worg[2 * p] + (widx >> 3), col = worg[2 * p + 1] + (widx & 7);
        const Pose ps{sp[4 * p], sp[4 * p + 1], sp[4 * p + 2], sp[4 * p + 3]};
        const P2 pt{(col + 0.5) * cfg.resolution + cfg.map_min_x, (row + 0.5) * cfg.resolution + cfg.map_min_y};
        const double bearing = bearing_of<false>(ps, pt, nullptr, nullptr);
        return bearing < cfg.max_bearing && bearing > cfg.min_bearing;
      };
      int e0 = 0;
      // four rounds at a time as straight-line code (the LDS loads of the four candidates overlap), one LDS atomic per
      // wave and group
      for (; e0 + 4 * kThreads <= nc * 64; e0 += 4 * kThreads) {
        int t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = pair_test(e0 + r * kThreads + tid);
        if (__ballot((t[0] | t[1] | t[2] | t[3]) & 2)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t[r] == 2) t[r] = pair_exact(e0 + r * kThreads + tid) ? 1 : 0;
        }
        unsigned long long bal[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bal[r] = __ballot(t[r] == 1);
        const int n0 = __popcll(bal[0]), n1 = __popcll(bal[1]), n2 = __popcll(bal[2]), n3 = __popcll(bal[3]);
        int base = 0;
        if (lane == 0 && (n0 + n1 + n2 + n3)) base = atomicAdd(pcount, n0 + n1 + n2 + n3);
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int off[4] = {0, n0, n0 + n1, n0 + n1 + n2};
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t[r] == 1) plist[base + off[r] + __popcll(bal[r] & below)] = (unsigned short)(e0 + r * kThreads + tid);
      }
      for (; e0 < nc * 64; e0 += kThreads) {  // remaining rounds (waves beyond the last pose's window skip)
        const int e = e0 + tid;
        if ((e0 >> 6) + wave >= nc) continue;
        int t = pair_test(e);
        if (t == 2) t = pair_exact(e) ? 1 : 0;
        const unsigned long long bal = __ballot(t == 1);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(pcount, __popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
        if (t == 1) plist[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)e;
      }
      __syncthreads();
      if (c0 == 0) DRLGX_PROF(S, 43);
      const int npairs = *pcount;
      if (kCompact && npairs > nc * kPairsPerPose) {  // (cannot happen for a disc-shaped footprint: flag it, leave the map alone)
        if (tid == 0) atomicMin(S.status, DRLGX_E_CAPACITY);
        return;
      }
      for (int k = tid; k < npairs; k += kThreads) {
        const int e = plist[k];
        const int pl = e >> 6, widx = e & 63;
        const int p = c0 + pl;
        const int wr = widx >> 3, wc = widx & 7;
        const int row = worg[2 * p] + wr, col = worg[2 * p + 1] + wc;
        const Pose ps{sp[4 * p], sp[4 * p + 1], sp[4 * p + 2], sp[4 * p + 3]};
        const P2 pt{(col + 0.5) * cfg.resolution + cfg.map_min_x, (row + 0.5) * cfg.resolution + cfg.map_min_y};
        const bool in_bbox = !use_bbox || !(row < bbox[4 * p] || row > bbox[4 * p + 1] || col < bbox[4 * p + 2] || col > bbox[4 * p + 3]);
        if (in_bbox) atomicOr(&omask[row * cols + col], 1ull << pl);  // OccupancyMap::update visits this cell
        double a, b, d;
        const bool upd_ok = predict_cell<false>(S, ps, sl + 9 * p, pt, a, b, d);
        if constexpr (kCompact) {
          // entry k of the pair list; the cell pass finds it through sidx[(pose, window slot)].  A pair that is seen but does not
          // update (inside min_range) keeps its bit - the one mask is the sees-me mask - and carries a negative sentinel
          double *o = stage + (size_t)k * 3;
          o[0] = upd_ok ? a : -1.0; o[1] = b; o[2] = d;
          sidx[pl * 64 + (row & 7) * 8 + (col & 7)] = (unsigned short)k;
        } else if (upd_ok) {
          // stage slot of (pose, cell): the window spans at most 8 consecutive rows / columns, so (row mod 8, col mod 8)
          // is unique within it - the cell pass finds the entry without the window origin
          double *o = stage + ((size_t)pl * 64 + (row & 7) * 8 + (col & 7)) * 3;
          o[0] = a; o[1] = b; o[2] = d;
          atomicOr(&mask[row * cols + col], 1ull << pl);
        }
      }
      __syncthreads();
      if (c0 == 0) DRLGX_PROF(S, 21);
      const bool wprof = S.prof && blockIdx.x == S.prof_block && lane == 0 && c0 == 0;
      long long ci_clk = 0;
      // cell pass: one 8 x 8 tile of cells per wave and round (lane = 8 (row mod 8) + (col mod 8)).  Cells of a tile
      // are seen by nearly the same poses, so the lanes' covariance-intersection chains have similar lengths and tiles
      // away from the trajectory skip the loop altogether (row-major strips of 64 cells cross the whole map instead).
      // (Tiles on the trajectory cost several times the others and the round-robin deal leaves some waves idle from 2.2 us
      // while others work until 6.5 us; a tile queue - with the utility terms summed per tile, or parked per cell and summed
      // afterwards, to keep the sums reproducible - and a deal by ranked cost both evened the waves out and both made the
      // kernel slower: ranking costs more than it saves, and with the queue the barrier after the pass completed 1.9 us
      // after the last wave instead of 0.2 us.  Round 6 measured two more forms, both slower: two tiles per wave with their chains
      // interleaved in one loop (11.9 against 6.9 us: the waves are bound by fp64 issue on their SIMD, not by the chain's latency),
      // and the cells sorted by chain length - ballots and a prefix, deterministic - with 64 consecutive ones per wave (9.2 us: the
      // two classification passes and the prefix cost 4.9 us, and the longest chain alone, ~30 fusions of ~140 ns, lasts 4.2 us -
      // that chain is the floor of this pass; 97 against 89 us at 2 048 instances).)
      const int tiles_c = (cols + 7) >> 3, ntiles = ((rows + 7) >> 3) * tiles_c;
      // the untouched cell (prior information I / sigma0^2, ladder state 0), as the general path computes it
      double pv_prior = 0.0;
      if (S.lo_ntab > 0) {
        pv_prior = lpv[0];
      } else {
        const double pv1 = logodds2prob(0.0);
        for (int s2 = 0; s2 < cfg.num_samples; ++s2) pv_prior += pv1 / cfg.num_samples;
      }
      const double rdet_prior = rcp_n1(i0 * i0 - 0.0 * 0.0), tr_prior = (i0 + i0) * rdet_prior;
      const double wgt_prior = pv_prior > 0.49 ? 1.0 : 0.0;
      const bool expl_prior = pv_prior < 0.49 || pv_prior > 0.6;
      // the three per-cell LDS words of a tile (update mask, sees-me mask, landmark count) are fetched one tile ahead: each
      // is the head of a dependent chain and a wave has nothing else to cover the LDS latency with
      auto cell_of = [&](int t, int &row, int &col) -> int {
        const int trow = t / tiles_c, tcol = t - trow * tiles_c;
        row = 8 * trow + (lane >> 3);
        col = 8 * tcol + (lane & 7);
        return (row < rows && col < cols) ? row * cols + col : -1;
      };
      int nrow = 0, ncol = 0;
      int nv = wave < ntiles ? cell_of(wave, nrow, ncol) : -1;
      unsigned long long nm = nv >= 0 ? mask[nv] : 0ull, nom = (!kCompact && nv >= 0) ? omask[nv] : 0ull;
      int nlmc = nv >= 0 ? lmc[nv] : 0;
      for (int t = wave; t < ntiles; t += kWaves) {
        const int row = nrow, col = ncol;
        const bool ok = nv >= 0;
        const int v = ok ? nv : 0;
        unsigned long long m = nm;
        const unsigned long long om_cell = kCompact ? nm : nom;
        const int lmc_cell = nlmc;
        if (t + kWaves < ntiles) {
          nv = cell_of(t + kWaves, nrow, ncol);
          nm = nv >= 0 ? mask[nv] : 0ull;
          if (!kCompact) nom = nv >= 0 ? omask[nv] : 0ull;
          nlmc = nv >= 0 ? lmc[nv] : 0;
        }
        // a middle chunk of poses (trajectories beyond 128 poses) that neither updates nor sees any cell of the tile leaves
        // it as it is: no read-modify-write of its planes
        if (c0 > 0 && !last && __ballot((m | om_cell) != 0ull) == 0ull) continue;
        // a tile that no pose sees and no landmark lies in (most of the map: ~15 of 25 tiles at the bench state), whole
        // trajectory in one chunk: every cell is the untouched prior - the same values the general path below computes for
        // such a cell (same expressions, same order of the utility sums), without its ladder walk and per-cell algebra
        if (c0 == 0 && last && __ballot((m | om_cell) != 0ull || lmc_cell != 0) == 0ull) {
          if (ok) {
            ixx[v] = i0; ixy[v] = 0.0; iyy[v] = i0;
            upd[v] = (uint8_t)0;
            prob[v] = pv_prior;
            vtr[v] = tr_prior;
            utr += 1.0 * tr_prior;
            if (pv_prior < cfg.occupancy_threshold) known += 1.0;
            udet += wgt_prior * rdet_prior;
            uwtr += wgt_prior * tr_prior;
            if (expl_prior) {
              const double x = (col + 0.5) * cfg.resolution + cfg.map_min_x, y = (row + 0.5) * cfg.resolution + cfg.map_min_y;
              if (cfg.map_min_x + extg <= x && x <= cfg.map_max_x - extg && cfg.map_min_y + extg <= y && y <= cfg.map_max_y - extg) expl += 1.0;
            }
          }
          continue;
        }
        double axx = i0, axy = 0.0, ayy = i0;
        int u = 0;
        if (c0 > 0) {
          axx = ixx[v]; axy = ixy[v]; ayy = iyy[v];
          u = upd[v];
        }
        const double *nx = stage + lane * 3;  // + 192 * (pose within the chunk): the slot is the lane
        const long long tc0 = wprof ? wall_clock64() : 0;
        if constexpr (kCompact) {
          // the same walk over the set bits in trajectory order; an entry is found through the index table and fetched one
          // step ahead of its use, an entry with the sentinel is a pose that sees the cell without updating it
          auto fetch = [&](unsigned long long &mm, double &xx, double &xy, double &yy) {
            const int pl = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const double *o = stage + (size_t)sidx[pl * 64 + lane] * 3;
            xx = o[0]; xy = o[1]; yy = o[2];
          };
          if (m) {
            double nxx, nxy, nyy;
            fetch(m, nxx, nxy, nyy);
            while (true) {
              const double bxx = nxx, bxy = nxy, byy = nyy;
              const bool more = m != 0ull;
              if (more) fetch(m, nxx, nxy, nyy);
              if (bxx >= 0.0) {
                if (!u) {  // the first update of an untouched cell replaces the prior (VirtualMap.cpp:300-304)
                  axx = bxx; axy = bxy; ayy = byy;
                  u = 1;
                } else {
                  ci_fuse(axx, axy, ayy, bxx, bxy, byy);
                }
              }
              if (!more) break;
            }
          }
        } else {
        if (m && !u) {  // the first update of an untouched cell replaces the prior (VirtualMap.cpp:300-304)
          const double *o = nx + (__ffsll((long long)m) - 1) * 192;
          axx = o[0]; axy = o[1]; ayy = o[2];
          u = 1;
          m &= m - 1;
        }
        if (m) {
          // every further update is fused; the next entry is loaded before the current one is fused, so that the LDS
          // latency stays off the dependent chain
          const double *o = nx + (__ffsll((long long)m) - 1) * 192;
          double nxx = o[0], nxy = o[1], nyy = o[2];
          m &= m - 1;
          while (true) {
            const double bxx = nxx, bxy = nxy, byy = nyy;
            const bool more = m != 0ull;
            if (more) {
              const double *o2 = nx + (__ffsll((long long)m) - 1) * 192;
              nxx = o2[0]; nxy = o2[1]; nyy = o2[2];
              m &= m - 1;
            }
            ci_fuse(axx, axy, ayy, bxx, bxy, byy);
            if (!more) break;
          }
        }
        }
        if (wprof) ci_clk += wall_clock64() - tc0;
        if (ok) {
          ixx[v] = axx; ixy[v] = axy; iyy[v] = ayy;
          upd[v] = (uint8_t)u;
          // occupancy ladder (OccupancyMap.cpp:64-138): the landmarks of the cell, then the poses that see it in
          // trajectory order (ascending mask bits); between chunks the log-odds value is parked in prob[]
          double l = 0.0;  // LOGODDS_UNKNOWN
          int st = 0;      // ... as a state of the precomputed ladder (DrlgxState::lo_tr) when it is closed
          const bool fsm = S.lo_ntab > 0;
          if (c0 == 0) {
            for (int n = lmc_cell; n > 0; --n) {
              l = fmin(S.lo_max, fmax(S.lo_min, l + S.lo_occ));
              st = ltr[4 * st];
            }
          } else {
            l = prob[v];
            st = (int)l;
          }
          m = om_cell;
          if (fsm_reg) {
            // a state that maps to itself is absorbing (the transition depends on the state only): the remaining bits
            // cannot change it (cells at the clamped minimum / maximum, i.e. every cell seen more than a few times)
            while (m) {
              m &= m - 1;
              const int f = (t_flag >> (2 * st)) & 3;
              const int nst = (f & 1) ? st : (int)((((f & 2) ? t_occ : t_free) >> (4 * st)) & 15);
              if (nst == st) break;
              st = nst;
            }
            l = (double)st;
          } else if (fsm) {
            while (m) {
              m &= m - 1;
              const int f = ltr[4 * st + 2];
              st = (f & 1) ? st : ((f & 2) ? ltr[4 * st] : ltr[4 * st + 1]);
            }
            l = (double)st;
          } else {
            while (m) {
              m &= m - 1;
              if (fabs(l - S.lo_min) < 1e-5) continue;
              const double add = (l > S.occ_thresh + 1e-8) ? S.lo_occ : S.lo_free;
              l = fmin(S.lo_max, fmax(S.lo_min, l + add));
            }
          }
          if (!last) {
            prob[v] = l;
            continue;
          }
          // VirtualMap::updateProbability: prob = sum over num_samples identical maps of p / n
          double pv = 0.0;
          if (fsm) {
            pv = lpv[st];
          } else {
            const double pv1 = logodds2prob(l);
            for (int s2 = 0; s2 < cfg.num_samples; ++s2) pv += pv1 / cfg.num_samples;
          }
          prob[v] = pv;
          // reductions (Planner2D.cpp:321-366, VirtualMap.cpp:47-59)
          // trace and determinant of the covariance (= information^-1, 2 x 2): (a + d) / det, 1 / det
          const double rdet = rcp_n1(axx * ayy - axy * axy);
          const double tr = (axx + ayy) * rdet;
          vtr[v] = tr;
          utr += 1.0 * tr;
          if (pv < cfg.occupancy_threshold) known += 1.0;
          const double wgt = pv > 0.49 ? 1.0 : 0.0;
          udet += wgt * rdet;
          uwtr += wgt * tr;
          const double x = (col + 0.5) * cfg.resolution + cfg.map_min_x, y = (row + 0.5) * cfg.resolution + cfg.map_min_y;
          if ((pv < 0.49 || pv > 0.6) && cfg.map_min_x + extg <= x && x <= cfg.map_max_x - extg && cfg.map_min_y + extg <= y &&
              y <= cfg.map_max_y - extg)
            expl += 1.0;
        }
      }
      if (wprof) {
        S.prof[48 + wave] = wall_clock64();
        S.prof[56 + wave] = ci_clk;
      }
      __syncthreads();
    }
  } else {
    // reductions only (after reset): the cells are read back
    const int extg = 20;
    for (int v = tid; v < V; v += kThreads) {
      const int row = v / cols, col = v - row * cols;
      const double a = ixx[v], b = ixy[v], d = iyy[v], pv = prob[v];
      double ca, cb, cd;
      inv2_llt_s(a, b, d, ca, cb, cd);
      const double tr = ca + cd;
      vtr[v] = tr;
      utr += 1.0 * tr;
      if (pv < cfg.occupancy_threshold) known += 1.0;
      const double wgt = pv > 0.49 ? 1.0 : 0.0;
      udet += wgt / (a * d - b * b);
      uwtr += wgt * tr;
      const double x = (col + 0.5) * cfg.resolution + cfg.map_min_x, y = (row + 0.5) * cfg.resolution + cfg.map_min_y;
      if ((pv < 0.49 || pv > 0.6) && cfg.map_min_x + extg <= x && x <= cfg.map_max_x - extg && cfg.map_min_y + extg <= y &&
          y <= cfg.map_max_y - extg)
        expl += 1.0;
    }
  }
  // ---- phase R: block reduction of the five utility sums ----
  DRLGX_PROF(S, 19);
  {
    double r5[5] = {utr, known, expl, udet, uwtr};
#pragma unroll
    for (int k = 0; k < 5; ++k) r5[k] = wave_sum63(r5[k]);
    __syncthreads();  // stage[] is free again
    if ((tid & 63) == 63)
      for (int k = 0; k < 5; ++k) stage[k * kWaves + (tid >> 6)] = r5[k];
    __syncthreads();
    if (tid == 0) {  // wave-major order: the same summation order as before
      double acc[5];
      for (int k = 0; k < 5; ++k) {
        acc[k] = 0;
        for (int w = 0; w < kWaves; ++w) acc[k] += stage[k * kWaves + w];
      }
      utr = acc[0]; known = acc[1]; expl = acc[2]; udet = acc[3]; uwtr = acc[4];
    }
  }
  DRLGX_PROF(S, 20);
  if (tid == 0) {
    double *red = S.red + (size_t)inst * DRLGX_RED_STRIDE;
    red[R_UTR] = utr;
    red[R_KNOWN] = known;
    red[R_EXPL] = expl;
    red[R_UDET] = udet;
    red[R_UWTR] = uwtr;
  }
}

__global__ __launch_bounds__(kThreads) void k_map(DRLGX_KS_PARAM, LaunchSel sel, int rebuild, int chunk) {
  const DrlgxState &S = DRLGX_KS_REF;
  map_body<false>(S, sel, rebuild, chunk);
}
// The multi-resident form: <= 128 VGPRs (four waves per SIMD) and, with the compact carve, <= 80 KB of LDS - two workgroups
// per CU, each covering the other's dependent chains and barriers.  Launched when there are more instances than CUs.
__global__ __launch_bounds__(kThreads, KMAPC_WAVES) void k_map_c(DRLGX_KS_PARAM, LaunchSel sel, int rebuild, int chunk) {
  const DrlgxState &S = DRLGX_KS_REF;
  map_body<true>(S, sel, rebuild, chunk);
}

}  // namespace kmap

static size_t map_lds_bytes(const DrlgxState &S, int chunk, int pc, bool compact = false) {
  // pose tables, stage, the u64 mask(s) per cell, reduction scratch + ladder values
  size_t d = (size_t)pc * 19 + (size_t)chunk * (compact ? kmap::kPairsPerPose : 64) * 3 + (compact ? 1 : 2) * (size_t)S.V + kmap::kWaves + DRLGX_LO_TAB;
  // per-pose ints, landmark counts, ladder transitions, pair counter, pair list (+ the compact stage's index table)
  size_t i = (size_t)pc * 7 + (size_t)S.V + DRLGX_LO_TAB + 1 + (size_t)chunk * 32 * (compact ? 2 : 1);
  return d * sizeof(double) + i * sizeof(int) + 16;
}
namespace kmap {
// byte offset of the cell masks inside map_body's LDS carve (k_step checks that what the SLAM stage hands over lies below)
__host__ __device__ inline size_t masks_offset(int pc, int chunk) { return ((size_t)pc * 19 + (size_t)chunk * 64 * 3) * sizeof(double); }
}  // namespace kmap

// LDS bytes of k_map and the poses per A/C pass: all of them when the stage fits the LDS (<= 64: one mask bit per pose)
// (pcap: the launch's pose bound - the tables and the chunk follow it, not the engine's capacity)
size_t drlgx_map_lds_bytes(const DrlgxState &S, int *chunk_out, int pcap) {
  const int pc = pcap > 0 && pcap < S.P_max ? pcap : S.P_max;
  int chunk = pc < 64 ? pc : 64;
  while (chunk > 1 && map_lds_bytes(S, chunk, pc) > 160 * 1024) chunk /= 2;
  if (chunk_out) *chunk_out = chunk;
  return map_lds_bytes(S, chunk, pc);
}

// the compact carve for the launch's pose bound, if it lets two workgroups share a CU (<= 80 KB each): bytes, else 0
static size_t map_lds_bytes_compact(const DrlgxState &S, int pcap, int *chunk_out) {
  if (!S.bbox_noop) return 0;
  const int pc = pcap > 0 && pcap < S.P_max ? pcap : S.P_max;
  const int chunk = pc < 64 ? pc : 64;
  const size_t b = map_lds_bytes(S, chunk, pc, true);
  if (b > 80 * 1024) return 0;
  *chunk_out = chunk;
  return b;
}

// Which form of the stand-alone map kernel runs: DRLGX_MAP_COMPACT read ONCE (this is the host side of every map launch),
// overridden by drlgx_debug_map_form (the A/B of the parity test).
static int g_map_form = -2;
static int map_form() {
  if (g_map_form == -2) {
    const char *fv = getenv("DRLGX_MAP_COMPACT");
    g_map_form = fv ? atoi(fv) : -1;
  }
  return g_map_form;
}
extern "C" int drlgx_debug_map_form(int form) {
  const int was = map_form();
  g_map_form = form;
  return was;
}

// true when a rebuild launch of more instances than CUs would run the two-workgroups-per-CU form (the engine then keeps the map
// stage out of the fused step kernel and launches it separately)
bool drlgx_map_two_per_cu(const DrlgxState &S, int p_bound) {
  int cchunk = 0;
  return map_form() != 0 && map_lds_bytes_compact(S, p_bound, &cchunk) != 0;
}

void drlgx_launch_map(const DrlgxState &S, hipStream_t st, LaunchSel sel) {
  // sel.act_idx == -2 encodes "reductions only" (used after reset)
  int rebuild = 1;
  if (sel.act_idx == -2) {
    rebuild = 0;
    sel.act_idx = 0;
  }
  int chunk = 0;
  const size_t lds = drlgx_map_lds_bytes(S, &chunk, sel.pcap);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kmap::k_map), reinterpret_cast<const void *>(&kmap::k_map_c)};
  drlgx_ensure_lds_attr(attr_set, fns, 2, 160 * 1024);
  // the form of which two workgroups fit a CU, for launches with more instances than CUs (DRLGX_MAP_COMPACT=1)
  static int n_cu = 0;
  const int force = map_form();
  if (n_cu == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  int cchunk = 0;
  const size_t clds = rebuild ? map_lds_bytes_compact(S, sel.pcap, &cchunk) : 0;
  // more instances than CUs: the form of which two workgroups share a CU (each covers the other's barriers and dependent chains:
  // 97 against 136 us at 2 048 instances, profiles/r05_ab_machine_licm.txt); up to one instance per CU the resident form
  if (clds && (force == 1 || (force != 0 && sel.n > n_cu)))
    hipLaunchKernelGGL(kmap::k_map_c, dim3(sel.n), dim3(kmap::kThreads), clds, st, DRLGX_KS_ARG(S), sel, rebuild, cchunk);
  else
    hipLaunchKernelGGL(kmap::k_map, dim3(sel.n), dim3(kmap::kThreads), lds, st, DRLGX_KS_ARG(S), sel, rebuild, chunk);
}
