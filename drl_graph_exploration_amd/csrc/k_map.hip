// Virtual-map kernel: occupancy rebuild + covariance propagation (EKF push-through of every core
// pose onto the virtual-landmark grid, fused by covariance intersection) + utility reductions.
// One 512-thread workgroup per instance (8 waves: 2 per SIMD is what the ~160 VGPRs of the propagation code allow
// without spilling; the kernel is bound by fp64 instruction issue, not by latency, so more waves do not help).
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
// The kernel is issue-bound on fp64, so (1) only (cell, pose) pairs that interact are visited, (2) divisions and square
// roots whose results are only compared against a tolerance use v_rcp/v_rsq + Newton (rcp_n / rsqrt_n), never the
// ones that feed a decision, and
// (3) two exact shortcuts avoid fp64 transcendentals:
//   * the field-of-view test needs atan2 only inside a thin wedge around the sensor's blind ray; cells
//     that are provably inside the FOV (d.x >= 0, or |d.y| > tan(blind half-angle + 1 mrad) |d.x|) skip it;
//   * when the 3-degree sector sweep covers the whole circle its bounding box provably contains every
//     in-range cell, so the sweep (120 sincos per pose) and the box tests are skipped.
// Compiled with -ffp-contract=off (thresholded decisions must round like the CPU reference).
#include "drlgx_dev.h"

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

// VirtualMap::predictVirtualLandmark (VirtualMap.cpp:213-229). info: symmetric xx xy xt yy yt tt.
template <bool kCheckFov>
__device__ __forceinline__ bool predict_cell(const DrlgxState &S, const Pose &ps, const double *pi, const P2 &pt,
                                             double &oxx, double &oxy, double &oyy) {
  const drlgx_config &cfg = S.cfg;
  // Jacobians of bearing (Pose2::bearing) and range (Pose2::range); the bearing VALUE is only needed for the
  // FOV check, which in_fov() answers without atan2 for almost every cell
  const P2 d = transform_to(ps, pt);
  const double d2 = d.x * d.x + d.y * d.y;
  const double gx = pt.x - ps.x, gy = pt.y - ps.y;
  const double g2 = gx * gx + gy * gy;
  // range < max_range && range > min_range, decided exactly on the squared distance (host-computed thresholds)
  if (!(g2 < S.r2_max_lt && g2 > S.r2_min_gt)) return false;
  if (kCheckFov && !in_fov(S, ps, pt)) return false;
  const double rrange = rsqrt_n(g2);  // 1 / range
  double Hbx[3], Hbl[2], Hrx[3], Hrl[2];
  if (d2 > 1e-10) {  // |d| > 1e-5
    const double rd2 = rcp_n(d2);
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
  // (Hl^T Hl)^-1 Hl^T  (Eigen fixed 2x2 inverse = adjugate / det)
  const double h00 = Hl0 * Hl0 + Hl2 * Hl2, h01 = Hl0 * Hl1 + Hl2 * Hl3;
  const double h10 = Hl1 * Hl0 + Hl3 * Hl2, h11 = Hl1 * Hl1 + Hl3 * Hl3;
  const double id = rcp_n(h00 * h11 - h01 * h10);
  const double i00 = h11 * id, i01 = -h01 * id, i10 = -h10 * id, i11 = h00 * id;
  const double p00 = i00 * Hl0 + i01 * Hl1, p01 = i00 * Hl2 + i01 * Hl3;
  const double p10 = i10 * Hl0 + i11 * Hl1, p11 = i10 * Hl2 + i11 * Hl3;
  // S = R + Hx * info.llt().solve(Hx^T); the LLT factor of the pose information (and the reciprocals of its
  // diagonal) was computed once per pose: pi = l00 l10 l11 l20 l21 l22 r00 r11 r22
  double xb0, xb1, xb2, xr0, xr1, xr2;
  {
    const double l10 = pi[1], l20 = pi[3], l21 = pi[4], r00 = pi[6], r11 = pi[7], r22 = pi[8];
    double y0 = Hbx[0] * r00, y1 = (Hbx[1] - l10 * y0) * r11, y2 = (Hbx[2] - l20 * y0 - l21 * y1) * r22;
    xb2 = y2 * r22; xb1 = (y1 - l21 * xb2) * r11; xb0 = (y0 - l10 * xb1 - l20 * xb2) * r00;
    y0 = Hrx[0] * r00; y1 = (Hrx[1] - l10 * y0) * r11; y2 = (Hrx[2] - l20 * y0 - l21 * y1) * r22;
    xr2 = y2 * r22; xr1 = (y1 - l21 * xr2) * r11; xr0 = (y0 - l10 * xr1 - l20 * xr2) * r00;
  }
  double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
  s00 += Hbx[0] * xb0; s00 += Hbx[1] * xb1; s00 += Hbx[2] * xb2;
  s01 += Hbx[0] * xr0; s01 += Hbx[1] * xr1; s01 += Hbx[2] * xr2;
  s10 += Hrx[0] * xb0; s10 += Hrx[1] * xb1; s10 += Hrx[2] * xb2;
  s11 += Hrx[0] * xr0; s11 += Hrx[1] * xr1; s11 += Hrx[2] * xr2;
  s00 = R0 + s00; s01 = 0.0 + s01; s10 = 0.0 + s10; s11 = R3 + s11;
  // cov = Hp S Hp^T
  const double t00 = p00 * s00 + p01 * s10, t01 = p00 * s01 + p01 * s11;
  const double t10 = p10 * s00 + p11 * s10, t11 = p10 * s01 + p11 * s11;
  const double c00 = t00 * p00 + t01 * p01;
  const double c10 = t10 * p00 + t11 * p01, c11 = t10 * p10 + t11 * p11;
  // information = inverse(cov) by LLT (lower triangle of cov)
  inv2_llt_fast(c00, c10, c11, oxx, oxy, oyy);
  return true;
}

// VirtualMap::covarianceIntersection2D (VirtualMap.cpp:364-378), symmetric storage.  The LLT solve is
// written with two reciprocals instead of eight divisions (fp64 division is ~30 instructions here).
__device__ __forceinline__ void ci_fuse(double &axx, double &axy, double &ayy, double bxx, double bxy, double byy) {
  const double a = axx * ayy - axy * axy;
  const double b = bxx * byy - bxy * bxy;
  // m1.llt().solve(m2).trace(), with reciprocal square roots instead of sqrt + divisions (this loop is issue-bound)
  const double r00 = rsqrt_n(axx);
  const double l10 = axy * r00;
  const double r11 = rsqrt_n(ayy - l10 * l10);
  double tr = 0;
  {
    double y0 = bxx * r00, y1 = (bxy - l10 * y0) * r11;
    double x1 = y1 * r11, x0 = (y0 - l10 * x1) * r00;
    tr += x0;
    y0 = bxy * r00;
    y1 = (byy - l10 * y0) * r11;
    x1 = y1 * r11;
    tr += x1;
  }
  const double c = a * tr;
  const double d = a + b - c;
  double w = 0.5 * (2 * b - c) * rcp_n(d);
  if ((w < 0 && d < 0) || (w > 1 && d > 0))
    w = 0.0;
  else if ((w < 0 && d > 0) || (w > 1 && d < 0))
    w = 1.0;
  axx = w * axx + (1.0 - w) * bxx;
  axy = w * axy + (1.0 - w) * bxy;
  ayy = w * ayy + (1.0 - w) * byy;
}

__device__ __forceinline__ double block_sum(double v, double *scratch, int tid) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  double s = 0;
  for (int w = 0; w < kWaves; ++w) s += scratch[w];
  return s;
}

__device__ __forceinline__ void map_body(const DrlgxState &S, const LaunchSel &sel, int rebuild, int chunk) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = blockIdx.x;
  if (!sel.on(bi) || !sel.map_on(bi)) return;
  const int inst = sel.base + bi;
  const int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  // a rejected move leaves the belief as it was: nothing to rebuild, unless this is the one rebuild of a rollout
  if (cnt[C_FLAG] && !sel.map_last_only) return;
  const drlgx_config &cfg = S.cfg;
  const int P = cnt[C_P], L = cnt[C_L];
  const int V = S.V, cols = S.cols, rows = S.rows, W = S.win;
  // LDS carve
  double *sp = smem;                       // [P_max][4]
  double *si = sp + (size_t)S.P_max * 4;   // [P_max][6]
  double *sl = si + (size_t)S.P_max * 6;   // [P_max][9] LLT factor of the pose information + reciprocals
  double *stage = sl + (size_t)S.P_max * 9;  // [chunk][64][3]
  unsigned long long *mask = reinterpret_cast<unsigned long long *>(stage + (size_t)chunk * 64 * 3);  // [V]
  unsigned long long *omask = mask + V;  // [V] poses that see the cell (occupancy ladder)
  double *scratch = reinterpret_cast<double *>(omask + V);  // [kWaves]
  int *bbox = reinterpret_cast<int *>(scratch + kWaves + DRLGX_LO_TAB);  // [P_max][4] min_row max_row min_col max_col
  int *worg = bbox + (size_t)S.P_max * 4;             // [P_max][2] window origin row, col
  int *pskip = worg + (size_t)S.P_max * 2;            // [P_max]
  int *lmc = pskip + S.P_max;                         // [V] estimated landmarks per cell
  uint8_t *ltr = reinterpret_cast<uint8_t *>(lmc + V);  // [DRLGX_LO_TAB][4] ladder transitions
  double *lpv = scratch + kWaves;                      // [DRLGX_LO_TAB] ladder state -> cell probability
  int *pcount = reinterpret_cast<int *>(ltr + 4 * DRLGX_LO_TAB);  // number of (pose, cell) pairs in range (phase A)
  unsigned short *plist = reinterpret_cast<unsigned short *>(pcount + 1);  // [chunk * 64] their pair indices
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
    for (int e = tid; e < P * 4; e += kThreads) sp[e] = ep[e];
    for (int e = tid; e < P * 6; e += kThreads) si[e] = pin[e];
    const double *el = S.est_lm + (size_t)inst * S.L_max * 2;
    for (int v = tid; v < V; v += kThreads) lmc[v] = 0;
    for (int t = tid; t < S.lo_ntab; t += kThreads) {
      lpv[t] = S.lo_pv[t];
      reinterpret_cast<uint32_t *>(ltr)[t] = reinterpret_cast<const uint32_t *>(S.lo_tr)[t];
    }
    __syncthreads();
    for (int j = tid; j < L; j += kThreads) {
      // OccupancyMap::update(map): landmark cell (OccupancyMap.cpp:127-131); every landmark in a cell is one occupied update
      int r = (int)floor((el[2 * j + 1] - cfg.map_min_y) / cfg.resolution);
      int c = (int)floor((el[2 * j] - cfg.map_min_x) / cfg.resolution);
      if (!(r >= rows || r < 0 || c >= cols || c < 0)) atomicAdd(&lmc[r * cols + c], 1);
    }
    __syncthreads();
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
      // state.information.llt(): factor once per pose, keep reciprocals of the diagonal
      const LLT3 f(pi[0], pi[1], pi[2], pi[3], pi[4], pi[5]);
      double *o = sl + 9 * p;
      o[0] = f.l00; o[1] = f.l10; o[2] = f.l11; o[3] = f.l20; o[4] = f.l21; o[5] = f.l22;
      o[6] = 1.0 / f.l00; o[7] = 1.0 / f.l11; o[8] = 1.0 / f.l22;
    }
    __syncthreads();
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
    const double i0 = 1.0 / pow(cfg.sigma0, 2);
    const int extg = 20;
    // the ladder's transition table packed into registers when it has <= 16 states (4 bits per next state, 2 per flag):
    // a cell's walk over its sees-me bits is then pure ALU instead of one dependent LDS byte load per pose
    const bool fsm_reg = S.lo_ntab > 0 && S.lo_ntab <= 16;
    unsigned long long t_occ = 0ull, t_free = 0ull;
    unsigned int t_flag = 0u;
    if (fsm_reg)
      for (int st2 = 0; st2 < S.lo_ntab; ++st2) {
        t_occ |= (unsigned long long)(ltr[4 * st2] & 15) << (4 * st2);
        t_free |= (unsigned long long)(ltr[4 * st2 + 1] & 15) << (4 * st2);
        t_flag |= (unsigned int)(ltr[4 * st2 + 2] & 3) << (2 * st2);
      }
    for (int c0 = 0; c0 < P; c0 += chunk) {
      const int nc = min(chunk, P - c0);
      const bool last = c0 + nc >= P;
      for (int v = tid; v < V; v += kThreads) {
        mask[v] = 0ull;
        omask[v] = 0ull;
      }
      if (tid == 0) *pcount = 0;
      __syncthreads();
      // (pose, window cell) pairs: a cheap pass keeps the ones in range and in the field of view (~45 % of the window)
      // in a compact list, so that the EKF push-through below runs on full waves
      // (candidate e = 64 pl + 8 wr + wc: an 8 x 8 slot grid per pose whatever the window width W <= 8 - no integer divisions)
      for (int e0 = 0; e0 < nc * 64; e0 += kThreads) {
        const int e = e0 + tid;
        bool valid = false;
        if (e < nc * 64) {
          const int pl = e >> 6, widx = e & 63;
          const int p = c0 + pl;
          const int wr = widx >> 3, wc = widx & 7;
          if (!pskip[p] && wr < W && wc < W) {
            const int row = worg[2 * p] + wr, col = worg[2 * p + 1] + wc;
            if (row >= 0 && row < rows && col >= 0 && col < cols) {
              const Pose ps{sp[4 * p], sp[4 * p + 1], sp[4 * p + 2], sp[4 * p + 3]};
              const P2 pt{(col + 0.5) * cfg.resolution + cfg.map_min_x, (row + 0.5) * cfg.resolution + cfg.map_min_y};
              const double dx = ps.x - pt.x, dy = ps.y - pt.y;
              // KDTreeR2::queryRadiusNeighbors / OccupancyMap range test: sqrt(d2) < max_range, exactly; then the field of view
              valid = dx * dx + dy * dy < S.r2_max_lt && in_fov(S, ps, pt);
            }
          }
        }
        const unsigned long long bal = __ballot(valid);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(pcount, __popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
        if (valid) plist[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)e;
      }
      __syncthreads();
      const int npairs = *pcount;
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
        if (predict_cell<false>(S, ps, sl + 9 * p, pt, a, b, d)) {
          double *o = stage + ((size_t)pl * 64 + widx) * 3;
          o[0] = a; o[1] = b; o[2] = d;
          atomicOr(&mask[row * cols + col], 1ull << pl);
        }
      }
      __syncthreads();
      if (c0 == 0) DRLGX_PROF(S, 21);
      for (int v = tid; v < V; v += kThreads) {
        const int row = v / cols, col = v - row * cols;
        double axx = i0, axy = 0.0, ayy = i0;
        int u = 0;
        if (c0 > 0) {
          axx = ixx[v]; axy = ixy[v]; ayy = iyy[v];
          u = upd[v];
        }
        unsigned long long m = mask[v];
        while (m) {
          const int pl = __ffsll((long long)m) - 1;
          m &= m - 1;
          const int p = c0 + pl;
          const double *o = stage + ((size_t)pl * 64 + (row - worg[2 * p]) * 8 + (col - worg[2 * p + 1])) * 3;
          if (u) {
            ci_fuse(axx, axy, ayy, o[0], o[1], o[2]);
          } else {
            axx = o[0]; axy = o[1]; ayy = o[2];
            u = 1;
          }
        }
        ixx[v] = axx; ixy[v] = axy; iyy[v] = ayy;
        upd[v] = (uint8_t)u;
        // occupancy ladder (OccupancyMap.cpp:64-138): the landmarks of the cell, then the poses that see it in
        // trajectory order (ascending mask bits); between chunks the log-odds value is parked in prob[]
        double l = 0.0;  // LOGODDS_UNKNOWN
        int st = 0;      // ... as a state of the precomputed ladder (DrlgxState::lo_tr) when it is closed
        const bool fsm = S.lo_ntab > 0;
        if (c0 == 0) {
          for (int n = lmc[v]; n > 0; --n) {
            l = fmin(S.lo_max, fmax(S.lo_min, l + S.lo_occ));
            st = ltr[4 * st];
          }
        } else {
          l = prob[v];
          st = (int)l;
        }
        m = omask[v];
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
        if (S.prof && tid == 0 && blockIdx.x == 0 && v == 0) S.prof[22] = wall_clock64() + (l == 1.2345e300 ? 1 : 0);
        // VirtualMap::updateProbability: prob = sum over num_samples identical maps of p / n
        double pv = 0.0;
        if (fsm) {
          pv = lpv[st];
        } else {
          const double pv1 = logodds2prob(l);
          for (int s = 0; s < cfg.num_samples; ++s) pv += pv1 / cfg.num_samples;
        }
        prob[v] = pv;
        if (S.prof && tid == 0 && blockIdx.x == 0 && v == 0) S.prof[23] = wall_clock64() + (pv == 1.2345e300 ? 1 : 0);
        // reductions (Planner2D.cpp:321-366, VirtualMap.cpp:47-59)
        double ca, cb, cd;
        inv2_llt_fast(axx, axy, ayy, ca, cb, cd);
        const double tr = ca + cd;
        vtr[v] = tr;
        utr += 1.0 * tr;
        if (pv < cfg.occupancy_threshold) known += 1.0;
        const double wgt = pv > 0.49 ? 1.0 : 0.0;
        udet += wgt * rcp_n(axx * ayy - axy * axy);
        uwtr += wgt * tr;
        const double x = (col + 0.5) * cfg.resolution + cfg.map_min_x, y = (row + 0.5) * cfg.resolution + cfg.map_min_y;
        if ((pv < 0.49 || pv > 0.6) && cfg.map_min_x + extg <= x && x <= cfg.map_max_x - extg && cfg.map_min_y + extg <= y &&
            y <= cfg.map_max_y - extg)
          expl += 1.0;
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
    for (int k = 0; k < 5; ++k)
      for (int o = 32; o > 0; o >>= 1) r5[k] += __shfl_down(r5[k], o);
    __syncthreads();  // stage[] is free again
    if ((tid & 63) == 0)
      for (int k = 0; k < 5; ++k) stage[k * kWaves + (tid >> 6)] = r5[k];
    __syncthreads();
    if (tid < 5) {
      double acc = 0;
      for (int w = 0; w < kWaves; ++w) acc += stage[tid * kWaves + w];
      stage[5 * kWaves + tid] = acc;
    }
    __syncthreads();
    utr = stage[5 * kWaves + 0]; known = stage[5 * kWaves + 1]; expl = stage[5 * kWaves + 2];
    udet = stage[5 * kWaves + 3]; uwtr = stage[5 * kWaves + 4];
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

__global__ __launch_bounds__(kThreads) void k_map(DrlgxState S, LaunchSel sel, int rebuild, int chunk) {
  map_body(S, sel, rebuild, chunk);
}

}  // namespace kmap

static size_t map_lds_bytes(const DrlgxState &S, int chunk) {
  size_t d = (size_t)S.P_max * 19 + (size_t)chunk * 64 * 3 + 2 * (size_t)S.V + kmap::kWaves + DRLGX_LO_TAB;  // + two u64 masks per cell
  size_t i = (size_t)S.P_max * 7 + (size_t)S.V + DRLGX_LO_TAB + 1 + (size_t)chunk * 32;  // (+ pair counter, pair list)
  return d * sizeof(double) + i * sizeof(int) + 16;
}

// LDS bytes of k_map and the poses per A/C pass: all of them when the stage fits the LDS (<= 64: one mask bit per pose)
size_t drlgx_map_lds_bytes(const DrlgxState &S, int *chunk_out) {
  int chunk = S.P_max < 64 ? S.P_max : 64;
  while (chunk > 1 && map_lds_bytes(S, chunk) > 160 * 1024) chunk /= 2;
  if (chunk_out) *chunk_out = chunk;
  return map_lds_bytes(S, chunk);
}

void drlgx_launch_map(const DrlgxState &S, hipStream_t st, LaunchSel sel) {
  // sel.act_idx == -2 encodes "reductions only" (used after reset)
  int rebuild = 1;
  if (sel.act_idx == -2) {
    rebuild = 0;
    sel.act_idx = 0;
  }
  int chunk = 0;
  const size_t lds = drlgx_map_lds_bytes(S, &chunk);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kmap::k_map)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, 160 * 1024);
  hipLaunchKernelGGL(kmap::k_map, dim3(sel.n), dim3(kmap::kThreads), lds, st, S, sel, rebuild, chunk);
}
