// Simulator kernels: environment reset (SS2D.__init__) and the move/measure/factor-append part of a
// belief step.  One 64-lane wave per instance: the lanes scan the ground-truth landmarks in
// parallel (ballot keeps libstdc++'s hash iteration order); the normal variates of a measure() call are
// produced by all lanes at once (speculative Marsaglia-polar candidates ranked by a ballot, drlgx_dev.h:
// draw_normals) and consumed in the reference's draw order, so the RNG streams stay bit-identical.
//
// Reference: src/em_exploration/Simulator2D.cpp:113-132,161-182,445-464,491-527;
// src/em_exploration/SLAM2D.cpp:44-57,70-89,103-124; scripts/envs/pyss2d.py:102-138,171-206.
// Compiled with -ffp-contract=off so that products/sums round as in the CPU reference.
#include "drlgx_dev.h"

namespace ksim {

struct SimCtx {
  const DrlgxState &S;
  int inst, lane;
  MtStream sensor, control;
  NormalState ns_sensor, ns_control;
  Pose veh;
  int P, L, M;
  int err;
  // the initial guess of the newest pose (what SLAM2D::addOdometry inserted, SLAM2D.cpp:70-89) when this wave has just formed it
  // itself: a first sighting's initial estimate is relative to it (Simulator2D.cpp:95-98), and reading it back from th_pose
  // would mean waiting for lane 0's store to drain first
  Pose origin;
  int has_origin;
};

// Simulator2D::measure (Simulator2D.cpp:505-527) + SLAM2D::addMeasurement (SLAM2D.cpp:103-124), wave-parallel.
//  1. the in-range ground-truth landmarks are compacted in libstdc++'s hash iteration order (ballot + prefix rank);
//  2. their 2 * n_in noise variates (bearing, range per landmark, drawn BEFORE the validity check:
//     BearingRangeSensorModel::measure, Simulator2D.cpp:113-132) come from draw_normals in stream order;
//  3. lane k evaluates landmark k; the valid ones are appended in order (factor index and new-landmark slot by prefix rank;
//     a measure() call sees every key at most once, so there is no intra-call conflict).
// `record == false` is SS2D.simulate's first measure() (obstacle logic, inert at safe_distance = 0): only the RNG advances.
// in-range ground-truth landmarks in libstdc++'s hash iteration order (ballot + prefix rank) -> inr[0..n_in); the set only
// depends on the vehicle pose, so the two measure() calls of a real step share one scan
// (GtPrefetch: up to 128 ground-truth landmarks - two per lane - fetched by the caller while it was waiting for other data)
struct GtPrefetch {
  bool ok;
  int key[2];
  double x[2], y[2];
};
__device__ inline void gt_prefetch_points(const SimCtx &c, GtPrefetch &g) {
  const double *gl = c.S.gt_lm + (size_t)c.S.parent[c.inst] * c.S.LG * 2;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    g.x[r] = g.ok ? gl[2 * g.key[r]] : 0.0;
    g.y[r] = g.ok ? gl[2 * g.key[r] + 1] : 0.0;
  }
}
__device__ inline int scan_in_range(SimCtx &c, int *inr, const GtPrefetch *pf = nullptr) {
  const DrlgxState &S = c.S;
  const drlgx_config &cfg = S.cfg;
  const double *gl = S.gt_lm + (size_t)S.parent[c.inst] * S.LG * 2;
  const int n_gt = cfg.num_landmarks;
  const unsigned long long below = (1ull << c.lane) - 1ull;
  int n_in = 0;
  if (pf && pf->ok) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool valid = 64 * r + c.lane < n_gt;
      const double dx = pf->x[r] - c.veh.x, dy = pf->y[r] - c.veh.y;
      const bool in = valid && (dx * dx + dy * dy < S.r2_max_lt);
      const unsigned long long mask = __ballot(in);
      if (in) inr[n_in + __popcll(mask & below)] = pf->key[r];
      n_in += __popcll(mask);
    }
    wave_lds_sync();
    return n_in;
  }
  for (int base = 0; base < n_gt; base += 64) {
    const int idx = base + c.lane;
    const bool valid = idx < n_gt;
    const int key = valid ? S.lm_order[idx] : 0;
    const double lx = valid ? gl[2 * key] : 0.0, ly = valid ? gl[2 * key + 1] : 0.0;
    const double dx = lx - c.veh.x, dy = ly - c.veh.y;
    const bool in = valid && (dx * dx + dy * dy < S.r2_max_lt);  // sqrt(d2) < max_range, exactly (host-computed threshold)
    const unsigned long long mask = __ballot(in);
    if (in) inr[n_in + __popcll(mask & below)] = key;
    n_in += __popcll(mask);
  }
  wave_lds_sync();
  return n_in;
}
// exp_*: the staged interface (drlgx_stage_measure) exports the valid measurements (key, bearing, range) in order instead
// of appending them: Simulator2D::measure as the object-level API sees it
// `lead`: variates of earlier measure() calls that are drawn (and dropped) together with this call's
// `lmbox` (k_step): LDS for 2 LG doubles.  The appended factors are then ALSO left in LDS for the SLAM stage of the same
// workgroup, which would otherwise wait for them to come back from HBM: factor M0 + r has its bearing / range in
// nrm[2 r], nrm[2 r + 1] and its landmark slot in inr[r] (in place: r <= the index a lane read its own inputs from), new
// landmark L0 + r its initial estimate in lmbox[2 r], lmbox[2 r + 1].
__device__ inline void measure(SimCtx &c, bool record, double *nrm, const int *inr, int n_in, int32_t *exp_keys = nullptr,
                               double *exp_br = nullptr, int32_t *exp_count = nullptr, int lead = 0, double *lmbox = nullptr) {
  const DrlgxState &S = c.S;
  const drlgx_config &cfg = S.cfg;
  const double *gl = S.gt_lm + (size_t)S.parent[c.inst] * S.LG * 2;
  const unsigned long long below = (1ull << c.lane) - 1ull;
  int *key_slot = S.key_slot + (size_t)c.inst * S.LG;
  // what the lanes need from HBM for their landmarks (ground-truth point, landmark slot) is requested BEFORE the draws: the round
  // trip then runs under them instead of at the head of the loop below (the first 128 in-range landmarks: two per lane)
  int pf_key[2] = {0, 0}, pf_slot[2] = {0, 0};
  double pf_x[2] = {0, 0}, pf_y[2] = {0, 0};
  if (record || exp_keys) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int k = 64 * r + c.lane;
      if (k < n_in) {
        pf_key[r] = inr[k];
        pf_x[r] = gl[2 * pf_key[r]];
        pf_y[r] = gl[2 * pf_key[r] + 1];
        if (record) pf_slot[r] = key_slot[pf_key[r]];
      }
    }
  }
  draw_normals(c.sensor, c.ns_sensor, lead + 2 * n_in, nrm, c.lane, (record || exp_keys) ? lead : lead + 2 * n_in);
  if (exp_keys) {
    int n_out = 0;
    for (int base = 0; base < n_in; base += 64) {
      const int k = base + c.lane;
      const bool v = k < n_in;
      const bool pf = base < 128;
      const int key = pf ? pf_key[base >> 6] : (v ? inr[k] : 0);
      const P2 lm = pf ? P2{pf_x[base >> 6], pf_y[base >> 6]} : P2{gl[2 * key], gl[2 * key + 1]};
      const double bn = (v ? nrm[2 * k] : 0.0) * cfg.bearing_noise + 0.0;
      const double rn = (v ? nrm[2 * k + 1] : 0.0) * cfg.range_noise + 0.0;
      const double bearing = bearing_of<false>(c.veh, lm, nullptr, nullptr) + bn;
      const double range = range_of<false>(c.veh, lm, nullptr, nullptr) + rn;
      const bool ok = v && bearing < cfg.max_bearing && bearing > cfg.min_bearing && range < cfg.max_range && range > cfg.min_range;
      const unsigned long long okm = __ballot(ok);
      if (ok) {
        const int f = n_out + __popcll(okm & below);
        exp_keys[f] = key;
        exp_br[2 * f] = bearing;
        exp_br[2 * f + 1] = range;
      }
      n_out += __popcll(okm);
    }
    if (c.lane == 0) *exp_count = n_out;
    return;
  }
  if (!record) return;
  const int L_first = c.L, M_first = c.M;
  for (int base = 0; base < n_in; base += 64) {
    const int k = base + c.lane;
    const bool v = k < n_in;
    const bool pf = base < 128;
    const int key = pf ? pf_key[base >> 6] : (v ? inr[k] : 0);
    const P2 lm = pf ? P2{pf_x[base >> 6], pf_y[base >> 6]} : P2{gl[2 * key], gl[2 * key + 1]};
    const double bn = (v ? nrm[2 * k] : 0.0) * cfg.bearing_noise + 0.0;  // RNG::normal(0, sd) = n01 * sd + 0
    const double rn = (v ? nrm[2 * k + 1] : 0.0) * cfg.range_noise + 0.0;
    const double bearing = bearing_of<false>(c.veh, lm, nullptr, nullptr) + bn;
    const double range = range_of<false>(c.veh, lm, nullptr, nullptr) + rn;
    const bool ok = v && bearing < cfg.max_bearing && bearing > cfg.min_bearing && range < cfg.max_range && range > cfg.min_range;
    int slot = ok ? (pf ? pf_slot[base >> 6] : key_slot[key]) : 0;
    const bool isnew = ok && slot < 0;
    const unsigned long long okm = __ballot(ok), newm = __ballot(isnew);
    const int n_ok = __popcll(okm), n_new = __popcll(newm);
    if (c.L + n_new > S.L_max || c.M + n_ok > S.M_max) {
      c.err = DRLGX_E_CAPACITY;
      return;
    }
    if (isnew) {
      slot = c.L + __popcll(newm & below);
      // origin = initial estimate of the measuring pose (it is never in result_ yet)
      Pose origin = c.origin;
      if (!c.has_origin) {
        const double *tp = S.th_pose + ((size_t)c.inst * S.P_max + (c.P - 1)) * 4;
        origin = Pose{tp[0], tp[1], tp[2], tp[3]};
      }
      const P2 g = transform_from(origin, P2{range * cos(bearing), range * sin(bearing)});  // Simulator2D.cpp:95-98
      double *tl = S.th_lm + ((size_t)c.inst * S.L_max + slot) * 2;
      tl[0] = g.x;
      tl[1] = g.y;
      double *dl = S.d_lm + ((size_t)c.inst * S.L_max + slot) * 2;
      dl[0] = 0;
      dl[1] = 0;
      S.lm_key[(size_t)c.inst * S.L_max + slot] = key;
      key_slot[key] = slot;
      if (lmbox) {
        lmbox[2 * (slot - L_first)] = g.x;
        lmbox[2 * (slot - L_first) + 1] = g.y;
      }
    }
    if (ok) {
      const int f = c.M + __popcll(okm & below);
      S.meas_pose[(size_t)c.inst * S.M_max + f] = c.P - 1;
      S.meas_lm[(size_t)c.inst * S.M_max + f] = slot;
      double *br = S.meas_br + ((size_t)c.inst * S.M_max + f) * 2;
      br[0] = bearing;
      br[1] = range;
      if (lmbox) {
        nrm[2 * (f - M_first)] = bearing;
        nrm[2 * (f - M_first) + 1] = range;
        const_cast<int *>(inr)[f - M_first] = slot;
      }
    }
    c.L += n_new;
    c.M += n_ok;
  }
}

// park: the caller copies the LDS images of the two streams (2 x 626 words, contiguous) to S.mt itself, later and with
// more than one wave (k_step); the counters are put into the images here
// mail: LDS words that receive the final counts (P, L, M) for the waves that wait for this one (k_step)
__device__ inline void store_ctx(SimCtx &c, bool park = false, int *mail = nullptr) {
  const DrlgxState &S = c.S;
  if (park) {
    mt_park(c.sensor, c.lane);
    mt_park(c.control, c.lane);
  } else {
    mt_store(c.sensor, S.mt + ((size_t)c.inst * 2 + 0) * DRLGX_MT_STRIDE, c.lane);
    mt_store(c.control, S.mt + ((size_t)c.inst * 2 + 1) * DRLGX_MT_STRIDE, c.lane);
  }
  if (c.lane == 0) {
    S.nrm_saved[c.inst * 2 + 0] = c.ns_sensor.saved;
    S.nrm_has[c.inst * 2 + 0] = c.ns_sensor.has;
    S.nrm_saved[c.inst * 2 + 1] = c.ns_control.saved;
    S.nrm_has[c.inst * 2 + 1] = c.ns_control.has;
    double *gp = S.gt_pose + (size_t)c.inst * 4;
    gp[0] = c.veh.x; gp[1] = c.veh.y; gp[2] = c.veh.c; gp[3] = c.veh.s;
    int *cnt = S.cnt + (size_t)c.inst * DRLGX_CNT_STRIDE;
    cnt[C_P] = c.P;
    cnt[C_L] = c.L;
    cnt[C_M] = c.M;
    if (mail) {
      mail[0] = c.P;
      mail[1] = c.L;
      mail[2] = c.M;
    }
    if (c.err) atomicMin(S.status, c.err);
  }
}

// SS2D.__init__ (pyss2d.py:102-138): seed, vehicle, landmarks, prior, first measure.
__global__ __launch_bounds__(64) void k_reset(DRLGX_KS_PARAM, int first_measure, const int32_t *env_ids, const uint32_t *seeds,
                                              const double *start) {
  const DrlgxState &S = DRLGX_KS_REF;
  __shared__ uint32_t lds[3][DRLGX_MT_STRIDE];
  extern __shared__ double dyn[];  // nrm[2 LG + 2] doubles, inr[LG] ints
  double *nrm = dyn;
  int *inr = reinterpret_cast<int *>(dyn + 2 * S.LG + 2);
  const int lane = threadIdx.x;
  const int inst = env_ids[blockIdx.x];
  const uint32_t seed = seeds[blockIdx.x];
  const drlgx_config &cfg = S.cfg;
  SimCtx c{S, inst, lane, {}, {}, {0, 0}, {0, 0}, {}, 0, 0, 0, 0};
  c.sensor = mt_seed(lds[0], seed, lane);   // Simulator2D.cpp:436-443: three RNGs, same seed
  c.control = mt_seed(lds[1], seed, lane);
  MtStream simrng = mt_seed(lds[2], seed, lane);
  const double *sp = start + (size_t)blockIdx.x * 3;
  c.veh = make_pose(sp[0], sp[1], sp[2]);
  // Simulator2D::addLandmarks (Simulator2D.cpp:445-464)
  double *gl = S.gt_lm + (size_t)inst * S.LG * 2;
  const int n_fixed = S.fixed_lm ? min(S.n_fixed, cfg.num_landmarks) : 0;  // the listed landmarks come first (keys 0 .. n_fixed - 1)
  for (int k = lane; k < 2 * n_fixed; k += 64) gl[k] = S.fixed_lm[k];
  for (int i = n_fixed; i < cfg.num_landmarks;) {
    double x = rng_uniform_real(simrng, cfg.env_min_x, cfg.env_max_x, lane);
    double y = rng_uniform_real(simrng, cfg.env_min_y, cfg.env_max_y, lane);
    double dx = x - c.veh.x, dy = y - c.veh.y;
    if (sqrt(dx * dx + dy * dy) < 2.0) continue;
    if (lane == 0) {
      gl[2 * i] = x;
      gl[2 * i + 1] = y;
    }
    i++;
  }
  for (int k = lane; k < S.LG; k += 64) S.key_slot[(size_t)inst * S.LG + k] = -1;
  if (lane == 0) {
    S.parent[inst] = inst;
    int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
    cnt[C_STEP] = 1;
    cnt[C_ISAM] = 0;
    cnt[C_NEWP] = 0;
    cnt[C_NEWL] = 0;
    cnt[C_FLAG] = 0;
    // a new episode never continues the previous one's covariance panel (k_inc.hip): the staged reset runs no solve that
    // would overwrite it, and (valid, P = 1) of an old episode would pass inc_precheck at this episode's second pose
    if (S.jc_meta) S.jc_meta[(size_t)inst * 4] = 0;
    // SLAM2D::addPrior (SLAM2D.cpp:44-57)
    double *pr = S.prior + (size_t)inst * DRLGX_PRIOR_STRIDE;
    pr[0] = c.veh.x; pr[1] = c.veh.y; pr[2] = c.veh.c; pr[3] = c.veh.s;
    for (int k = 0; k < 9; ++k) pr[4 + k] = 0.0;
    pr[4 + 0] = 1.0 / (cfg.sigma_x0 * cfg.sigma_x0);
    pr[4 + 4] = 1.0 / (cfg.sigma_y0 * cfg.sigma_y0);
    pr[4 + 8] = 1.0 / (cfg.sigma_theta0 * cfg.sigma_theta0);
    double *tp = S.th_pose + (size_t)inst * S.P_max * 4;
    tp[0] = c.veh.x; tp[1] = c.veh.y; tp[2] = c.veh.c; tp[3] = c.veh.s;
    double *ep = S.est_pose + (size_t)inst * S.P_max * 4;
    ep[0] = c.veh.x; ep[1] = c.veh.y; ep[2] = c.veh.c; ep[3] = c.veh.s;
    double *dp = S.d_pose + (size_t)inst * S.P_max * 3;
    dp[0] = dp[1] = dp[2] = 0;
    double *red = S.red + (size_t)inst * DRLGX_RED_STRIDE;
    for (int k = 0; k < DRLGX_RED_STRIDE; ++k) red[k] = 0;
  }
  __syncthreads();
  c.P = 1;
  if (first_measure) measure(c, true, nrm, inr, scan_in_range(c, inr));  // pyss2d.py:135 self.measure()
  store_ctx(c);
  // VirtualMap::initialize (VirtualMap.cpp:318-362): prob 0.5, information I / sigma0^2
  const double i0 = 1.0 / pow(cfg.sigma0, 2);
  for (int v = lane; v < S.V; v += 64) {
    S.vm_prob[(size_t)inst * S.V + v] = 0.5;
    S.vm_info[((size_t)inst * 3 + 0) * S.V + v] = i0;
    S.vm_info[((size_t)inst * 3 + 1) * S.V + v] = 0.0;
    S.vm_info[((size_t)inst * 3 + 2) * S.V + v] = i0;
    S.vm_upd[(size_t)inst * S.Vu + v] = 0;
  }
}

// SS2D.simulate bounds-checks the odometry increment against the map box (pyss2d.py:173-176): the one predicate both the
// simulator below and the fused step kernel's prelude (which decides whether a SLAM front end may run ahead) evaluate
__device__ __forceinline__ bool odom_in_bounds(const drlgx_config &cfg, double ox, double oy) {
  return (cfg.map_min_x < ox && ox < cfg.map_max_x) && (cfg.map_min_y < oy && oy < cfg.map_max_y);
}
// ... and the whole acceptance test of a move for an instance that holds P poses (capacity: DRLGX_E_CAPACITY)
__device__ __forceinline__ bool move_accepted(const DrlgxState &S, double ox, double oy, int P) {
  return odom_in_bounds(S.cfg, ox, oy) && P < S.P_max;
}

// What the simulator wave reads of its instance that does not depend on the pose count, as the caller's own loads left it in
// registers: k_step's prelude requests it together with the counts, so that the wave starts to draw the moment it is released
// instead of a round trip to HBM (~2.7 us at the head of a 256-workgroup launch) later.
struct SimPre {
  uint4 mt[5];            // the two random streams (mt_load2_issue)
  double ns_s, ns_c;      // std::normal_distribution's saved variates and flags
  int nh_s, nh_c;
  Pose veh;               // ground-truth pose
  int step0;
  double dist0;
  GtPrefetch gt;          // keys of the range scan
  bool have = false;
};
__device__ __forceinline__ void sim_preload(const DrlgxState &S, int inst, int lane, int n_measure, SimPre &p) {
  p.ns_s = S.nrm_saved[inst * 2 + 0];
  p.ns_c = S.nrm_saved[inst * 2 + 1];
  p.nh_s = S.nrm_has[inst * 2 + 0];
  p.nh_c = S.nrm_has[inst * 2 + 1];
  const double *gp = S.gt_pose + (size_t)inst * 4;
  p.veh = Pose{gp[0], gp[1], gp[2], gp[3]};
  p.step0 = S.cnt[(size_t)inst * DRLGX_CNT_STRIDE + C_STEP];
  p.dist0 = S.red[(size_t)inst * DRLGX_RED_STRIDE + R_DIST];
  mt_load2_issue(S.mt + (size_t)inst * 2 * DRLGX_MT_STRIDE, lane, p.mt);
  p.gt.ok = false;
  if (n_measure > 0) {
    const int n_gt = S.cfg.num_landmarks;
    p.gt.ok = n_gt <= 128;
#pragma unroll
    for (int r = 0; r < 2; ++r) p.gt.key[r] = (p.gt.ok && 64 * r + lane < n_gt) ? S.lm_order[64 * r + lane] : 0;
  }
}

// move + addOdometry + measure(s) + addMeasurement for one belief step, executed by ONE wave (lane = 0..63).
// lds0 / lds1: 626 words each, dyn: (2 LG + 2) doubles + LG ints of LDS scratch.
// kMove / exp_*: the staged interface runs the move (with addOdometry) and a single exporting measure() as separate
// launches (drlgx_stage_move / drlgx_stage_measure); the fused step is <true> with n_measure = 2 and no export.
template <bool kMove = true>
__device__ __forceinline__ void sim_step_body(const DrlgxState &S, const LaunchSel &sel, const double *odom, int odom_stride,
                                              int n_measure, uint32_t *lds0, uint32_t *lds1, double *dyn, int lane,
                                              int32_t *exp_keys = nullptr, double *exp_br = nullptr, int32_t *exp_count = nullptr,
                                              bool park_streams = false, int *mail = nullptr, double *lmbox = nullptr,
                                              int known_P = -1, int known_L = 0, int known_M = 0, const SimPre &pre_in = SimPre{}) {
  uint32_t *lds[2] = {lds0, lds1};
  double *nrm = dyn;
  int *inr = reinterpret_cast<int *>(dyn + 2 * S.LG + 2);
  const int i = drlgx_bid();
  if (!sel.on(i)) return;
  const int inst = sel.base + i;
  const drlgx_config &cfg = S.cfg;
  int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  double ox = 0, oy = 0, oth = 0;
  if constexpr (kMove) {
    const double *od = odom + (size_t)i * odom_stride + (size_t)sel.act_idx * 3;
    ox = od[0]; oy = od[1]; oth = od[2];
    if (!odom_in_bounds(cfg, ox, oy)) {
      if (lane == 0) cnt[C_FLAG] = 1;
      return;
    }
  }
  // (known_*: the counts as the caller read them already - k_step's prelude - so that the loads that depend on the pose
  // count do not wait for a second round trip to the counters)
  SimCtx c{S, inst, lane, {}, {}, {0, 0}, {0, 0}, {}, known_P >= 0 ? known_P : cnt[C_P], known_P >= 0 ? known_L : cnt[C_L],
           known_P >= 0 ? known_M : cnt[C_M], 0};
  if (kMove && c.P >= S.P_max) {
    if (lane == 0) {
      cnt[C_FLAG] = 1;
      atomicMin(S.status, DRLGX_E_CAPACITY);
    }
    return;
  }
  DRLGX_PROF(S, 8);
  // Loads first, in the order of their first use (vmcnt counts in order): what does not depend on the pose count (the caller may
  // have it already: SimPre), then the last pose's estimate.  Nothing is STORED before the simulator's last load has been
  // consumed: the memory counter is decremented in issue order, so a load requested behind a store is not consumed before that
  // store has been acknowledged.  Lane 0's stores of the new pose are issued at the end, with the counts (store_ctx); the new
  // pose's initial guess - formed late, when the estimate it needs has had the draws and the scan to arrive - travels in
  // registers (SimCtx::origin).
  SimPre pre = pre_in;
  if (!pre.have) sim_preload(S, inst, lane, n_measure, pre);
  const double *ep = S.est_pose + ((size_t)inst * S.P_max + (c.P - 1)) * 4;
  const Pose last_est{ep[0], ep[1], ep[2], ep[3]};
  c.veh = pre.veh;
  const int step0 = pre.step0;
  const double dist0 = pre.dist0;
  GtPrefetch gtp = pre.gt;
  c.ns_sensor = NormalState{pre.ns_s, pre.nh_s};
  c.ns_control = NormalState{pre.ns_c, pre.nh_c};
  Pose odomP{0, 0, 1, 0}, p2{0, 0, 1, 0};
  const int P_before = c.P, L_before = c.L;
  double dist_new = 0.0;
  if constexpr (kMove) {
    odomP = make_pose(ox, oy, oth);
    // Planner2D.cpp:1440: dist += sqrt(x^2 + y^2 + angle_weight * theta^2), theta = Pose2::theta() (evaluated here, beside the
    // loads, by every lane; lane 0 stores it with the rest at the end)
    const double th = theta_of(odomP);
    dist_new = dist0 + sqrt(ox * ox + oy * oy + cfg.angle_weight * (th * th));
  }
  mt_load2_commit(lds[0], lds[1], pre.mt, lane, c.sensor, c.control);
  if (n_measure > 0) gt_prefetch_points(c, gtp);  // (the keys arrived with the streams; the points travel during the move)
  DRLGX_PROF(S, 9);
  if constexpr (kMove) {
  // SimpleControlModel::evolve (Simulator2D.cpp:161-182)
  draw_normals(c.control, c.ns_control, 3, nrm, lane);
  const double xn = nrm[0] * cfg.translation_noise + 0.0;
  const double yn = nrm[1] * cfg.translation_noise + 0.0;
  const double tn = nrm[2] * cfg.rotation_noise + 0.0;
  wave_lds_sync();  // (the variates in LDS)
  c.veh = compose(compose(c.veh, odomP), make_pose(xn, yn, tn));
  c.P += 1;
  }
  DRLGX_PROF(S, 10);
  if constexpr (kMove) {
    // SLAM2D::addOdometry (SLAM2D.cpp:70-89): initial guess = last estimate * odom
    p2 = compose(last_est, odomP);
    c.origin = p2;
    c.has_origin = 1;
  }
  if (n_measure > 0) {
    const int n_in = scan_in_range(c, inr, &gtp);
    if (n_measure == 2 && !exp_keys) {
      // SS2D.simulate's two measure() calls (pyss2d.py:171-206): the first one only advances the sensor stream (obstacle
      // logic, inert at safe_distance = 0) - its 2 n_in variates are drawn and dropped together with the second call's
      DRLGX_PROF(S, 11);
      measure(c, true, nrm, inr, n_in, nullptr, nullptr, nullptr, 2 * n_in, lmbox);
      DRLGX_PROF(S, 12);
    } else {
      for (int m = 0; m < n_measure; ++m) {
        measure(c, m == n_measure - 1 && !exp_keys, nrm, inr, n_in, exp_keys, exp_br, exp_count);
        DRLGX_PROF(S, 11 + m);
      }
    }
  }
  if constexpr (kMove) {
    if (lane == 0) {
      double *tp = S.th_pose + ((size_t)inst * S.P_max + P_before) * 4;
      tp[0] = p2.x; tp[1] = p2.y; tp[2] = p2.c; tp[3] = p2.s;
      double *dp = S.d_pose + ((size_t)inst * S.P_max + P_before) * 3;
      dp[0] = dp[1] = dp[2] = 0;
      double *oo = S.odo + ((size_t)inst * S.P_max + (P_before - 1)) * 4;
      oo[0] = odomP.x; oo[1] = odomP.y; oo[2] = odomP.c; oo[3] = odomP.s;
      cnt[C_NEWP] = P_before;
      cnt[C_NEWL] = L_before;
      cnt[C_FLAG] = 0;
      cnt[C_STEP] = step0 + 1;
      S.red[(size_t)inst * DRLGX_RED_STRIDE + R_DIST] = dist_new;
    }
  }
  store_ctx(c, park_streams, mail);
  DRLGX_PROF(S, 13);
}

// ------------------------------------------------------------------------------------------------------------------
// The simulator run AHEAD over a whole action list, and its replay (look-ahead rollouts: EMPlanner2D::simulations_reward,
// Planner2D.cpp:1432-1460).  What Simulator2D::move / measure produce for action a - the noisy ground-truth pose, the noisy
// bearing-range list, which landmarks are first sightings - depends on the ground truth, the two random streams and the
// earlier actions' sightings only, NOT on the SLAM state.  k_presim therefore runs the simulator of a rollout for all its
// actions in one go (one wave per rollout, a dozen rollouts per CU at once) and logs per action what it appends; the belief
// step of action a then REPLAYS that entry (replay_step_body, ~1 us) instead of simulating (8 - 10 us on the critical path of
// every update).  The two values of the append that do depend on the SLAM state - the new pose's initial guess (last estimate
// * odometry, SLAM2D.cpp:70-89) and a first sighting's initial estimate (that guess * measurement, Simulator2D.cpp:95-98) - are
// formed at replay time, with the expressions sim_step_body / measure use.  Same draws in the same order, same expressions:
// the rollouts' beliefs are bit-equal (test_whole_plans_in_one_launch_equal_one_launch_per_action).
// Log entry: int32 hdr[8] = {status (0 rejected: odometry out of bounds, 1 appended, 2 pose capacity), P, L, M after the action,
// new factors, new landmarks, error code, -}, then per new factor {int32 slot, int32 key, double bearing, double range}.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kLogHdr = 32, kLogRec = 24;

__global__ __launch_bounds__(64) void k_presim(DRLGX_KS_PARAM, LaunchSel sel, const double *odom, int odom_stride, int n_measure, int a_end,
                                               unsigned char *simlog, size_t simlog_roll, int simlog_act) {
  const DrlgxState &S = DRLGX_KS_REF;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][DRLGX_MT_STRIDE];
  extern __shared__ double dyn[];  // nrm[2 * n_measure * LG + 2] doubles, inr[LG] ints, kslot[LG] ints
  const int lane = threadIdx.x, i = blockIdx.x;
  if (sel.active && !sel.active[i]) return;
  const int n_mine = sel.n_act ? min(a_end, sel.n_act[i]) : a_end;
  if (n_mine <= sel.act_idx) return;
  const int inst = sel.base + i;
  const drlgx_config &cfg = S.cfg;
  double *nrm = dyn;
  int *inr = reinterpret_cast<int *>(dyn + 2 * n_measure * S.LG + 2);
  int *kslot = inr + S.LG;
  const int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  SimCtx c{S, inst, lane, {}, {}, {0, 0}, {0, 0}, {}, cnt[C_P], cnt[C_L], cnt[C_M], 0};
  c.ns_sensor = NormalState{S.nrm_saved[inst * 2 + 0], S.nrm_has[inst * 2 + 0]};
  c.ns_control = NormalState{S.nrm_saved[inst * 2 + 1], S.nrm_has[inst * 2 + 1]};
  const double *gp = S.gt_pose + (size_t)inst * 4;
  c.veh = Pose{gp[0], gp[1], gp[2], gp[3]};
  mt_load2(lds[0], lds[1], S.mt + (size_t)inst * 2 * DRLGX_MT_STRIDE, lane, c.sensor, c.control);
  for (int k = lane; k < S.LG; k += 64) kslot[k] = S.key_slot[(size_t)inst * S.LG + k];
  wave_lds_sync();
  const double *gl = S.gt_lm + (size_t)S.parent[inst] * S.LG * 2;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int a = sel.act_idx; a < n_mine; ++a) {
    unsigned char *entry = simlog + (size_t)i * simlog_roll + (size_t)a * simlog_act;
    int *hdr = reinterpret_cast<int *>(entry);
    const double *od = odom + (size_t)i * odom_stride + (size_t)a * 3;
    const double ox = od[0], oy = od[1], oth = od[2];
    if (!odom_in_bounds(cfg, ox, oy) || c.P >= S.P_max) {
      if (lane == 0) {
        hdr[0] = odom_in_bounds(cfg, ox, oy) ? 2 : 0;
        hdr[1] = c.P; hdr[2] = c.L; hdr[3] = c.M; hdr[4] = 0; hdr[5] = 0; hdr[6] = 0; hdr[7] = 0;
      }
      continue;
    }
    // SimpleControlModel::evolve (Simulator2D.cpp:161-182): as sim_step_body
    const Pose odomP = make_pose(ox, oy, oth);
    draw_normals(c.control, c.ns_control, 3, nrm, lane);
    const double xn = nrm[0] * cfg.translation_noise + 0.0;
    const double yn = nrm[1] * cfg.translation_noise + 0.0;
    const double tn = nrm[2] * cfg.rotation_noise + 0.0;
    wave_lds_sync();
    c.veh = compose(compose(c.veh, odomP), make_pose(xn, yn, tn));
    c.P += 1;
    int nf = 0, nn = 0, err = 0;
    if (n_measure > 0) {
      const int n_in = scan_in_range(c, inr);
      const int lead = (n_measure - 1) * 2 * n_in;  // SS2D.simulate's first, discarded measure() call: variates drawn and dropped
      draw_normals(c.sensor, c.ns_sensor, lead + 2 * n_in, nrm, lane, lead);
      // Simulator2D::measure + SLAM2D::addMeasurement: the expressions of ksim::measure's recording branch
      for (int base = 0; base < n_in && !err; base += 64) {
        const int k = base + lane;
        const bool v = k < n_in;
        const int key = v ? inr[k] : 0;
        const P2 lm{gl[2 * key], gl[2 * key + 1]};
        const double bn = (v ? nrm[2 * k] : 0.0) * cfg.bearing_noise + 0.0;
        const double rn = (v ? nrm[2 * k + 1] : 0.0) * cfg.range_noise + 0.0;
        const double bearing = bearing_of<false>(c.veh, lm, nullptr, nullptr) + bn;
        const double range = range_of<false>(c.veh, lm, nullptr, nullptr) + rn;
        const bool ok = v && bearing < cfg.max_bearing && bearing > cfg.min_bearing && range < cfg.max_range && range > cfg.min_range;
        int slot = ok ? kslot[key] : 0;
        const bool isnew = ok && slot < 0;
        const unsigned long long okm = __ballot(ok), newm = __ballot(isnew);
        const int n_ok = __popcll(okm), n_new = __popcll(newm);
        if (c.L + n_new > S.L_max || c.M + n_ok > S.M_max) {
          err = DRLGX_E_CAPACITY;
          break;
        }
        if (isnew) {
          slot = c.L + __popcll(newm & below);
          kslot[key] = slot;
        }
        if (ok) {
          unsigned char *rec = entry + kLogHdr + (size_t)(nf + __popcll(okm & below)) * kLogRec;
          reinterpret_cast<int *>(rec)[0] = slot;
          reinterpret_cast<int *>(rec)[1] = key;
          reinterpret_cast<double *>(rec + 8)[0] = bearing;
          reinterpret_cast<double *>(rec + 8)[1] = range;
        }
        c.L += n_new;
        c.M += n_ok;
        nf += n_ok;
        nn += n_new;
        wave_lds_sync();
      }
    }
    if (lane == 0) {
      hdr[0] = 1; hdr[1] = c.P; hdr[2] = c.L; hdr[3] = c.M; hdr[4] = nf; hdr[5] = nn; hdr[6] = err; hdr[7] = 0;
    }
  }
  // What the simulator itself carries - the ground-truth pose, the two random streams, the normal distribution's saved variates -
  // goes back to the rollout as the steps' own simulator would have left it after the last action: a rollout that is read or
  // continued after a replayed look-ahead then holds a ground truth that matches its belief.  (The counts are the replay's to
  // advance, action by action.)
  mt_store(c.sensor, S.mt + ((size_t)inst * 2 + 0) * DRLGX_MT_STRIDE, lane);
  mt_store(c.control, S.mt + ((size_t)inst * 2 + 1) * DRLGX_MT_STRIDE, lane);
  if (lane == 0) {
    S.nrm_saved[inst * 2 + 0] = c.ns_sensor.saved;
    S.nrm_has[inst * 2 + 0] = c.ns_sensor.has;
    S.nrm_saved[inst * 2 + 1] = c.ns_control.saved;
    S.nrm_has[inst * 2 + 1] = c.ns_control.has;
    double *gpo = S.gt_pose + (size_t)inst * 4;
    gpo[0] = c.veh.x; gpo[1] = c.veh.y; gpo[2] = c.veh.c; gpo[3] = c.veh.s;
  }
}

// One logged action of instance blockIdx.x appended to its belief state, by ONE wave: everything sim_step_body + measure write
// for the SLAM stage (the new pose's initial guess, the odometry factor, the bearing-range factors, first sightings, counters, the
// travelled distance) - not the ground truth and the random streams: k_presim left those as they are after the rollout's last action.
__device__ __forceinline__ void replay_step_body(const DrlgxState &S, const LaunchSel &sel, const double *odom, int odom_stride, int lane,
                                                 int *mail = nullptr) {
  const int i = drlgx_bid();
  if (!sel.on(i)) return;
  const int inst = sel.base + i;
  const drlgx_config &cfg = S.cfg;
  int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  const unsigned char *entry = sel.simlog + (size_t)i * sel.simlog_roll + (size_t)sel.act_idx * sel.simlog_act;
  const int *hdr = reinterpret_cast<const int *>(entry);
  const int status = hdr[0];
  if (status != 1) {  // rejected move: nothing appended (sim_step_body's two early returns)
    if (lane == 0) {
      cnt[C_FLAG] = 1;
      if (status == 2) atomicMin(S.status, DRLGX_E_CAPACITY);
    }
    return;
  }
  const int P0 = cnt[C_P], L0 = cnt[C_L], M0 = cnt[C_M], step0 = cnt[C_STEP];
  const double *od = odom + (size_t)i * odom_stride + (size_t)sel.act_idx * 3;
  const double ox = od[0], oy = od[1], oth = od[2];
  const double *ep = S.est_pose + ((size_t)inst * S.P_max + (P0 - 1)) * 4;
  const Pose last_est{ep[0], ep[1], ep[2], ep[3]};
  const double dist0 = S.red[(size_t)inst * DRLGX_RED_STRIDE + R_DIST];
  const Pose odomP = make_pose(ox, oy, oth);
  const Pose p2 = compose(last_est, odomP);  // SLAM2D::addOdometry (SLAM2D.cpp:70-89): initial guess = last estimate * odom
  if (lane == 0) {
    double *tp = S.th_pose + ((size_t)inst * S.P_max + P0) * 4;
    tp[0] = p2.x; tp[1] = p2.y; tp[2] = p2.c; tp[3] = p2.s;
    double *dp = S.d_pose + ((size_t)inst * S.P_max + P0) * 3;
    dp[0] = dp[1] = dp[2] = 0;
    double *oo = S.odo + ((size_t)inst * S.P_max + (P0 - 1)) * 4;
    oo[0] = odomP.x; oo[1] = odomP.y; oo[2] = odomP.c; oo[3] = odomP.s;
    cnt[C_NEWP] = P0;
    cnt[C_NEWL] = L0;
    cnt[C_FLAG] = 0;
    cnt[C_STEP] = step0 + 1;
    const double th = theta_of(odomP);
    S.red[(size_t)inst * DRLGX_RED_STRIDE + R_DIST] = dist0 + sqrt(ox * ox + oy * oy + cfg.angle_weight * (th * th));
  }
  const int nf = hdr[4];
  for (int f = lane; f < nf; f += 64) {
    const unsigned char *rec = entry + kLogHdr + (size_t)f * kLogRec;
    const int slot = reinterpret_cast<const int *>(rec)[0], key = reinterpret_cast<const int *>(rec)[1];
    const double bearing = reinterpret_cast<const double *>(rec + 8)[0], range = reinterpret_cast<const double *>(rec + 8)[1];
    if (slot >= L0) {  // first sighting: origin = initial estimate of the measuring pose (Simulator2D.cpp:95-98)
      const P2 g = transform_from(p2, P2{range * cos(bearing), range * sin(bearing)});
      double *tl = S.th_lm + ((size_t)inst * S.L_max + slot) * 2;
      tl[0] = g.x;
      tl[1] = g.y;
      double *dl = S.d_lm + ((size_t)inst * S.L_max + slot) * 2;
      dl[0] = 0;
      dl[1] = 0;
      S.lm_key[(size_t)inst * S.L_max + slot] = key;
      S.key_slot[(size_t)inst * S.LG + key] = slot;
    }
    S.meas_pose[(size_t)inst * S.M_max + M0 + f] = P0;
    S.meas_lm[(size_t)inst * S.M_max + M0 + f] = slot;
    double *br = S.meas_br + ((size_t)inst * S.M_max + M0 + f) * 2;
    br[0] = bearing;
    br[1] = range;
  }
  if (lane == 0) {
    cnt[C_P] = hdr[1];
    cnt[C_L] = hdr[2];
    cnt[C_M] = hdr[3];
    if (mail) {
      mail[0] = hdr[1];
      mail[1] = hdr[2];
      mail[2] = hdr[3];
    }
    if (hdr[6]) atomicMin(S.status, hdr[6]);
  }
}

__global__ __launch_bounds__(64) void k_sim_step(DRLGX_KS_PARAM, LaunchSel sel, const double *odom, int odom_stride,
                                                 int n_measure) {
  const DrlgxState &S = DRLGX_KS_REF;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][DRLGX_MT_STRIDE];
  extern __shared__ double dyn[];  // nrm[2 LG + 2] doubles, inr[LG] ints
  sim_step_body(S, sel, odom, odom_stride, n_measure, lds[0], lds[1], dyn, threadIdx.x);
}

// drlgx_stage_move (mode 0): Simulator2D::move + SLAM2D::addOdometry.  drlgx_stage_measure (mode 1): one
// Simulator2D::measure whose valid measurements are exported: keys [n][LG], br [n][LG][2], count [n].
__global__ __launch_bounds__(64) void k_sim_stage(DRLGX_KS_PARAM, LaunchSel sel, const double *odom, int mode, int32_t *keys,
                                                  double *br, int32_t *count) {
  const DrlgxState &S = DRLGX_KS_REF;
  __shared__ __attribute__((aligned(16))) uint32_t lds[2][DRLGX_MT_STRIDE];
  extern __shared__ double dyn[];
  const int i = blockIdx.x;
  if (mode == 0)
    sim_step_body<true>(S, sel, odom, 3, 0, lds[0], lds[1], dyn, threadIdx.x);
  else
    sim_step_body<false>(S, sel, nullptr, 0, 1, lds[0], lds[1], dyn, threadIdx.x, keys + (size_t)i * S.LG, br + (size_t)i * S.LG * 2,
                         count + i);
}

// drlgx_stage_add_measurements: SLAM2D::addMeasurement (SLAM2D.cpp:103-124) of every listed (key, bearing, range) at the
// newest pose, in list order (a key may repeat: the second one is no longer new).  Not a hot path: one lane per instance.
__global__ void k_add_measurements(DRLGX_KS_PARAM, LaunchSel sel, const int32_t *keys, const double *br, const int32_t *count) {
  const DrlgxState &S = DRLGX_KS_REF;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sel.n || !sel.on(i)) return;
  const int inst = sel.base + i;
  int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  int P = cnt[C_P], L = cnt[C_L], M = cnt[C_M];
  int *key_slot = S.key_slot + (size_t)inst * S.LG;
  if (count[i] < 0 || count[i] > S.LG) {  // the lists have LG entries per instance
    atomicMin(S.status, DRLGX_E_INVALID);
    return;
  }
  for (int k = 0; k < count[i]; ++k) {
    const int key = keys[(size_t)i * S.LG + k];
    const double bearing = br[((size_t)i * S.LG + k) * 2], range = br[((size_t)i * S.LG + k) * 2 + 1];
    if (key < 0 || key >= S.cfg.num_landmarks) {
      atomicMin(S.status, DRLGX_E_INVALID);
      break;
    }
    int slot = key_slot[key];
    if (slot < 0) {
      if (L >= S.L_max) {
        atomicMin(S.status, DRLGX_E_CAPACITY);
        break;
      }
      slot = L++;
      const double *tp = S.th_pose + ((size_t)inst * S.P_max + (P - 1)) * 4;
      const P2 g = transform_from(Pose{tp[0], tp[1], tp[2], tp[3]}, P2{range * cos(bearing), range * sin(bearing)});
      double *tl = S.th_lm + ((size_t)inst * S.L_max + slot) * 2;
      tl[0] = g.x; tl[1] = g.y;
      double *dl = S.d_lm + ((size_t)inst * S.L_max + slot) * 2;
      dl[0] = 0; dl[1] = 0;
      S.lm_key[(size_t)inst * S.L_max + slot] = key;
      key_slot[key] = slot;
    }
    if (M >= S.M_max) {
      atomicMin(S.status, DRLGX_E_CAPACITY);
      break;
    }
    S.meas_pose[(size_t)inst * S.M_max + M] = P - 1;
    S.meas_lm[(size_t)inst * S.M_max + M] = slot;
    S.meas_br[((size_t)inst * S.M_max + M) * 2] = bearing;
    S.meas_br[((size_t)inst * S.M_max + M) * 2 + 1] = range;
    ++M;
  }
  cnt[C_L] = L;
  cnt[C_M] = M;
}

}  // namespace ksim

size_t drlgx_simlog_entry_bytes(const DrlgxState &S) { return ((size_t)ksim::kLogHdr + (size_t)ksim::kLogRec * S.LG + 31) & ~(size_t)31; }
void drlgx_launch_presim(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure, int a_end,
                         unsigned char *simlog, size_t simlog_roll, int simlog_act) {
  const size_t dyn = (size_t)(2 * (n_measure > 0 ? n_measure : 1) * S.LG + 2) * sizeof(double) + (size_t)2 * S.LG * sizeof(int);
  hipLaunchKernelGGL(ksim::k_presim, dim3(sel.n), dim3(64), dyn, st, DRLGX_KS_ARG(S), sel, odom, odom_stride, n_measure, a_end, simlog, simlog_roll, simlog_act);
}
void drlgx_launch_sim_stage(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int mode, int32_t *keys,
                            double *br, int32_t *count) {
  const size_t dyn = (size_t)(2 * S.LG + 2) * sizeof(double) + (size_t)S.LG * sizeof(int);
  hipLaunchKernelGGL(ksim::k_sim_stage, dim3(sel.n), dim3(64), dyn, st, DRLGX_KS_ARG(S), sel, odom, mode, keys, br, count);
}
void drlgx_launch_add_measurements(const DrlgxState &S, hipStream_t st, LaunchSel sel, const int32_t *keys, const double *br,
                                   const int32_t *count) {
  hipLaunchKernelGGL(ksim::k_add_measurements, dim3((sel.n + 63) / 64), dim3(64), 0, st, DRLGX_KS_ARG(S), sel, keys, br, count);
}
void drlgx_launch_reset(const DrlgxState &S, hipStream_t st, int n, const int32_t *env_ids_dev,
                        const uint32_t *seeds_dev, const double *start_dev, int first_measure) {
  const size_t dyn = (size_t)(2 * S.LG + 2) * sizeof(double) + (size_t)S.LG * sizeof(int);
  hipLaunchKernelGGL(ksim::k_reset, dim3(n), dim3(64), dyn, st, DRLGX_KS_ARG(S), first_measure, env_ids_dev, seeds_dev, start_dev);
}
void drlgx_launch_sim(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride,
                      int n_measure) {
  const size_t dyn = (size_t)(2 * S.LG + 2) * sizeof(double) + (size_t)S.LG * sizeof(int);
  hipLaunchKernelGGL(ksim::k_sim_step, dim3(sel.n), dim3(64), dyn, st, DRLGX_KS_ARG(S), sel, odom, odom_stride, n_measure);
}
