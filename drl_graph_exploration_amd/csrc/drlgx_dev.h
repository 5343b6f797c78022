// drlgx device-side common definitions: HBM state layout, Pose2 algebra, small SPD solves,
// mt19937 / normal_distribution streams.  gfx950 only.
//
// Reference semantics (paths under /root/reference): gtsam Pose2/Rot2 (third-party, absent —
// SURVEY.md App. A.1), include/em_exploration/RNG.h:47-126 (libstdc++ mt19937 +
// uniform_real_distribution + ONE shared normal_distribution), include/em_exploration/Utils.h:29-33.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/drlgx.h"

#define DRLGX_LO_TAB 64
#define DRLGX_MT_N 624
#define DRLGX_MT_STRIDE 626  // 624 state words + gen index + cons index
#define DRLGX_CNT_STRIDE 8
#define DRLGX_RED_STRIDE 8
#define DRLGX_PRIOR_STRIDE 16

// cnt[] slots
enum { C_P = 0, C_L = 1, C_M = 2, C_STEP = 3, C_ISAM = 4, C_NEWP = 5, C_NEWL = 6, C_FLAG = 7 };
// red[] slots (outputs of the map kernel)
enum { R_UTR = 0, R_KNOWN = 1, R_EXPL = 2, R_UDET = 3, R_UWTR = 4, R_DIST = 5 };

// All per-instance arrays live in ONE HBM allocation; instance-major, contiguous per instance so a
// workgroup streams its own slice with coalesced accesses.
struct DrlgxState {
  drlgx_config cfg;
  int n_envs, n_inst, n_roll;  // instances: [0,n_envs) live, [n_envs, 2 n_envs) look-ahead bases, then rollouts
  int P_max, L_max, M_max, LG, V, Vu, rows, cols, A_max;  // Vu = V rounded up to 4 (byte plane stride)
  int win;  // (win x win) candidate window of cells around a pose = ceil(2*max_range/res)+1
  int count_explored;
  double lo_free, lo_occ, lo_min, lo_max, occ_thresh;  // log-odds constants computed on the HOST (bit-exact ladder)
  int bbox_noop;   // 1: the sector-sweep bounding box provably contains every in-range cell (full-circle FOV)
  int fov_fast;    // 1: cells with d.x >= 0 or |d.y| > fov_tan |d.x| are provably inside the field of view
  double fov_tan;  // tan(blind half-angle + 1 mrad)
  // exact squared-distance thresholds (host-computed for IEEE sqrt):
  //   sqrt(x) < max_range  <=>  x < r2_max_lt ;   sqrt(x) > min_range  <=>  x > r2_min_gt
  double r2_max_lt, r2_min_gt;
  int n_sweep;                                         // bbox sweep table length
  const double *sweep_b;                               // [n_sweep] b values of OccupancyMap.cpp:86
  // occupancy ladder as a finite state machine (closure of l -> clamp(l + lo_occ / lo_free) from l = 0, built on the
  // host): state 0 is LOGODDS_UNKNOWN; lo_tr[4 i] = {next on occupied, next on free, flags (1: at the minimum - frozen,
  // 2: above the occupancy threshold), 0}; lo_pv[i] = cell probability of state i (host libm).  lo_ntab == 0: not closed
  // within DRLGX_LO_TAB states, the kernels fall back to the arithmetic ladder.
  int lo_ntab;
  const double *lo_pv;
  const uint8_t *lo_tr;
  unsigned long long lo_tocc, lo_tfree;  // lo_tr packed for <= 16 states: 4 bits per next state (occupied / free)
  unsigned int lo_tflag;                 // 2 bits of flags per state
  double vm_i0;                          // 1 / sigma0^2 (host pow, like the reference's initialisation)
  double w_trans, w_rot, w_bear, w_range;  // 1 / noise^2 of the odometry (translation, rotation) and bearing-range factors
  const int *lm_order;                                 // [LG] libstdc++ unordered_map iteration order of GT keys
  // --- simulator
  double *gt_pose;    // [n_inst][4] x,y,c,s
  double *gt_lm;      // [n_envs][LG][2]
  const double *fixed_lm;  // [n_fixed][2] listed landmarks (keys 0 .. n_fixed - 1 of every env; drlgx_set_fixed_landmarks_host) or null
  int n_fixed;
  int *parent;        // [n_inst] env that owns the ground-truth landmarks
  uint32_t *mt;       // [n_inst][2][MT_STRIDE]  0 = sensor stream, 1 = control stream
  double *nrm_saved;  // [n_inst][2]
  int *nrm_has;       // [n_inst][2]
  // --- isam (linearisation point, delta, factors)
  int *cnt;          // [n_inst][CNT_STRIDE]
  double *th_pose;   // [n_inst][P_max][4]
  double *d_pose;    // [n_inst][P_max][3]
  double *th_lm;     // [n_inst][L_max][2]
  double *d_lm;      // [n_inst][L_max][2]
  int *lm_key;       // [n_inst][L_max]
  int *key_slot;     // [n_inst][LG]
  double *prior;     // [n_inst][16]: pose(4) + information(9)
  double *odo;       // [n_inst][P_max][4] measured odometry between pose i and i+1 (x,y,c,s)
  int *meas_pose;    // [n_inst][M_max]
  int *meas_lm;      // [n_inst][M_max]
  double *meas_br;   // [n_inst][M_max][2]
  // --- results of the last optimise (SLAM2D::result_ and Map)
  double *est_pose;   // [n_inst][P_max][4]
  double *est_lm;     // [n_inst][L_max][2]
  double *pose_info;  // [n_inst][P_max][6]  symmetric 3x3: xx xy xt yy yt tt
  double *lm_info;    // [n_inst][L_max][3]  symmetric 2x2: xx xy yy
  double *pose_tr;    // [n_inst][P_max]  trace of marginal covariance
  double *lm_tr;      // [n_inst][L_max]
  // --- virtual map
  double *vm_prob;  // [n_inst][V]
  double *vm_info;  // [n_inst][3][V]  planes xx, xy, yy
  uint8_t *vm_upd;  // [n_inst][Vu]
  double *vm_tr;    // [n_inst][V]  trace of covariance per cell (VirtualMap::toCovTrace)
  double *red;      // [n_inst][RED_STRIDE]
  // --- scratch
  double *slam_ws;  // [n_inst][slam_ws_stride] dense Schur system when it does not fit LDS
  size_t slam_ws_stride;
  int *slam_iws;  // [n_inst][slam_iws_stride] observation table + per-pose factor ranges
  size_t slam_iws_stride;
  int *status;  // [1]
  // --- incremental belief update (k_inc.hip): the columns of the joint covariance that later updates can touch - every
  // variable against the ACTIVE set (current pose, landmarks) - plus the 3x3 marginal of every pose.  Null: disabled.
  //   jc[inst][row][jc_ld]: rows 3 i + r = pose i, 3 P_max + 2 j + e = landmark j; columns 0..2 = the current (newest) pose,
  //   3 + 2 j + e = landmark j.  jd[inst][P_max][6]: pose marginal covariance (xx yx yy tx ty tt).
  //   jc_meta[inst][4] = {valid, P, L, M at the update that left the panel}.  inc_stats[2] = {incremental, full} updates.
  double *jc;
  size_t jc_stride;
  int jc_ld;
  double *jd;
  int *jc_meta;
  unsigned long long *inc_stats;
  long long *prof;  // [1024] development aid: wall_clock64() stamps of the phases of ONE workgroup (or null)
  int prof_block;   // ... this one
  // This struct in device memory (the engine keeps it current; null: no copy).  A kernel that takes the struct BY VALUE reads
  // every member it uses from the kernel-argument segment before its first instruction - for the fused step that is most of the
  // ~0.8 KB, a dozen cache lines of memory nobody has touched before, and it cost 2.6 us per launch against reading the members
  // where they are needed from this copy, which stays in L2 from launch to launch (k_step_ref, profiles/r05_ab_state_pointer.txt).
  const DrlgxState *self_dev;
};

// The device-resident copy as a kernel parameter: a pointer into the CONSTANT address space.  The members are then read with scalar
// loads where they are used, and - what matters more - the pointers among them are known to point to global memory: loaded through
// a plain (flat / global) pointer every access of the kernel became a flat_load / flat_store with a 64-bit address pair per access
// (256 VGPRs + 108 B of scratch per thread in round 5, and every LDS wait also waited for the memory accesses in flight).
typedef const __attribute__((address_space(4))) DrlgxState *DrlgxStateConst;
// The state struct's cache lines through the scalar cache once, at the head of the fused step (one wave: the cache is the CU's): the
// members are read where they are used, and every FIRST touch of a line is otherwise an L2 round trip (~0.3 us) somewhere along the
// workgroup's critical path - a dozen of them between the prelude and the map stage's constants.  One dword per 64-byte line; the wait is
// the price of naming destination registers.  -0.3 ... -0.7 us per fused step over eight interleaved same-box pairs
// (profiles/r06_ab_state_const.txt).
__device__ __forceinline__ void drlgx_warm_state(DrlgxStateConst Sp) {
  static_assert(sizeof(DrlgxState) <= 14 * 64, "drlgx_warm_state touches fourteen lines");
  const unsigned long long sp = reinterpret_cast<unsigned long long>(Sp);
  unsigned int w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13;
  asm volatile("s_load_dword %0, %14, 0x0\n\ts_load_dword %1, %14, 0x40\n\ts_load_dword %2, %14, 0x80\n\t"
               "s_load_dword %3, %14, 0xc0\n\ts_load_dword %4, %14, 0x100\n\ts_load_dword %5, %14, 0x140\n\t"
               "s_load_dword %6, %14, 0x180\n\ts_load_dword %7, %14, 0x1c0\n\ts_load_dword %8, %14, 0x200\n\t"
               "s_load_dword %9, %14, 0x240\n\ts_load_dword %10, %14, 0x280\n\ts_load_dword %11, %14, 0x2c0\n\t"
               "s_load_dword %12, %14, 0x300\n\ts_load_dword %13, %14, 0x340\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=s"(w0), "=s"(w1), "=s"(w2), "=s"(w3), "=s"(w4), "=s"(w5), "=s"(w6), "=s"(w7), "=s"(w8), "=s"(w9), "=s"(w10), "=s"(w11),
                 "=s"(w12), "=s"(w13)
               : "s"(sp));
}

// How the belief kernels of the unity build receive the state: DRLGX_KS_PARAM in the signature, `const DrlgxState &S = DRLGX_KS_REF;`
// as the first line, DRLGX_KS_ARG(S) at the launch.  -DDRLGX_STATE_BY_VALUE builds the by-value form (the A/B of profiles/r06_ab_state_const.txt).
#ifdef DRLGX_STATE_BY_VALUE
#define DRLGX_KS_PARAM DrlgxState S_
#define DRLGX_KS_REF S_
#define DRLGX_KS_ARG(S) (S)
#else
#define DRLGX_KS_PARAM DrlgxStateConst S_
#define DRLGX_KS_REF (*(const DrlgxState *)S_)
#define DRLGX_KS_ARG(S) ((DrlgxStateConst)(S).self_dev)
#endif

// phase stamp (100 MHz constant clock) — only block 0 / thread 0, only when profiling is armed
#define DRLGX_PROF(S, slot)                                                        \
  do {                                                                             \
    if ((S).prof && threadIdx.x == 0 && blockIdx.x == (S).prof_block) (S).prof[slot] = wall_clock64(); \
  } while (0)

// The workgroup's index as a value the optimiser cannot look through.  The stage bodies derive every per-instance address from
// it; inside k_step_loop (one workgroup runs a whole action list) a plain blockIdx.x let loop-invariant code motion hoist those
// addresses - dozens of 64-bit pointers per thread - out of the action loop, and the body spilled 2 KB per thread.
__device__ __forceinline__ int drlgx_bid() {
  int b = blockIdx.x;
  asm volatile("" : "+v"(b));                 // (opaque, and not movable: volatile)
  return __builtin_amdgcn_readfirstlane(b);  // ... and uniform again: a scalar register for the selectors and the scalar loads
}

// ... and the thread index likewise (the stage bodies keep dozens of values derived from it - lane constants, tile offsets -
// which the same motion would otherwise carry across the whole action loop)
__device__ __forceinline__ int drlgx_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

struct LaunchSel {
  int base;                // first instance
  int n;                   // number of instances (= grid size)
  const uint8_t *active;   // [n] or null
  const int32_t *n_act;    // [n] or null: instance active iff act_idx < n_act[i]
  int act_idx;
  // look-ahead rollouts: the virtual map is a pure function of the SLAM state (both rebuilds start from the untouched
  // map), and only the utility after the LAST action enters the reward, so the map stage runs for an instance only
  // when act_idx is its last action
  int map_last_only = 0;
  // Host-side upper bound of the pose count of every selected instance after this launch (0: the capacity S.P_max).  The
  // kernels size their per-pose LDS tables with it, so that a large pose capacity costs short trajectories nothing; an
  // instance that exceeds it raises DRLGX_E_CAPACITY instead of overrunning.
  int pcap = 0;
  // the fused step kernel leaves the map stage out: the caller launches the stand-alone map kernel behind it (more instances
  // than CUs: its two-workgroups-per-CU form beats the stage fused into a one-per-CU workgroup)
  int skip_map = 0;
  // Rollouts whose simulator was run AHEAD for the whole action list (ksim::k_presim): the log of what every action's move and
  // measurements appended; wave 0 of the step then replays one entry (ksim::replay_step_body) instead of simulating - the draws
  // of an action depend on the ground truth and the random streams only, so they need not sit on every update's critical path.
  // Entry of (instance i, action a) at simlog + i * simlog_roll + a * simlog_act.
  const unsigned char *simlog = nullptr;
  size_t simlog_roll = 0;
  int simlog_act = 0;
  __host__ __device__ __forceinline__ int cap(int P_max) const { return pcap > 0 && pcap < P_max ? pcap : P_max; }
  __device__ __forceinline__ bool map_on(int i) const {
    return !(map_last_only && n_act) || act_idx == n_act[i] - 1;
  }
  __device__ __forceinline__ bool on(int i) const {
    if (active && !active[i]) return false;
    if (n_act && act_idx >= n_act[i]) return false;
    return true;
  }
};

// ------------------------------------------------------------------------------------------------
// Pose2 algebra (x, y, cos, sin)
// ------------------------------------------------------------------------------------------------
struct Pose {
  double x, y, c, s;
};
struct P2 {
  double x, y;
};

__device__ __forceinline__ void rot_from_cos_sin(double c, double s, double &oc, double &os) {
  if (fabs(c * c + s * s - 1.0) > 1e-9) {
    double n = sqrt(c * c + s * s);
    c /= n;
    s /= n;
  }
  oc = c;
  os = s;
}
__device__ __forceinline__ Pose make_pose(double x, double y, double th) { return Pose{x, y, cos(th), sin(th)}; }
__device__ __forceinline__ double theta_of(const Pose &p) { return atan2(p.s, p.c); }
__device__ __forceinline__ Pose compose(const Pose &a, const Pose &b) {
  Pose r;
  rot_from_cos_sin(a.c * b.c - a.s * b.s, a.s * b.c + a.c * b.s, r.c, r.s);
  r.x = a.x + (a.c * b.x - a.s * b.y);
  r.y = a.y + (a.s * b.x + a.c * b.y);
  return r;
}
__device__ __forceinline__ Pose between(const Pose &p1, const Pose &p2, double *H1) {
  double c = p1.c * p2.c + p1.s * p2.s, s = -p1.s * p2.c + p1.c * p2.s;
  Pose r;
  rot_from_cos_sin(c, s, r.c, r.s);
  double dx = p2.x - p1.x, dy = p2.y - p1.y;
  r.x = p1.c * dx + p1.s * dy;
  r.y = -p1.s * dx + p1.c * dy;
  if (H1) {
    double dt1 = -p2.s * dx + p2.c * dy;
    double dt2 = -p2.c * dx - p2.s * dy;
    H1[0] = -r.c; H1[1] = -r.s; H1[2] = dt1;
    H1[3] = r.s;  H1[4] = -r.c; H1[5] = dt2;
    H1[6] = 0;    H1[7] = 0;    H1[8] = -1;
  }
  return r;
}
__device__ __forceinline__ P2 transform_to(const Pose &p, const P2 &pt) {
  double dx = pt.x - p.x, dy = pt.y - p.y;
  return P2{p.c * dx + p.s * dy, -p.s * dx + p.c * dy};
}
__device__ __forceinline__ P2 transform_from(const Pose &p, const P2 &q) {
  return P2{p.c * q.x - p.s * q.y + p.x, p.s * q.x + p.c * q.y + p.y};
}
template <bool JAC>
__device__ __forceinline__ double bearing_of(const Pose &p, const P2 &pt, double *Hx, double *Hl) {
  P2 d = transform_to(p, pt);
  double d2 = d.x * d.x + d.y * d.y, n = sqrt(d2);
  if (fabs(n) > 1e-5) {
    if (JAC) {
      double a = -d.y / d2, b = d.x / d2;
      Hx[0] = a * -1.0;
      Hx[1] = b * -1.0;
      Hx[2] = a * d.y + b * -d.x;
      Hl[0] = a * p.c + b * -p.s;
      Hl[1] = a * p.s + b * p.c;
    }
    double c, s;
    rot_from_cos_sin(d.x / n, d.y / n, c, s);
    return atan2(s, c);
  }
  if (JAC) {
    Hx[0] = Hx[1] = Hx[2] = 0;
    Hl[0] = Hl[1] = 0;
  }
  return 0.0;
}
template <bool JAC>
__device__ __forceinline__ double range_of(const Pose &p, const P2 &pt, double *Hx, double *Hl) {
  double dx = pt.x - p.x, dy = pt.y - p.y;
  double r = sqrt(dx * dx + dy * dy);
  if (JAC) {
    double ux = dx / r, uy = dy / r;
    Hx[0] = ux * -p.c + uy * -p.s;
    Hx[1] = ux * p.s + uy * -p.c;
    Hx[2] = 0;
    Hl[0] = ux;
    Hl[1] = uy;
  }
  return r;
}

// ------------------------------------------------------------------------------------------------
// small SPD helpers — symmetric storage: 3x3 = (a00,a01,a02,a11,a12,a22); 2x2 = (a00,a01,a11)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double det2s(double a, double b, double d) { return a * d - b * b; }
// 1/x and 1/sqrt(x) to double round-off (v_rcp_f64 / v_rsq_f64 + two Newton steps): a correctly rounded fp64 division or
// sqrt is ~30 instructions here, and these per-cell / per-(pose, cell) kernels are bound by instruction issue.  Used only
// where the parity bar is a tolerance (information / covariance values), never where a decision is taken.
__device__ __forceinline__ double rcp_n(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}
// one Newton step on v_rcp_f64 / v_rsq_f64 (raw relative error 5e-8 -> 2e-15 / 4e-15, scripts/micro/rcp_precision.hip):
// for algebra whose results are only compared against a tolerance
__device__ __forceinline__ double rcp_n1(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return r * (2.0 - x * r);
}
__device__ __forceinline__ double rsqrt_n1(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  return r * (1.5 - 0.5 * x * r * r);
}
__device__ __forceinline__ double rsqrt_n(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}
// inv2_llt_s with reciprocals (same algebra, each division / sqrt replaced by a ~1 ulp reciprocal product)
__device__ __forceinline__ void inv2_llt_fast(double a, double b, double d, double &oa, double &ob, double &od) {
  const double r00 = rsqrt_n(a);         // 1 / l00
  const double l10 = b * r00;
  const double r11 = rsqrt_n(d - l10 * l10);  // 1 / l11
  const double x1 = (0.0 - l10 * r00) * r11 * r11;
  oa = (r00 - l10 * x1) * r00;
  ob = x1;
  od = r11 * r11;
}
__device__ __forceinline__ void inv2_llt_s(double a, double b, double d, double &oa, double &ob, double &od) {
  // m.llt().solve(I) for symmetric [[a,b],[b,d]]
  double l00 = sqrt(a), l10 = b / l00, l11 = sqrt(d - l10 * l10);
  // column 0
  double y0 = 1.0 / l00, y1 = (0.0 - l10 * y0) / l11;
  double x1 = y1 / l11, x0 = (y0 - l10 * x1) / l00;
  oa = x0;
  ob = x1;
  // column 1
  y0 = 0.0 / l00;
  y1 = (1.0 - l10 * y0) / l11;
  x1 = y1 / l11;
  od = x1;
}
struct LLT3 {
  double l00, l10, l11, l20, l21, l22;
  __device__ __forceinline__ LLT3(double a00, double a01, double a02, double a11, double a12, double a22) {
    l00 = sqrt(a00);
    l10 = a01 / l00;
    l20 = a02 / l00;
    l11 = sqrt(a11 - l10 * l10);
    l21 = (a12 - l20 * l10) / l11;
    l22 = sqrt(a22 - l20 * l20 - l21 * l21);
  }
  __device__ __forceinline__ void solve(double b0, double b1, double b2, double &x0, double &x1, double &x2) const {
    double y0 = b0 / l00, y1 = (b1 - l10 * y0) / l11, y2 = (b2 - l20 * y0 - l21 * y1) / l22;
    x2 = y2 / l22;
    x1 = (y1 - l21 * x2) / l11;
    x0 = (y0 - l10 * x1 - l20 * x2) / l00;
  }
};
// inverse of a symmetric positive definite 3x3 (c00 c10 c20 c11 c21 c22) by cofactors and ONE reciprocal, in the storage
// order of the pose information (i00 i10 i20 i11 i21 i22).  Used where the reference inverts a marginal covariance by LLT
// (SLAM2D.cpp:395-408): the values are compared at 1e-7, no decision depends on them.
__device__ __forceinline__ void inv3_sym_fast(double c00, double c10, double c20, double c11, double c21, double c22, double *o) {
  const double k00 = c11 * c22 - c21 * c21, k10 = c20 * c21 - c10 * c22, k20 = c10 * c21 - c20 * c11;
  const double id = rcp_n(c00 * k00 + c10 * k10 + c20 * k20);
  o[0] = k00 * id;
  o[1] = k10 * id;
  o[2] = k20 * id;
  o[3] = (c00 * c22 - c20 * c20) * id;
  o[4] = (c10 * c20 - c00 * c21) * id;
  o[5] = (c00 * c11 - c10 * c10) * id;
}
__device__ __forceinline__ double det3s(double a00, double a01, double a02, double a11, double a12, double a22) {
  return a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
}

// ------------------------------------------------------------------------------------------------
// mt19937 stream held in LDS by ONE wave: state[624] + gen (how far the lazy recurrence has been
// applied) + cons (next word to hand out).  The recurrence is applied 64 words at a time by the
// whole wave; consumption is wave-uniform (every lane computes the same scalars).
// Identical output sequence to libstdc++ std::mt19937 (bits/random.tcc).
// ------------------------------------------------------------------------------------------------
struct MtStream {
  uint32_t *st;        // LDS, 624 state words (+2 words used only for load/store of gen/cons)
  uint32_t gen, cons;  // wave-uniform registers: words generated / consumed so far (monotonic)
};

// one-wave barrier that also orders LDS accesses
// The simulator code runs in ONE wave (alone in its workgroup, or as wave 0 of the fused step kernel): ordering of that
// wave's LDS / global accesses needs fences and a wave barrier, not a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// ... and the one for LDS traffic INSIDE the wave only (a lane reads what another lane of the same wave wrote to LDS): the
// wave's LDS operations execute in order, so nothing has to drain - the workgroup-scope fences above also wait for every
// global load and STORE in flight (vmcnt(0)): in the simulator wave that was a memory round trip per refill of the generator
// and per commit that followed a global access.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// load a stream from HBM into LDS (64 lanes) and back
__device__ inline MtStream mt_load(uint32_t *lds, const uint32_t *g, int lane) {
  for (int i = lane; i < DRLGX_MT_STRIDE; i += 64) lds[i] = g[i];
  wave_lds_sync();
  MtStream s;
  s.st = lds;
  s.gen = lds[DRLGX_MT_N];
  s.cons = lds[DRLGX_MT_N + 1];
  return s;
}
// both streams of an instance (contiguous in HBM, 2 x 626 words) in one go: every load is issued before the first wait.
// Two halves, so that the caller can put work that does not need the streams between the issue and the first use.
__device__ inline void mt_load2_issue(const uint32_t *g, int lane, uint4 (&v)[5]) {
  const uint4 *g4 = reinterpret_cast<const uint4 *>(g);  // 2 * 626 words = 313 uint4, 16-byte aligned per instance
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = lane + 64 * k;
    v[k] = (i < 313) ? g4[i] : make_uint4(0, 0, 0, 0);
  }
}
__device__ inline void mt_load2_commit(uint32_t *lds0, uint32_t *lds1, const uint4 (&v)[5], int lane, MtStream &a, MtStream &b) {
  // (the two images are contiguous in LDS - lds1 == lds0 + DRLGX_MT_STRIDE, 16-byte aligned, every caller - like the two streams
  // in HBM: 313 whole 16-byte pieces, five stores per lane; word by word with a range test each this was ~80 branches)
  uint4 *l4 = reinterpret_cast<uint4 *>(lds0);
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = lane + 64 * k;
    if (i < 313) l4[i] = v[k];
  }
  wave_lds_sync();
  a.st = lds0; a.gen = lds0[DRLGX_MT_N]; a.cons = lds0[DRLGX_MT_N + 1];
  b.st = lds1; b.gen = lds1[DRLGX_MT_N]; b.cons = lds1[DRLGX_MT_N + 1];
}
__device__ inline void mt_load2(uint32_t *lds0, uint32_t *lds1, const uint32_t *g, int lane, MtStream &a, MtStream &b) {
  uint4 v[5];
  mt_load2_issue(g, lane, v);
  mt_load2_commit(lds0, lds1, v, lane, a, b);
}
// the stream's counters into the two spare words of its LDS image (as mt_store writes them): the image can then be copied
// to HBM by anyone, 626 words per stream
__device__ inline void mt_park(const MtStream &s, int lane) {
  if (lane == 0) {
    const uint32_t k = (s.cons / DRLGX_MT_N) * DRLGX_MT_N;
    s.st[DRLGX_MT_N] = s.gen - k;
    s.st[DRLGX_MT_N + 1] = s.cons - k;
  }
}
__device__ inline void mt_store(const MtStream &s, uint32_t *g, int lane) {
  wave_lds_sync();
  for (int i = lane; i < DRLGX_MT_N; i += 64) g[i] = s.st[i];
  if (lane == 0) {
    const uint32_t k = (s.cons / DRLGX_MT_N) * DRLGX_MT_N;  // keep the monotonic counters small (same positions mod 624)
    g[DRLGX_MT_N] = s.gen - k;
    g[DRLGX_MT_N + 1] = s.cons - k;
  }
}
// std::mt19937(seed) state (bits/random.tcc seed()); gen = cons = 0 <=> libstdc++'s _M_p = 624
__device__ inline MtStream mt_seed(uint32_t *lds, uint32_t seed, int lane) {
  if (lane == 0) {
    uint32_t x = seed;
    lds[0] = x;
    for (int i = 1; i < DRLGX_MT_N; ++i) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
      lds[i] = x;
    }
  }
  wave_lds_sync();
  MtStream s;
  s.st = lds;
  s.gen = 0;
  s.cons = 0;
  return s;
}
// advance the lazy recurrence by up to 64 words (all 64 lanes).  Word i (<623) needs the OLD word
// i+1 and word 623 the NEW word 0, so a batch never straddles the wrap (the last batch of a round is
// short) and every lane reads before any lane writes.  Word (i+397)%624 is old for i<227 and was
// renewed >= 227 words ago otherwise — never inside the same batch.
__device__ inline void mt_refill(MtStream &s, int lane) {
  uint32_t p = s.gen % DRLGX_MT_N;
  uint32_t nvalid = DRLGX_MT_N - p;
  if (nvalid > 64) nvalid = 64;
  uint32_t i = p + (uint32_t)lane;
  uint32_t a = 0, b = 0, c = 0;
  if ((uint32_t)lane < nvalid) {
    a = s.st[i];
    b = s.st[(i + 1) % DRLGX_MT_N];
    c = s.st[(i + 397) % DRLGX_MT_N];
  }
  wave_lds_sync();
  if ((uint32_t)lane < nvalid) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    s.st[i] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  wave_lds_sync();
  s.gen += nvalid;
}
// next 32-bit output; wave-uniform
__device__ inline uint32_t mt_next(MtStream &s, int lane) {
  if (s.cons == s.gen) mt_refill(s, lane);
  uint32_t y = s.st[s.cons % DRLGX_MT_N];
  s.cons += 1;
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
// std::generate_canonical<double,53>(mt19937): two draws, low word first
__device__ inline double mt_canonical(MtStream &s, int lane) {
  double lo = (double)mt_next(s, lane);
  double hi = (double)mt_next(s, lane);
  double sum = lo + hi * 4294967296.0;
  double r = sum / 18446744073709551616.0;
  if (r >= 1.0) r = 0.99999999999999988897769753748;  // nextafter(1, 0)
  return r;
}
// Same recurrence, up to 192 words per call (3 per lane).  A batch of n <= 227 words that does not straddle the wrap
// has no internal dependency: word i reads the OLD words i, i+1 and word (i+397)%624, which is old for i < 227 and was
// renewed >= 227 words earlier otherwise.  All lanes read before any lane writes.
__device__ inline void mt_refill_wide(MtStream &s, int lane) {
  const uint32_t p = s.gen % DRLGX_MT_N;
  uint32_t nvalid = DRLGX_MT_N - p;
  if (nvalid > 192) nvalid = 192;
  uint32_t a[3], b[3], c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t o = (uint32_t)lane + 64u * k, i = p + o;
    a[k] = b[k] = c[k] = 0;
    if (o < nvalid) {
      a[k] = s.st[i];
      b[k] = s.st[(i + 1) % DRLGX_MT_N];
      c[k] = s.st[(i + 397) % DRLGX_MT_N];
    }
  }
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t o = (uint32_t)lane + 64u * k, i = p + o;
    if (o < nvalid) {
      uint32_t y = (a[k] & 0x80000000u) | (b[k] & 0x7fffffffu);
      s.st[i] = c[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
  }
  wave_lds_sync();
  s.gen += nvalid;
}
// make at least n (<= 256) generated-but-unconsumed words available.  A refill overwrites outputs produced 624 words
// earlier, all consumed because gen - cons never exceeds 255 + 192 here.
__device__ inline void mt_ensure(MtStream &s, int lane, uint32_t n) {
  while (s.gen - s.cons < n) mt_refill_wide(s, lane);
}
// tempered output word number `idx` (cons <= idx < gen) without consuming it
__device__ inline uint32_t mt_peek(const MtStream &s, uint32_t idx) {
  uint32_t y = s.st[idx % DRLGX_MT_N];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
__device__ inline double mt_canonical_words(uint32_t w_lo, uint32_t w_hi) {
  double sum = (double)w_lo + (double)w_hi * 4294967296.0;
  double r = sum / 18446744073709551616.0;
  if (r >= 1.0) r = 0.99999999999999988897769753748;
  return r;
}
struct NormalState {
  double saved;
  int has;
};
// std::normal_distribution<double>(0,1)::operator() (Marsaglia polar, libstdc++ bits/random.tcc)
__device__ inline double normal01(MtStream &s, NormalState &ns, int lane) {
  if (ns.has) {
    ns.has = 0;
    return ns.saved;
  }
  double x, y, r2;
  do {
    x = 2.0 * mt_canonical(s, lane) - 1.0;
    y = 2.0 * mt_canonical(s, lane) - 1.0;
    r2 = x * x + y * y;
  } while (r2 > 1.0 || r2 == 0.0);
  double mult = sqrt(-2 * log(r2) / r2);
  ns.saved = x * mult;
  ns.has = 1;
  return y * mult;
}
// The next `count` variates of std::normal_distribution<double>(0,1) on this stream, all lanes cooperating: lane i examines
// candidate pair i of the Marsaglia polar loop speculatively (words cons+4i .. cons+4i+3), a ballot ranks the accepted
// pairs, the j-th accepted pair yields variates 2j (y*mult) and 2j+1 (x*mult, libstdc++'s saved value), and exactly the
// words up to the last pair that the sequential loop would have examined are consumed.  Bit-identical to `count`
// successive normal01() calls.  The VALUES of the first `skip` variates are not wanted (the caller only advances the
// stream over them): variate number i >= skip goes to out[i - skip] (LDS, count - skip + 1 doubles).
__device__ inline void draw_normals(MtStream &s, NormalState &ns, int count, double *out, int lane, int skip = 0) {
  if (count <= 0) return;
  int offset = 0;
  if (ns.has) {
    if (lane == 0 && skip == 0) out[0] = ns.saved;
    offset = 1;
    ns.has = 0;
  }
  const int R = count - offset;
  const int Np = (R + 1) >> 1;
  int have = 0;
  while (have < Np) {
    mt_ensure(s, lane, 256);
    const uint32_t w = s.cons + 4u * (uint32_t)lane;
    const double x = 2.0 * mt_canonical_words(mt_peek(s, w), mt_peek(s, w + 1)) - 1.0;
    const double y = 2.0 * mt_canonical_words(mt_peek(s, w + 2), mt_peek(s, w + 3)) - 1.0;
    const double r2 = x * x + y * y;
    const bool acc = !(r2 > 1.0 || r2 == 0.0);
    const unsigned long long m = __ballot(acc);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    const int j = have + rank;
    const int i0 = offset + 2 * j;  // numbers of this pair's two variates (the second one may be the saved extra: i0 + 1 == count)
    if (acc && j < Np && i0 + 1 >= skip) {
      const double mult = sqrt(-2 * log(r2) / r2);
      if (i0 >= skip) out[i0 - skip] = y * mult;
      out[i0 + 1 - skip] = x * mult;
    }
    const int A = __popcll(m);
    if (have + A >= Np) {
      const unsigned long long last = __ballot(acc && rank == Np - have - 1);
      const int t = __ffsll((long long)last) - 1;
      s.cons += 4u * (uint32_t)(t + 1);
      have = Np;
    } else {
      s.cons += 256u;
      have += A;
    }
  }
  wave_lds_sync();
  if (R & 1) {
    ns.saved = out[count - skip];
    ns.has = 1;
  }
}
// RNG::normal(m, std) (RNG.h:87-96)
__device__ inline double rng_normal(MtStream &s, NormalState &ns, double m, double sd, int lane) {
  return normal01(s, ns, lane) * sd + m;
}
// RNG::uniformReal (RNG.h:68-71)
__device__ inline double rng_uniform_real(MtStream &s, double lo, double hi, int lane) {
  return (hi - lo) * mt_canonical(s, lane) + lo;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: set it once per (call site, device)
inline void drlgx_ensure_lds_attr(bool (&done)[32], const void *const *fns, int n, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
  if (done[dev]) return;
  for (int i = 0; i < n; ++i) (void)hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done[dev] = true;
}

// LDS of the simulator wave inside the fused step kernel (two mt19937 streams, the normal variates and the in-range list of
// a measure() call, the new landmarks' initial estimates); the SLAM stage of k_step carves behind it (k_step.hip)
__host__ __device__ inline size_t drlgx_sim_lds_bytes(int LG, int P_max) {
  size_t b = (size_t)2 * DRLGX_MT_STRIDE * 4 + (size_t)(2 * LG + 2) * 8 + (size_t)LG * 4 + 16;
  b = ((b + 7) & ~(size_t)7) + (size_t)2 * LG * 8;  // the new landmarks' initial estimates, for the SLAM stage (ksim::measure)
  // ... and, when that costs little, wide enough for the map stage's pose tables (19 P_max doubles), so that the SLAM stage
  // can leave its outputs in them (see k_step)
  if (P_max <= 64) b = b > (size_t)P_max * 19 * 8 + 32 ? b : (size_t)P_max * 19 * 8 + 32;
  return (b + 31) & ~(size_t)31;
}

// ---- launchers implemented in the kernel translation units ------------------------------------
struct DrlgxField;
void drlgx_launch_reset(const DrlgxState &S, hipStream_t st, int n, const int32_t *env_ids_dev, const uint32_t *seeds_dev,
                        const double *start_dev, int first_measure = 1);
// staged interface: mode 0 = move + addOdometry, mode 1 = one exporting measure (keys [n][LG], br [n][LG][2], count [n])
void drlgx_launch_sim_stage(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int mode, int32_t *keys,
                            double *br, int32_t *count);
void drlgx_launch_add_measurements(const DrlgxState &S, hipStream_t st, LaunchSel sel, const int32_t *keys, const double *br,
                                   const int32_t *count);
// bytes of one action's entry in the look-ahead's simulator log
size_t drlgx_simlog_entry_bytes(const DrlgxState &S);
// run the simulator of every selected instance over its WHOLE action list [0, min(a_end, n_act[i])) and log what each action appends
void drlgx_launch_presim(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure, int a_end,
                         unsigned char *simlog, size_t simlog_roll, int simlog_act);
void drlgx_launch_sim(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride,
                      int n_measure);
// p_bound: host-side upper bound of the pose count of every selected instance after this launch (<= P_max); it picks
// the kernel variant (the kernels flag DRLGX_E_CAPACITY instead of overrunning if the bound is violated)
void drlgx_launch_slam(const DrlgxState &S, hipStream_t st, LaunchSel sel, int p_bound);
size_t drlgx_map_lds_bytes(const DrlgxState &S, int *chunk_out, int pcap = 0);  // pcap: LaunchSel::cap (0: S.P_max)
// fused simulate + SLAM + map kernel (k_step.hip); usable when the SLAM system and the map stage fit the LDS
bool drlgx_step_fusable(const DrlgxState &S, int p_bound);
bool drlgx_step_arrow_fusable(const DrlgxState &S);
void drlgx_launch_step_arrow(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure);
void drlgx_launch_step(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure);  // requires drlgx_step_fusable
// every instance's actions [sel.act_idx, min(a_end, n_act[i])) in ONE launch; sel.pcap = the pose bound of the FIRST action of the
// range (action a runs with sel.pcap + a - sel.act_idx); requires drlgx_step_fusable for the bound of the LAST one
void drlgx_launch_step_loop(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure, int a_end);
void drlgx_launch_step_arrow_loop(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure, int a_end);  // requires drlgx_step_arrow_fusable
void drlgx_launch_map(const DrlgxState &S, hipStream_t st, LaunchSel sel);  // sel.act_idx == -2: reductions only
bool drlgx_map_two_per_cu(const DrlgxState &S, int p_bound);
void drlgx_launch_copy(const DrlgxField *fields_dev, int n_fields, hipStream_t st, int n, const int32_t *src,
                       const int32_t *dst, int src_off, int dst_off, int skip_mask,  // skip fields with cls & mask
                       const int *cnt = nullptr,  // S.cnt: copy the live part of per-pose / -landmark / -factor fields only
                       const DrlgxState *panel = nullptr);  // &S: the same launch copies the instances' covariance panels too
void drlgx_launch_rebase(const DrlgxState &S, hipStream_t st, int base0, int n);
// head of a packed status read: the status word + n_envs pose counts, rounded up to 16 bytes (k_fetch_pack)
inline size_t drlgx_fetch_head_bytes(int n_envs) { return ((size_t)(n_envs + 1) * sizeof(int) + 15) & ~(size_t)15; }
void drlgx_launch_fetch_pack(const DrlgxState &S, hipStream_t st, const void *src, size_t bytes, void *out);
void drlgx_launch_fix_rollouts(const DrlgxState &S, hipStream_t st, int n_cand, const int32_t *cand_env, int roll0);
void drlgx_launch_rewards(const DrlgxState &S, hipStream_t st, int n_cand, const int32_t *cand_env, int roll0,
                          double *rewards);
void drlgx_launch_utility(const DrlgxState &S, hipStream_t st, const double *dist, double *out, int mode);
// FastMarginals2 (k_fm2.hip): dense prior covariance per environment, then the per-candidate update
void drlgx_launch_fm2_prior(const DrlgxState &S, hipStream_t st, const int32_t *env_ids, int n_env, double *sig, size_t sig_stride,
                            int n_max);
size_t drlgx_fm2_scratch_doubles(const DrlgxState &S, int nm_max);
void drlgx_launch_fm2_update(const DrlgxState &S, hipStream_t st, int c0, int nc, const int32_t *cand_env, const int32_t *env_slot,
                             const double *actions, const int32_t *n_actions, const double *sig, size_t sig_stride, double *scratch,
                             size_t scratch_stride, int *iscratch, int nm_max, double *cov_out, int out_stride, int32_t *n_out);
void drlgx_launch_metrics(const DrlgxState &S, hipStream_t st, double sigma0, double *out);
void drlgx_launch_cov_array(const DrlgxState &S, hipStream_t st, double *length, double *angle);
void drlgx_launch_line_plan(const DrlgxState &S, hipStream_t st, int n_cand, const int32_t *cand_env, const double *goal,
                            double *actions, int32_t *n_actions);
bool drlgx_slam_in_lds(int P_max, int L_max, int M_max);       // the fused LDS-resident kernel serves up to P_max poses
bool drlgx_slam_capacity_ok(int P_max, int L_max, int M_max);  // capacities the SLAM kernels can serve at all
size_t drlgx_slam_ws_doubles(int P_max, int L_max, int M_max);  // HBM workspace per instance (doubles)
void drlgx_launch_graph(const DrlgxState &S, hipStream_t st, int *gi, int gi_stride, int32_t *node_off, int32_t *edge_off,
                        float *x, int64_t *edge_index, float *edge_attr, int32_t *n_frontier, double *frontier_xy,
                        int32_t *nearest_node, int max_frontier);
