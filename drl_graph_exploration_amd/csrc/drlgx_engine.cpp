// drlgx engine: host side of the C ABI (include/drlgx.h) — HBM state allocation, kernel
// orchestration for reset / step / look-ahead, state export.  No CPU compute path exists here: every
// belief-step quantity is produced by the HIP kernels in k_*.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <cstdlib>
#include <vector>

#include "drlgx_dev.h"
#include "drlgx_fields.h"

#define HIPCHK(e, call)                                                                       \
  do {                                                                                        \
    hipError_t _r = (call);                                                                   \
    if (_r != hipSuccess) {                                                                   \
      (e)->last_error = std::string(#call) + ": " + hipGetErrorString(_r);                    \
      return DRLGX_E_HIP;                                                                     \
    }                                                                                         \
  } while (0)

struct TimedSpan {
  hipEvent_t a, b;
  int id;
};

struct drlgx_engine {
  DrlgxState S{};
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::vector<void *> allocs;
  // the state struct's copy in device memory (DrlgxState::self_dev) and what was uploaded last: every entry point compares and
  // re-uploads before it launches anything (state_sync), whatever changed the struct
  DrlgxState *state_dev = nullptr;
  unsigned char state_shadow[sizeof(DrlgxState)];
  std::vector<DrlgxField> fields;       // per-instance fields (n_inst instances)
  DrlgxField *fields_dev = nullptr;
  std::vector<int> lm_order;
  // staging
  int32_t *stage_i32 = nullptr;
  uint32_t *stage_u32 = nullptr;
  double *stage_f64 = nullptr;
  uint8_t *stage_mask = nullptr;
  int *graph_gi = nullptr;
  int graph_gi_stride = 0;
  // timing
  bool timing = false;
  // Host-side upper bound of every env's pose count: exact after drlgx_reset_host / drlgx_status_host / a restore of a
  // snapshot taken in an exact state, +1 per drlgx_step in between (the active mask lives on the device).  It selects
  // the k_slam variant and whether the fused step kernel applies, per launch, so that a large pose capacity does not
  // slow the steps of short trajectories down.
  std::vector<int> pbound;
  bool by_capacity = false;  // DRLGX_VARIANT_BY_CAPACITY=1 (tests): select the variants by max_poses as if every env were full
  std::vector<std::vector<int>> snap_pbound;
  bool per_stage = false;  // timing mode 2: launch the three stage kernels separately so that each gets its own span
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> free_events;
  double t_ms[DRLGX_N_TIMERS] = {0};
  int64_t t_n[DRLGX_N_TIMERS] = {0};
  std::string last_error;
  double *fixed_lm_dev = nullptr;  // drlgx_set_fixed_landmarks_host
  unsigned char *simlog_dev = nullptr;  // the look-ahead's simulator log (k_presim), n_rollouts x max_actions entries
  bool la_presim = true;
  // drlgx_status_host / drlgx_status_fetch_host wait by polling an event (DRLGX_SYNC_SPIN=0: hipStreamSynchronize): a trainer's vector
  // step reads the status word four times, each on the critical path between two launches, and a blocking wait hands the thread
  // back tens of microseconds after the stream drained
  bool spin_sync = true;
  hipEvent_t sync_event = nullptr;
  // packed status reads (k_fetch_pack): device staging and its pinned host mirror, grown on demand
  unsigned char *fetch_dev = nullptr, *fetch_host = nullptr;
  size_t fetch_cap = 0;
  int n_cu = 256;       // compute units of the device (drlgx_create)
  bool la_loop = true;  // look-ahead rollouts: one launch for a candidate's whole action list (k_step_loop)
  // FastMarginals2 workspaces (allocated on first use): dense prior covariances, per-candidate scratch
  double *fm2_sig = nullptr, *fm2_scratch = nullptr;
  int *fm2_iscratch = nullptr;
  int32_t *fm2_slot = nullptr;
  size_t fm2_sig_stride = 0, fm2_scratch_stride = 0;
};

// every entry point makes the engine's device current (a process may drive several engines on several devices)
#define DRLGX_ENTER(e)                          \
  do {                                          \
    if (e) {                                    \
      (void)hipSetDevice((e)->device);          \
      state_sync(e);                            \
    }                                           \
  } while (0)

struct drlgx_engine;
static void state_sync(drlgx_engine *e);

namespace {

template <typename T>
int dev_alloc(drlgx_engine *e, T **out, size_t count) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 256);
  HIPCHK(e, hipMalloc(&p, bytes));
  HIPCHK(e, hipMemsetAsync(p, 0, bytes, e->stream));
  e->allocs.push_back(p);
  *out = reinterpret_cast<T *>(p);
  return DRLGX_OK;
}

template <typename T>
int field_alloc(drlgx_engine *e, T **out, size_t per_inst, int cls = 0, int unit = 0, size_t per_unit = 0) {
  int r = dev_alloc(e, out, per_inst * (size_t)e->S.n_inst);
  if (r) return r;
  e->fields.push_back(DrlgxField{reinterpret_cast<char *>(*out), per_inst * sizeof(T), cls, unit, (int)(per_unit * sizeof(T)), 0});
  return DRLGX_OK;
}

hipEvent_t get_event(drlgx_engine *e) {
  if (!e->free_events.empty()) {
    hipEvent_t ev = e->free_events.back();
    e->free_events.pop_back();
    return ev;
  }
  hipEvent_t ev;
  hipEventCreate(&ev);
  return ev;
}
struct ScopedTimer {
  drlgx_engine *e;
  TimedSpan s;
  bool on;
  ScopedTimer(drlgx_engine *e_, int id) : e(e_), on(e_->timing) {
    if (on) {
      s.a = get_event(e);
      s.b = get_event(e);
      s.id = id;
      hipEventRecord(s.a, e->stream);
    }
  }
  ~ScopedTimer() {
    if (on) {
      hipEventRecord(s.b, e->stream);
      e->spans.push_back(s);
    }
  }
};

int check_launch(drlgx_engine *e) {
  hipError_t r = hipGetLastError();
  if (r != hipSuccess) {
    e->last_error = std::string("kernel launch: ") + hipGetErrorString(r);
    return DRLGX_E_HIP;
  }
  return DRLGX_OK;
}

}  // namespace

extern "C" {

const char *drlgx_strerror(int code) {
  switch (code) {
    case DRLGX_OK: return "ok";
    case DRLGX_E_INVALID: return "invalid argument";
    case DRLGX_E_NODEVICE: return "no HIP device";
    case DRLGX_E_CAPACITY: return "instance capacity exceeded (max_poses / max_landmarks / max_factors / max_actions)";
    case DRLGX_E_HIP: return "HIP runtime error";
    case DRLGX_E_NUMERIC: return "indeterminate linear system in the SLAM solve";
    default: return "unknown error";
  }
}

const char *drlgx_last_error(const drlgx_engine *e) { return e ? e->last_error.c_str() : "null engine"; }

int drlgx_create(const drlgx_config *cfg, int n_envs, int n_rollouts, int device, drlgx_engine **out) {
  if (!cfg || !out || n_envs <= 0 || n_rollouts < 0) return DRLGX_E_INVALID;
  if (cfg->max_poses < 2 || cfg->max_landmarks < 1 || cfg->max_factors < 1 || cfg->num_landmarks < 0 ||
      cfg->max_actions < 1 || cfg->num_samples < 1 || !(cfg->resolution > 0))
    return DRLGX_E_INVALID;
  // kernel limits: 16-bit pose / landmark / factor indices in LDS tables; the per-pose / per-landmark tables of k_slam_arrow
  // must fit the LDS (its landmark system is streamed from the workspace beyond 127 landmarks)
  if (cfg->max_poses > 65535 || cfg->max_landmarks > 65535 || cfg->max_factors > 65534) return DRLGX_E_INVALID;
  if (!drlgx_slam_capacity_ok(cfg->max_poses, cfg->max_landmarks, cfg->max_factors)) return DRLGX_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DRLGX_E_NODEVICE;
  drlgx_engine *e = new drlgx_engine();
  e->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete e;
    return DRLGX_E_NODEVICE;
  }
  if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete e;
    return DRLGX_E_NODEVICE;
  }
  e->stream = e->own_stream;
  DrlgxState &S = e->S;
  S.cfg = *cfg;
  S.n_envs = n_envs;
  S.n_roll = n_rollouts;
  if (cfg->max_snapshots < 0 || cfg->max_snapshots > 16) {
    drlgx_destroy(e);
    return DRLGX_E_INVALID;
  }
  // instances: [0,n) live envs | [n,2n) look-ahead bases | rollouts | max_snapshots x n snapshot copies
  S.n_inst = 2 * n_envs + n_rollouts + cfg->max_snapshots * n_envs;
  e->pbound.assign(n_envs, cfg->max_poses);  // unknown until the first reset
  {
    const char *v = getenv("DRLGX_VARIANT_BY_CAPACITY");
    e->by_capacity = v && v[0] == '1';
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->n_cu = prop.multiProcessorCount;
    const char *ss = getenv("DRLGX_SYNC_SPIN");
    e->spin_sync = !(ss && ss[0] == '0');
    const char *lp = getenv("DRLGX_LOOKAHEAD_PRESIM");  // 0: every rollout action simulates inside its belief step (the A/B of the parity test)
    e->la_presim = !(lp && lp[0] == '0');
    const char *ll = getenv("DRLGX_LOOKAHEAD_LOOP");  // 0: one launch per action index (the A/B of the look-ahead tests)
    e->la_loop = !(ll && ll[0] == '0');
  }
  e->snap_pbound.assign(cfg->max_snapshots > 0 ? cfg->max_snapshots : 0, std::vector<int>(n_envs, cfg->max_poses));
  S.P_max = cfg->max_poses;
  S.L_max = cfg->max_landmarks;
  S.M_max = cfg->max_factors;
  S.LG = std::max(cfg->num_landmarks, 1);
  S.A_max = cfg->max_actions;
  // VirtualMap::initialize (VirtualMap.cpp:318-343)
  S.cols = (int)std::floor((cfg->map_max_x - cfg->map_min_x) / cfg->resolution);
  S.rows = (int)std::floor((cfg->map_max_y - cfg->map_min_y) / cfg->resolution);
  S.V = S.rows * S.cols;
  S.Vu = (S.V + 3) / 4 * 4;
  {
    int extg = 20;
    S.count_explored = (S.rows - extg * 2 / (int)cfg->resolution) * (S.cols - extg * 2 / (int)cfg->resolution);
  }
  // OccupancyMap uses ceil() for its grid (OccupancyMap.cpp:13-14); the reference asserts equal sizes
  if ((int)std::ceil((cfg->map_max_x - cfg->map_min_x) / cfg->resolution) != S.cols ||
      (int)std::ceil((cfg->map_max_y - cfg->map_min_y) / cfg->resolution) != S.rows || S.V <= 0) {
    e->last_error = "map extent must be a multiple of the resolution";
    drlgx_destroy(e);
    return DRLGX_E_INVALID;
  }
  S.win = (int)std::ceil(2.0 * cfg->max_range / cfg->resolution) + 1;
  if (S.win * S.win > 64) {
    e->last_error = "max_range / resolution too large for the 64-lane cell window";
    drlgx_destroy(e);
    return DRLGX_E_INVALID;
  }
  // log-odds constants (OccupancyMap.h:10-19) evaluated with the HOST libm: exact ladder values
  auto p2l = [](double p) { return std::log(p / (1.0 - p)); };
  auto l2p = [](double l) { return std::exp(l) / (1.0 + std::exp(l)); };
  S.lo_free = p2l(0.3);
  S.lo_occ = p2l(0.7);
  S.lo_min = p2l(0.05);
  S.lo_max = l2p(0.95);  // sic: MAX_LOGODDS = LOGODDS2PROB(0.95)
  S.occ_thresh = p2l(0.5);
  S.vm_i0 = 1.0 / std::pow(cfg->sigma0, 2);
  S.w_trans = 1.0 / (cfg->translation_noise * cfg->translation_noise);
  S.w_rot = 1.0 / (cfg->rotation_noise * cfg->rotation_noise);
  S.w_bear = 1.0 / (cfg->bearing_noise * cfg->bearing_noise);
  S.w_range = 1.0 / (cfg->range_noise * cfg->range_noise);
  S.lo_tocc = S.lo_tfree = 0ull;
  S.lo_tflag = 0u;
  // occupancy ladder closure (see DrlgxState::lo_tr)
  std::vector<double> lo_val{0.0}, lo_pv;
  std::vector<uint8_t> lo_tr;
  {
    auto clampl = [&](double l) { return std::fmin(S.lo_max, std::fmax(S.lo_min, l)); };
    auto find_or_add = [&](double l) -> int {
      for (size_t i = 0; i < lo_val.size(); ++i)
        if (lo_val[i] == l) return (int)i;
      lo_val.push_back(l);
      return (int)lo_val.size() - 1;
    };
    // Only the transitions the update rules can take are followed (OccupancyMap.cpp:55-138): the landmark pre-updates
    // apply `occupied` along the chain 0 -> occ(0) -> ...; a pose update leaves a cell at the minimum alone, adds
    // `occupied` above the threshold and `free` otherwise.  (Following both successors of every state does not close:
    // lo_occ and lo_free are not exact negatives of each other in floating point, so the values drift by ulps.)
    bool closed = true;
    std::vector<int> is_chain{1};  // states reachable by landmark pre-updates alone
    for (size_t i = 0; i < lo_val.size(); ++i) {
      if (lo_val.size() > DRLGX_LO_TAB) {
        closed = false;
        break;
      }
      const double l = lo_val[i];
      const bool frozen = std::fabs(l - S.lo_min) < 1e-5, above = l > S.occ_thresh + 1e-8;
      int o = (int)i, f = (int)i;
      if (above || is_chain[i]) {
        o = find_or_add(clampl(l + S.lo_occ));
        if ((size_t)o >= is_chain.size()) is_chain.push_back(0);
        if (is_chain[i]) is_chain[o] = 1;
      }
      if (!above && !frozen) {
        f = find_or_add(clampl(l + S.lo_free));
        if ((size_t)f >= is_chain.size()) is_chain.push_back(0);
      }
      const uint8_t flags = (uint8_t)((frozen ? 1 : 0) | (above ? 2 : 0));
      lo_tr.push_back((uint8_t)o); lo_tr.push_back((uint8_t)f); lo_tr.push_back(flags); lo_tr.push_back(0);
    }
    if (closed && lo_val.size() <= DRLGX_LO_TAB) {
      for (double l : lo_val) {  // OccupancyMap LOGODDS2PROB + VirtualMap::updateProbability (num_samples identical maps)
        const double pv1 = l2p(l);
        double acc = 0.0;
        for (int s2 = 0; s2 < cfg->num_samples; ++s2) acc += pv1 / cfg->num_samples;
        lo_pv.push_back(acc);
      }
      S.lo_ntab = (int)lo_val.size();
      if (S.lo_ntab <= 16)
        for (int i = 0; i < S.lo_ntab; ++i) {
          S.lo_tocc |= (unsigned long long)(lo_tr[4 * i] & 15) << (4 * i);
          S.lo_tfree |= (unsigned long long)(lo_tr[4 * i + 1] & 15) << (4 * i);
          S.lo_tflag |= (unsigned int)(lo_tr[4 * i + 2] & 3) << (2 * i);
        }
    } else {
      S.lo_ntab = 0;
      lo_pv.assign(1, 0.5);
      lo_tr.assign(4, 0);
    }
  }
  // sector-sweep table (OccupancyMap.cpp:86): b accumulates in double exactly as the reference loop
  std::vector<double> sweep;
  for (double b = cfg->min_bearing; b < cfg->max_bearing + 1e-5; b += 3 * 0.01745329251994329575) sweep.push_back(b);
  S.n_sweep = (int)sweep.size();
  {
    // The bbox only prunes work if an in-range cell can fall outside it.  A cell centre within max_range of the
    // pose lies at most (max_range - res/2) past the pose's own cell boundary, so the box contains it as soon as
    // some sweep sample is within acos(1 - res / (2 max_range)) of each axis direction (k_map).
    const double two_pi = 6.283185307179586476925286766559;
    double max_gap = 3 * 0.01745329251994329575;
    if (!sweep.empty()) max_gap = std::max(max_gap, two_pi - (sweep.back() - sweep.front()));
    const double need = std::acos(std::max(-1.0, 1.0 - cfg->resolution / (2.0 * cfg->max_range)));
    S.bbox_noop = (0.5 * max_gap + 1e-6 <= need) ? 1 : 0;
    // field of view: blind half-angle around the backwards ray
    const double pi = 3.14159265358979323846;
    const double blind = std::max(pi - cfg->max_bearing, pi + cfg->min_bearing);
    S.fov_fast = (cfg->max_bearing > 1.7 && cfg->min_bearing < -1.7 && blind >= 0 && blind + 1e-3 < 1.4) ? 1 : 0;
    S.fov_tan = S.fov_fast ? std::tan(blind + 1e-3) : 0.0;
    // smallest x with sqrt(x) >= max_range, largest x with sqrt(x) <= min_range (sqrt is correctly rounded
    // and monotone on both host and device, so the squared comparisons are EXACTLY the reference's tests)
    double t = cfg->max_range * cfg->max_range;
    while (std::sqrt(std::nextafter(t, 0.0)) >= cfg->max_range) t = std::nextafter(t, 0.0);
    while (std::sqrt(t) < cfg->max_range) t = std::nextafter(t, INFINITY);
    S.r2_max_lt = t;
    t = cfg->min_range * cfg->min_range;
    while (std::sqrt(std::nextafter(t, INFINITY)) <= cfg->min_range) t = std::nextafter(t, INFINITY);
    while (t > 0 && std::sqrt(t) > cfg->min_range) t = std::nextafter(t, 0.0);
    S.r2_min_gt = t;
  }
  // libstdc++ iteration order of unordered_map<unsigned, ...> filled with keys 0..n-1 (Simulator2D.cpp:331-344)
  {
    std::unordered_map<unsigned, int> m;
    for (int i = 0; i < cfg->num_landmarks; ++i) m.emplace((unsigned)i, i);
    for (const auto &kv : m) e->lm_order.push_back((int)kv.first);
    if (e->lm_order.empty()) e->lm_order.push_back(0);
  }
  int r = DRLGX_OK;
  double *sweep_dev = nullptr;
  int *order_dev = nullptr;
#define TRY(x)            \
  if ((r = (x)) != 0) {   \
    drlgx_destroy(e);     \
    return r;             \
  }
  TRY(dev_alloc(e, &sweep_dev, sweep.size()));
  TRY(dev_alloc(e, &order_dev, e->lm_order.size()));
  hipMemcpyAsync(sweep_dev, sweep.data(), sweep.size() * sizeof(double), hipMemcpyHostToDevice, e->stream);
  hipMemcpyAsync(order_dev, e->lm_order.data(), e->lm_order.size() * sizeof(int), hipMemcpyHostToDevice, e->stream);
  S.sweep_b = sweep_dev;
  S.lm_order = order_dev;
  {
    double *pv_dev = nullptr;
    uint8_t *tr_dev = nullptr;
    TRY(dev_alloc(e, &pv_dev, lo_pv.size()));
    TRY(dev_alloc(e, &tr_dev, lo_tr.size()));
    HIPCHK(e, hipMemcpyAsync(pv_dev, lo_pv.data(), lo_pv.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemcpyAsync(tr_dev, lo_tr.data(), lo_tr.size(), hipMemcpyHostToDevice, e->stream));
    S.lo_pv = pv_dev;
    S.lo_tr = tr_dev;
  }
  const size_t P = S.P_max, L = S.L_max, M = S.M_max, V = S.V;
  TRY(field_alloc(e, &S.gt_pose, 4));
  TRY(field_alloc(e, &S.parent, 1));
  TRY(field_alloc(e, &S.mt, 2 * DRLGX_MT_STRIDE));
  TRY(field_alloc(e, &S.nrm_saved, 2));
  TRY(field_alloc(e, &S.nrm_has, 2));
  TRY(field_alloc(e, &S.cnt, DRLGX_CNT_STRIDE));
  TRY(field_alloc(e, &S.th_pose, P * 4, 0, 1, 4));
  TRY(field_alloc(e, &S.d_pose, P * 3, 0, 1, 3));
  TRY(field_alloc(e, &S.th_lm, L * 2, 0, 2, 2));
  TRY(field_alloc(e, &S.d_lm, L * 2, 0, 2, 2));
  TRY(field_alloc(e, &S.lm_key, L, 0, 2, 1));
  TRY(field_alloc(e, &S.key_slot, (size_t)S.LG));
  TRY(field_alloc(e, &S.prior, DRLGX_PRIOR_STRIDE));
  TRY(field_alloc(e, &S.odo, P * 4, 0, 1, 4));
  TRY(field_alloc(e, &S.meas_pose, M, 0, 3, 1));
  TRY(field_alloc(e, &S.meas_lm, M, 0, 3, 1));
  TRY(field_alloc(e, &S.meas_br, M * 2, 0, 3, 2));
  TRY(field_alloc(e, &S.est_pose, P * 4, 0, 1, 4));
  TRY(field_alloc(e, &S.est_lm, L * 2, 0, 2, 2));
  TRY(field_alloc(e, &S.pose_info, P * 6, 0, 1, 6));
  TRY(field_alloc(e, &S.lm_info, L * 3, 0, 2, 3));
  TRY(field_alloc(e, &S.pose_tr, P, 0, 1, 1));
  TRY(field_alloc(e, &S.lm_tr, L, 0, 2, 1));
  TRY(field_alloc(e, &S.red, DRLGX_RED_STRIDE));
  TRY(field_alloc(e, &S.vm_prob, V, 1));
  TRY(field_alloc(e, &S.vm_info, 3 * V, 1));
#ifndef COPY_EXP_NOSPLIT
  {
    // the three planes of the information as three entries of the copy table (same stride, a third of the bytes each): the 38 KB slice
    // was the copy kernel's longest workgroup by far
    DrlgxField f = e->fields.back();
    e->fields.pop_back();
    for (int k = 0; k < 3; ++k) {
      DrlgxField g = f;
      g.base = f.base + (size_t)k * V * sizeof(double);
      g.pad = (int)(V * sizeof(double));
      e->fields.push_back(g);
    }
  }
#endif
  TRY(field_alloc(e, &S.vm_upd, (size_t)S.Vu, 1));
  TRY(field_alloc(e, &S.vm_tr, V, 1));
  TRY(field_alloc(e, &S.gt_lm, (size_t)S.LG * 2, 2));  // rollouts read their parent's landmarks
  // SLAM workspace (not copied between instances)
  {
    // k_slam workspace (k_slam.hip: drlgx_slam_ws_doubles)
    S.slam_ws_stride = drlgx_slam_ws_doubles(S.P_max, S.L_max, S.M_max);
    S.slam_iws_stride = 2;
    TRY(dev_alloc(e, &S.slam_ws, S.slam_ws_stride * (size_t)S.n_inst));
    TRY(dev_alloc(e, &S.slam_iws, S.slam_iws_stride * (size_t)S.n_inst));
  }
  // Covariance panel of the incremental belief update (k_inc.hip): on unless DRLGX_INCREMENTAL=0 or the panels of all
  // instances would not fit the budget (DRLGX_INC_MAX_GB; default: 60 % of the memory that is free on the device now) or the
  // allocation fails - then every update is a full solve (the engine works either way; said once on stderr).
  {
    const char *v = getenv("DRLGX_INCREMENTAL");
    const char *g = getenv("DRLGX_INC_MAX_GB");
    size_t free_b = 0, total_b = 0;
    double max_gb = 32.0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) max_gb = 0.6 * (double)free_b / 1073741824.0;
    if (g) max_gb = atof(g);
    S.jc_ld = (3 + 2 * S.L_max + 31) & ~31;  // whole pairs of 16-column tiles, rows on 256-byte boundaries (k_inc.hip, B3)
    S.jc_stride = (size_t)(3 * S.P_max + 2 * S.L_max + 16) * (size_t)S.jc_ld;  // (+ 16: the row tiles are written whole)
    const double gb = (double)S.jc_stride * 8.0 * (double)S.n_inst / 1073741824.0;
    if (!(v && v[0] == '0')) {
      const size_t n_before = e->allocs.size();
      bool ok = gb <= max_gb;
      if (ok) {
        ok = dev_alloc(e, &S.jc, S.jc_stride * (size_t)S.n_inst) == DRLGX_OK && dev_alloc(e, &S.jd, (size_t)S.P_max * 6 * (size_t)S.n_inst) == DRLGX_OK &&
             dev_alloc(e, &S.jc_meta, (size_t)4 * (size_t)S.n_inst) == DRLGX_OK && dev_alloc(e, &S.inc_stats, 2) == DRLGX_OK;
      }
      if (!ok) {
        (void)hipGetLastError();  // a failed hipMalloc is not this engine's error
        while (e->allocs.size() > n_before) {
          hipFree(e->allocs.back());
          e->allocs.pop_back();
        }
        S.jc = nullptr;
        S.jd = nullptr;
        S.jc_meta = nullptr;
        S.inc_stats = nullptr;
        e->last_error.clear();
        fprintf(stderr, "drlgx: covariance panels (%.1f GB) not allocated (budget %.1f GB): every belief update is a full solve\n", gb, max_gb);
      }
    }
  }
  TRY(dev_alloc(e, &S.status, 1));
  // (the copy kernel deals a workgroup to every (instance, field): the largest slices first, so that none of them starts last)
  std::stable_sort(e->fields.begin(), e->fields.end(), [](const DrlgxField &a, const DrlgxField &b) {
    return (a.pad > 0 ? (size_t)a.pad : a.stride) > (b.pad > 0 ? (size_t)b.pad : b.stride);
  });
  TRY(dev_alloc(e, &e->fields_dev, e->fields.size()));
  hipMemcpyAsync(e->fields_dev, e->fields.data(), e->fields.size() * sizeof(DrlgxField), hipMemcpyHostToDevice, e->stream);
  TRY(dev_alloc(e, &e->stage_i32, (size_t)n_envs));
  TRY(dev_alloc(e, &e->stage_u32, (size_t)n_envs));
  TRY(dev_alloc(e, &e->stage_f64, (size_t)n_envs * 3));
  TRY(dev_alloc(e, &e->stage_mask, (size_t)n_envs));
  e->graph_gi_stride = 4 * S.L_max + 8;
  TRY(dev_alloc(e, &e->graph_gi, (size_t)n_envs * e->graph_gi_stride));
  {
    // the struct's device copy: what the belief kernels read their state from (DrlgxStateConst; drlgx_dev.h)
    unsigned char *raw = nullptr;
    TRY(dev_alloc(e, &raw, sizeof(DrlgxState)));
    e->state_dev = reinterpret_cast<DrlgxState *>(raw);
    S.self_dev = e->state_dev;
  }
#undef TRY
  if (hipStreamSynchronize(e->stream) != hipSuccess) {
    drlgx_destroy(e);
    return DRLGX_E_HIP;
  }
  state_sync(e);
  *out = e;
  return DRLGX_OK;
}

// The device copy of the state struct follows the host's: compared at every entry point (0.8 KB), uploaded when it differs -
// after whatever the stream still runs with the old one.
static void state_sync(drlgx_engine *e) {
  if (!e->state_dev || memcmp(e->state_shadow, &e->S, sizeof(DrlgxState)) == 0) return;
  (void)hipStreamSynchronize(e->stream);
  (void)hipMemcpy(e->state_dev, &e->S, sizeof(DrlgxState), hipMemcpyHostToDevice);
  memcpy(e->state_shadow, &e->S, sizeof(DrlgxState));
}

int drlgx_destroy(drlgx_engine *e) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  for (void *p : e->allocs) hipFree(p);
  for (auto &sp : e->spans) {
    hipEventDestroy(sp.a);
    hipEventDestroy(sp.b);
  }
  for (auto ev : e->free_events) hipEventDestroy(ev);
  if (e->sync_event) hipEventDestroy(e->sync_event);
  if (e->fetch_dev) hipFree(e->fetch_dev);
  if (e->fetch_host) hipHostFree(e->fetch_host);
  if (e->own_stream) hipStreamDestroy(e->own_stream);
  delete e;
  return DRLGX_OK;
}

int drlgx_set_stream(drlgx_engine *e, void *hip_stream) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  hipStream_t s = (hip_stream == reinterpret_cast<void *>(-1)) ? e->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
  if (s == e->stream) return DRLGX_OK;  // cheap when nothing changes: callers re-bind before every call
  hipStreamSynchronize(e->stream);      // work already queued on the old stream is ordered before the new one's
  e->stream = s;
  return DRLGX_OK;
}

int drlgx_synchronize(drlgx_engine *e) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return DRLGX_OK;
}

static int max_bound(const drlgx_engine *e) {
  if (e->by_capacity) return e->S.P_max;
  int m = 1;
  for (int v : e->pbound) m = std::max(m, v);
  return m;
}

// the stream drained, observed by polling (see drlgx_engine::spin_sync)
static hipError_t stream_wait(drlgx_engine *e) {
  if (!e->spin_sync) return hipStreamSynchronize(e->stream);
  if (!e->sync_event && hipEventCreateWithFlags(&e->sync_event, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    e->spin_sync = false;
    return hipStreamSynchronize(e->stream);
  }
  hipError_t r = hipEventRecord(e->sync_event, e->stream);
  if (r != hipSuccess) return r;
  // (polling pays for the waits of a vector step - a look-ahead, a plan execution: up to a few milliseconds; behind a longer queue the
  // thread hands the core back)
  const auto t0 = std::chrono::steady_clock::now();
  for (int polls = 0; (r = hipEventQuery(e->sync_event)) == hipErrorNotReady; ++polls) {
    if ((polls & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(5000)) return hipStreamSynchronize(e->stream);
  }
  return r;
}

// status word + pose counts (+ `bytes` of the caller's device memory -> dst_host) in one launch, one copy and one wait
static int status_fetch(drlgx_engine *e, const void *src_dev, size_t bytes, void *dst_host) {
  const size_t head = drlgx_fetch_head_bytes(e->S.n_envs), total = head + ((bytes + 15) & ~(size_t)15);
  if (total > e->fetch_cap) {
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->fetch_dev) (void)hipFree(e->fetch_dev);
    if (e->fetch_host) (void)hipHostFree(e->fetch_host);
    e->fetch_dev = e->fetch_host = nullptr;
    e->fetch_cap = 0;
    const size_t cap = std::max<size_t>(2 * total, 1 << 16);
    HIPCHK(e, hipMalloc(reinterpret_cast<void **>(&e->fetch_dev), cap));
    HIPCHK(e, hipHostMalloc(reinterpret_cast<void **>(&e->fetch_host), cap, hipHostMallocDefault));
    e->fetch_cap = cap;
  }
  drlgx_launch_fetch_pack(e->S, e->stream, src_dev, bytes, e->fetch_dev);
  HIPCHK(e, hipMemcpyAsync(e->fetch_host, e->fetch_dev, head + bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, stream_wait(e));
  const int32_t *h = reinterpret_cast<const int32_t *>(e->fetch_host);
  // the same synchronisation refreshes the host's pose-count bounds with the exact device values
  for (int i = 0; i < e->S.n_envs; ++i) e->pbound[i] = std::min(std::max((int)h[1 + i], 1), e->S.P_max);
  if (bytes) std::memcpy(dst_host, e->fetch_host + head, bytes);
  return h[0];
}

int drlgx_status_host(drlgx_engine *e) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  return status_fetch(e, nullptr, 0, nullptr);
}

int drlgx_status_fetch_host(drlgx_engine *e, const void *src_dev, size_t bytes, void *dst_host) {
  DRLGX_ENTER(e);
  if (!e || (bytes > 0 && (!src_dev || !dst_host))) return DRLGX_E_INVALID;
  // the caller's bytes ride on the status read's synchronisation (a vector step of a trainer needs a handful of small device
  // results on the host: every separate read drains the stream again)
  return status_fetch(e, src_dev, bytes, dst_host);
}

int drlgx_reset_host(drlgx_engine *e, int n, const int32_t *env_ids, const uint32_t *seeds, const double *start) {
  DRLGX_ENTER(e);
  if (!e || n <= 0 || n > e->S.n_envs || !env_ids || !seeds || !start) return DRLGX_E_INVALID;
  std::vector<uint8_t> mask(e->S.n_envs, 0);
  for (int i = 0; i < n; ++i) {
    if (env_ids[i] < 0 || env_ids[i] >= e->S.n_envs || mask[env_ids[i]]) return DRLGX_E_INVALID;
    mask[env_ids[i]] = 1;
  }
  hipSetDevice(e->device);
  HIPCHK(e, hipMemcpyAsync(e->stage_i32, env_ids, n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->stage_u32, seeds, n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->stage_f64, start, n * 3 * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->stage_mask, mask.data(), e->S.n_envs, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemsetAsync(e->S.status, 0, sizeof(int), e->stream));
  drlgx_launch_reset(e->S, e->stream, n, e->stage_i32, e->stage_u32, e->stage_f64);
  for (int i = 0; i < n; ++i) e->pbound[env_ids[i]] = 1;
  LaunchSel sel{0, e->S.n_envs, e->stage_mask, nullptr, 0};
  drlgx_launch_slam(e->S, e->stream, sel, e->by_capacity ? e->S.P_max : 1);
  sel.act_idx = -2;  // reductions only: the virtual map is in its untouched state
  sel.pcap = max_bound(e);
  drlgx_launch_map(e->S, e->stream, sel);
  int r = check_launch(e);
  if (r) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));  // staging buffers are reused
  return DRLGX_OK;
}

int drlgx_step(drlgx_engine *e, const double *odom_dev, const uint8_t *active_dev) {
  DRLGX_ENTER(e);
  if (!e || !odom_dev) return DRLGX_E_INVALID;
  LaunchSel sel{0, e->S.n_envs, active_dev, nullptr, 0};
  const int pb = std::min(max_bound(e) + 1, e->S.P_max);
  sel.pcap = pb;  // (the kernels size their per-pose LDS tables with the launch's bound, not with the capacity)
  for (int &v : e->pbound) v = std::min(v + 1, e->S.P_max);
  if (drlgx_step_fusable(e->S, pb) && !e->per_stage) {
    // one fused kernel per step (timer 5); timing mode 2 launches the stage kernels separately (timers 0-2)
    ScopedTimer t(e, 5);
    // more envs than CUs: simulator + SLAM fused, the map as its own launch in the two-workgroups-per-CU form (k_map_c) - at one
    // workgroup per CU the fused map stage cannot overlap anything, two map workgroups per CU cover each other's barriers
    const bool split = e->S.n_envs > e->n_cu && drlgx_map_two_per_cu(e->S, pb);
    sel.skip_map = split ? 1 : 0;
    drlgx_launch_step(e->S, e->stream, sel, odom_dev, 3, 2);
    if (split) {
      sel.skip_map = 0;
      drlgx_launch_map(e->S, e->stream, sel);
    }
  } else if (drlgx_step_arrow_fusable(e->S) && !e->per_stage) {
    ScopedTimer t(e, 5);  // longer trajectories: the same fusion around the pose-chain solver
    drlgx_launch_step_arrow(e->S, e->stream, sel, odom_dev, 3, 2);
  } else {
    {
      ScopedTimer t(e, 0);
      drlgx_launch_sim(e->S, e->stream, sel, odom_dev, 3, 2);
    }
    {
      ScopedTimer t(e, 1);
      drlgx_launch_slam(e->S, e->stream, sel, pb);
    }
    {
      ScopedTimer t(e, 2);
      drlgx_launch_map(e->S, e->stream, sel);
    }
  }
  {
    ScopedTimer t(e, 7);  // empty span: the event-pair overhead, so that callers can subtract it
  }
  return check_launch(e);
}

// One action index of the chosen plans of all envs (ExplorationEnv.step: `for a in actions: self._sim.simulate(a)`,
// scripts/envs/exploration_env.py:98-105): env i executes actions[i][action_index] while action_index < n_actions[i].
// map_last_only: the virtual map (a pure function of the SLAM state, rebuilt from the untouched map every time) is rebuilt
// at each env's LAST action only, and the marginals only where the map needs them - the state after the plan is the same.
int drlgx_step_plan(drlgx_engine *e, const double *actions_dev, const int32_t *n_actions_dev, int action_index, int map_last_only) {
  DRLGX_ENTER(e);
  if (!e || !actions_dev || !n_actions_dev || action_index < 0 || action_index >= e->S.A_max) return DRLGX_E_INVALID;
  LaunchSel sel{0, e->S.n_envs, nullptr, n_actions_dev, action_index};
  sel.map_last_only = map_last_only ? 1 : 0;
  const int pb = std::min(max_bound(e) + 1, e->S.P_max);
  sel.pcap = pb;
  for (int &v : e->pbound) v = std::min(v + 1, e->S.P_max);
  const int stride = e->S.A_max * 3;
  if (drlgx_step_fusable(e->S, pb) && !e->per_stage) {
    ScopedTimer t(e, 5);
    drlgx_launch_step(e->S, e->stream, sel, actions_dev, stride, 2);
  } else if (drlgx_step_arrow_fusable(e->S) && !e->per_stage) {
    ScopedTimer t(e, 5);
    drlgx_launch_step_arrow(e->S, e->stream, sel, actions_dev, stride, 2);
  } else {
    {
      ScopedTimer t(e, 0);
      drlgx_launch_sim(e->S, e->stream, sel, actions_dev, stride, 2);
    }
    {
      ScopedTimer t(e, 1);
      drlgx_launch_slam(e->S, e->stream, sel, pb);
    }
    {
      ScopedTimer t(e, 2);
      drlgx_launch_map(e->S, e->stream, sel);
    }
  }
  return check_launch(e);
}

// The whole loop `for a in actions: self._sim.simulate(a)` of every env (scripts/envs/exploration_env.py:98-105) in one call: env i
// executes actions[i][0 .. n_actions[i]); max_n_actions = a host-side bound of the plan lengths.  One launch when the fused
// step kernel serves every pose count the plans can reach (k_step_loop / k_step_arrow_loop: a workgroup runs its env's whole
// plan), else one drlgx_step_plan per action index.  Same results as that loop, bit for bit.
int drlgx_step_plans(drlgx_engine *e, const double *actions_dev, const int32_t *n_actions_dev, int max_n_actions, int map_last_only) {
  DRLGX_ENTER(e);
  if (!e || !actions_dev || !n_actions_dev || max_n_actions < 0 || max_n_actions > e->S.A_max) return DRLGX_E_INVALID;
  if (max_n_actions == 0) return DRLGX_OK;
  const DrlgxState &S = e->S;
  const int pbe = max_bound(e);
  int a_sw = 0;  // the leading actions the fused dense-solver step serves (as one drlgx_step_plan per action index would choose)
  while (a_sw < max_n_actions && drlgx_step_fusable(S, std::min(pbe + a_sw + 1, S.P_max))) ++a_sw;
  const bool rest_loops = a_sw == max_n_actions || drlgx_step_arrow_fusable(S);
  if (!e->la_loop || e->per_stage || !rest_loops) {
    for (int a = 0; a < max_n_actions; ++a) {
      const int r = drlgx_step_plan(e, actions_dev, n_actions_dev, a, map_last_only);
      if (r) return r;
    }
    return DRLGX_OK;
  }
  LaunchSel sel{0, S.n_envs, nullptr, n_actions_dev, 0};
  sel.map_last_only = map_last_only ? 1 : 0;
  for (int &v : e->pbound) v = std::min(v + max_n_actions, S.P_max);
  {
    ScopedTimer t(e, 5);
    if (a_sw > 0) {
      sel.pcap = std::min(pbe + 1, S.P_max);
      drlgx_launch_step_loop(S, e->stream, sel, actions_dev, S.A_max * 3, 2, a_sw);
    }
    if (a_sw < max_n_actions) {
      sel.act_idx = a_sw;
      sel.pcap = std::min(pbe + a_sw + 1, S.P_max);
      drlgx_launch_step_arrow_loop(S, e->stream, sel, actions_dev, S.A_max * 3, 2, max_n_actions);
    }
  }
  return check_launch(e);
}

// ---- staged form of the belief step: one call per call of SS2D.__init__ / SS2D.simulate (scripts/envs/pyss2d.py) -------
int drlgx_stage_reset_host(drlgx_engine *e, int n, const int32_t *env_ids, const uint32_t *seeds, const double *start) {
  DRLGX_ENTER(e);
  if (!e || n <= 0 || n > e->S.n_envs || !env_ids || !seeds || !start) return DRLGX_E_INVALID;
  for (int i = 0; i < n; ++i)
    if (env_ids[i] < 0 || env_ids[i] >= e->S.n_envs) return DRLGX_E_INVALID;
  HIPCHK(e, hipMemcpyAsync(e->stage_i32, env_ids, n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->stage_u32, seeds, n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->stage_f64, start, n * 3 * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemsetAsync(e->S.status, 0, sizeof(int), e->stream));
  drlgx_launch_reset(e->S, e->stream, n, e->stage_i32, e->stage_u32, e->stage_f64, 0);
  for (int i = 0; i < n; ++i) e->pbound[env_ids[i]] = 1;
  int r = check_launch(e);
  if (r) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return DRLGX_OK;
}
int drlgx_stage_move(drlgx_engine *e, const double *odom_dev, const uint8_t *active_dev) {
  DRLGX_ENTER(e);
  if (!e || !odom_dev) return DRLGX_E_INVALID;
  for (int &v : e->pbound) v = std::min(v + 1, e->S.P_max);
  drlgx_launch_sim_stage(e->S, e->stream, LaunchSel{0, e->S.n_envs, active_dev, nullptr, 0}, odom_dev, 0, nullptr, nullptr, nullptr);
  return check_launch(e);
}
int drlgx_stage_measure(drlgx_engine *e, const uint8_t *active_dev, int32_t *keys_dev, double *bearing_range_dev, int32_t *count_dev) {
  DRLGX_ENTER(e);
  if (!e || !keys_dev || !bearing_range_dev || !count_dev) return DRLGX_E_INVALID;
  HIPCHK(e, hipMemsetAsync(count_dev, 0, sizeof(int32_t) * e->S.n_envs, e->stream));
  drlgx_launch_sim_stage(e->S, e->stream, LaunchSel{0, e->S.n_envs, active_dev, nullptr, 0}, nullptr, 1, keys_dev, bearing_range_dev,
                         count_dev);
  return check_launch(e);
}
int drlgx_stage_add_measurements(drlgx_engine *e, const uint8_t *active_dev, const int32_t *keys_dev, const double *bearing_range_dev,
                                 const int32_t *count_dev) {
  DRLGX_ENTER(e);
  if (!e || !keys_dev || !bearing_range_dev || !count_dev) return DRLGX_E_INVALID;
  drlgx_launch_add_measurements(e->S, e->stream, LaunchSel{0, e->S.n_envs, active_dev, nullptr, 0}, keys_dev, bearing_range_dev, count_dev);
  return check_launch(e);
}
int drlgx_stage_optimize(drlgx_engine *e, const uint8_t *active_dev) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  drlgx_launch_slam(e->S, e->stream, LaunchSel{0, e->S.n_envs, active_dev, nullptr, 0}, max_bound(e));
  return check_launch(e);
}
int drlgx_stage_update_map(drlgx_engine *e, const uint8_t *active_dev, int rebuild) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  drlgx_launch_map(e->S, e->stream, LaunchSel{0, e->S.n_envs, active_dev, nullptr, rebuild ? 0 : -2});
  return check_launch(e);
}

// FastMarginals2::update for candidate action lists (k_fm2.hip)
static const int kFm2Chunk = 128, kFm2MaxMeas = 256;
int drlgx_fm2_update(drlgx_engine *e, int n_cand, const int32_t *cand_env_dev, const double *actions_dev, const int32_t *n_actions_dev,
                     double *cov_out_dev, int out_stride_poses, int32_t *n_out_dev) {
  DRLGX_ENTER(e);
  if (!e || n_cand < 0 || !cand_env_dev || !actions_dev || !n_actions_dev || !cov_out_dev || !n_out_dev ||
      out_stride_poses < e->S.P_max + e->S.A_max)
    return DRLGX_E_INVALID;
  if (n_cand == 0) return DRLGX_OK;
  const DrlgxState &S = e->S;
  if (!e->fm2_sig) {
    const size_t nmax = (size_t)3 * S.P_max + 2 * S.L_max;
    e->fm2_sig_stride = nmax * nmax;
    e->fm2_scratch_stride = drlgx_fm2_scratch_doubles(S, kFm2MaxMeas);
    int r;
    if ((r = dev_alloc(e, &e->fm2_sig, e->fm2_sig_stride * (size_t)S.n_envs))) return r;
    if ((r = dev_alloc(e, &e->fm2_scratch, e->fm2_scratch_stride * (size_t)kFm2Chunk))) return r;
    if ((r = dev_alloc(e, &e->fm2_iscratch, (size_t)2 * kFm2MaxMeas * kFm2Chunk))) return r;
    if ((r = dev_alloc(e, &e->fm2_slot, (size_t)S.n_envs))) return r;
    std::vector<int32_t> ident(S.n_envs);
    for (int i = 0; i < S.n_envs; ++i) ident[i] = i;
    HIPCHK(e, hipMemcpyAsync(e->fm2_slot, ident.data(), ident.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
  }
  drlgx_launch_fm2_prior(S, e->stream, nullptr, S.n_envs, e->fm2_sig, e->fm2_sig_stride, 3 * S.P_max + 2 * S.L_max);
  for (int c0 = 0; c0 < n_cand; c0 += kFm2Chunk)
    drlgx_launch_fm2_update(S, e->stream, c0, std::min(kFm2Chunk, n_cand - c0), cand_env_dev, e->fm2_slot, actions_dev, n_actions_dev,
                            e->fm2_sig, e->fm2_sig_stride, e->fm2_scratch, e->fm2_scratch_stride, e->fm2_iscratch, kFm2MaxMeas,
                            cov_out_dev, out_stride_poses, n_out_dev);
  return check_launch(e);
}

int drlgx_stage_set_prior_information_host(drlgx_engine *e, int env, const double *information9) {
  DRLGX_ENTER(e);
  if (!e || env < 0 || env >= e->S.n_envs || !information9) return DRLGX_E_INVALID;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < r; ++c)
      if (information9[3 * r + c] != information9[3 * c + r]) return DRLGX_E_INVALID;  // symmetric, as an information matrix is
  HIPCHK(e, hipMemcpyAsync(e->S.prior + (size_t)env * DRLGX_PRIOR_STRIDE + 4, information9, 9 * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return DRLGX_OK;
}

int drlgx_stage_set_prior_pose_host(drlgx_engine *e, int env, const double *xytheta) {
  DRLGX_ENTER(e);
  if (!e || env < 0 || env >= e->S.n_envs || !xytheta) return DRLGX_E_INVALID;
  const DrlgxState &S = e->S;
  // (between the staged reset and the first measurement: one pose, no factor)
  int cnt[DRLGX_CNT_STRIDE];
  HIPCHK(e, hipMemcpyAsync(cnt, S.cnt + (size_t)env * DRLGX_CNT_STRIDE, sizeof(cnt), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (cnt[C_P] != 1 || cnt[C_M] != 0) return DRLGX_E_INVALID;
  const double p4[4] = {xytheta[0], xytheta[1], std::cos(xytheta[2]), std::sin(xytheta[2])};
  HIPCHK(e, hipMemcpyAsync(S.prior + (size_t)env * DRLGX_PRIOR_STRIDE, p4, sizeof(p4), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(S.th_pose + (size_t)env * S.P_max * 4, p4, sizeof(p4), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(S.est_pose + (size_t)env * S.P_max * 4, p4, sizeof(p4), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return DRLGX_OK;
}

int drlgx_set_fixed_landmarks_host(drlgx_engine *e, int n_fixed, const double *xy) {
  DRLGX_ENTER(e);
  if (!e || n_fixed < 0 || n_fixed > e->S.cfg.num_landmarks || (n_fixed > 0 && !xy)) return DRLGX_E_INVALID;
  if (!e->fixed_lm_dev) {
    int r = dev_alloc(e, &e->fixed_lm_dev, (size_t)std::max(e->S.cfg.num_landmarks, 1) * 2);
    if (r) return r;
  }
  if (n_fixed > 0) {
    HIPCHK(e, hipMemcpyAsync(e->fixed_lm_dev, xy, (size_t)n_fixed * 2 * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));  // (the host buffer is the caller's)
  }
  e->S.fixed_lm = n_fixed > 0 ? e->fixed_lm_dev : nullptr;  // (the state struct travels with every launch: the next reset sees it)
  e->S.n_fixed = n_fixed;
  return DRLGX_OK;
}

int drlgx_set_planner_parameter(drlgx_engine *e, double angle_weight, double distance_weight0, double distance_weight1,
                                double occupancy_threshold, double max_edge_length, int algorithm) {
  DRLGX_ENTER(e);
  if (!e || !(max_edge_length > 0) || algorithm < 0 || algorithm > 3) return DRLGX_E_INVALID;
  drlgx_config &c = e->S.cfg;  // (the state struct is passed to every launch by value: the next launch sees the new values)
  c.angle_weight = angle_weight;
  c.distance_weight0 = distance_weight0;
  c.distance_weight1 = distance_weight1;
  c.occupancy_threshold = occupancy_threshold;
  c.max_edge_length = max_edge_length;
  c.algorithm = algorithm;
  return DRLGX_OK;
}

int drlgx_utility(drlgx_engine *e, const double *dist_dev, double *out_dev) {
  DRLGX_ENTER(e);
  if (!e || !out_dev) return DRLGX_E_INVALID;
  drlgx_launch_utility(e->S, e->stream, dist_dev, out_dev, 0);
  return check_launch(e);
}
int drlgx_uncertainty_em(drlgx_engine *e, int algorithm, double *out_dev) {
  DRLGX_ENTER(e);
  if (!e || !out_dev || (algorithm != DRLGX_ALG_EM_AOPT && algorithm != DRLGX_ALG_EM_DOPT)) return DRLGX_E_INVALID;
  drlgx_launch_utility(e->S, e->stream, nullptr, out_dev, algorithm == DRLGX_ALG_EM_DOPT ? 3 : 2);
  return check_launch(e);
}
int drlgx_explored(drlgx_engine *e, double *out_dev) {
  DRLGX_ENTER(e);
  if (!e || !out_dev) return DRLGX_E_INVALID;
  drlgx_launch_utility(e->S, e->stream, nullptr, out_dev, 1);
  return check_launch(e);
}

int drlgx_metrics(drlgx_engine *e, double sigma0, double *out_dev) {
  DRLGX_ENTER(e);
  if (!e || !out_dev) return DRLGX_E_INVALID;
  drlgx_launch_metrics(e->S, e->stream, sigma0, out_dev);
  return check_launch(e);
}
int drlgx_cov_array(drlgx_engine *e, double *length_dev, double *angle_dev) {
  DRLGX_ENTER(e);
  if (!e || !length_dev || !angle_dev) return DRLGX_E_INVALID;
  drlgx_launch_cov_array(e->S, e->stream, length_dev, angle_dev);
  return check_launch(e);
}

int drlgx_line_plan(drlgx_engine *e, int n_cand, const int32_t *cand_env_dev, const double *goal_dev,
                    double *actions_dev, int32_t *n_actions_dev) {
  DRLGX_ENTER(e);
  if (!e || n_cand < 0 || !cand_env_dev || !goal_dev || !actions_dev || !n_actions_dev) return DRLGX_E_INVALID;
  if (n_cand == 0) return DRLGX_OK;
  drlgx_launch_line_plan(e->S, e->stream, n_cand, cand_env_dev, goal_dev, actions_dev, n_actions_dev);
  return check_launch(e);
}

int drlgx_lookahead(drlgx_engine *e, int n_cand, const int32_t *cand_env_dev, const double *actions_dev,
                    const int32_t *n_actions_dev, double *rewards_dev) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  return drlgx_lookahead_bounded(e, n_cand, cand_env_dev, actions_dev, n_actions_dev, e->S.A_max, rewards_dev);
}

int drlgx_lookahead_bounded(drlgx_engine *e, int n_cand, const int32_t *cand_env_dev, const double *actions_dev,
                            const int32_t *n_actions_dev, int max_n_actions, double *rewards_dev) {
  DRLGX_ENTER(e);
  if (!e || n_cand < 0 || !cand_env_dev || !actions_dev || !n_actions_dev || !rewards_dev || max_n_actions < 1 ||
      max_n_actions > e->S.A_max)
    return DRLGX_E_INVALID;
  if (n_cand == 0) return DRLGX_OK;
  if (e->S.n_roll < 1) return DRLGX_E_CAPACITY;
  const DrlgxState &S = e->S;
  const int base0 = S.n_envs, roll0 = 2 * S.n_envs;
  const int pbe = max_bound(e);  // rollouts start from their env's trajectory and add one pose per action
  {
    ScopedTimer t(e, 3);
    // deep copy env -> base, SLAM2D::set_copy_isam (re-base at the best estimate + one batch update)
    drlgx_launch_copy(e->fields_dev, (int)e->fields.size(), e->stream, S.n_envs, nullptr, nullptr, 0, base0, 3, e->S.cnt);
    drlgx_launch_rebase(S, e->stream, base0, S.n_envs);
  }
  {
    ScopedTimer t(e, 1);
    drlgx_launch_slam(S, e->stream, LaunchSel{base0, S.n_envs, nullptr, nullptr, 0}, pbe);
  }
  // candidates beyond the rollout capacity are processed in successive waves over the same rollout instances
  for (int c0 = 0; c0 < n_cand; c0 += S.n_roll) {
    const int nc = std::min(S.n_roll, n_cand - c0);
    const int32_t *ce = cand_env_dev + c0;
    const double *act = actions_dev + (size_t)c0 * S.A_max * 3;
    const int32_t *na = n_actions_dev + c0;
    {
      ScopedTimer t(e, 3);
      drlgx_launch_copy(e->fields_dev, (int)e->fields.size(), e->stream, nc, ce, nullptr, base0, roll0, 3, e->S.cnt,
                        &S);  // (with the base solve's covariance panel)
      drlgx_launch_fix_rollouts(S, e->stream, nc, ce, roll0);
    }
    // Whole action lists per launch (k_step_loop / k_step_arrow_loop).  A launch per action index picks its kernel from the pose
    // bound of THAT action (the fused dense-solver step while it serves the bound, then the step around the pose-chain solver);
    // the loop form keeps that choice action by action - the leading actions [0, a_sw) in the dense loop kernel, the rest in the
    // pose-chain one - because the two solvers round differently and the reference's integer worlds hold cells at exactly
    // max_range from a pose: one ulp of a pose decides them, and with them O(0.1) of a reward.
    int a_sw = 0;
    while (a_sw < max_n_actions && drlgx_step_fusable(S, std::min(pbe + a_sw + 1, S.P_max))) ++a_sw;
    const bool rest_loops = a_sw == max_n_actions || drlgx_step_arrow_fusable(S);
    int a_begin = 0;  // first action index still to be launched one by one
    if (e->la_loop && !e->per_stage && (a_sw > 0 || rest_loops)) {
      LaunchSel sel{roll0, nc, nullptr, na, 0};
      sel.map_last_only = 1;
      if (e->la_presim && rest_loops) {  // (only when EVERY action is replayed: k_presim leaves the ground truth and the streams of the LAST action)
        // the simulator of every rollout for its whole list first (one wave per rollout), its log replayed action by action
        const size_t entry = drlgx_simlog_entry_bytes(S), roll = entry * (size_t)S.A_max;
        if (!e->simlog_dev && dev_alloc(e, &e->simlog_dev, roll * (size_t)S.n_roll) != DRLGX_OK) {
          (void)hipGetLastError();
          e->last_error.clear();
          e->la_presim = false;  // (no room for the log: the rollouts simulate inside their steps)
        } else {
          ScopedTimer t(e, 0);
          drlgx_launch_presim(S, e->stream, sel, act, S.A_max * 3, 1, max_n_actions, e->simlog_dev, roll, (int)entry);
          sel.simlog = e->simlog_dev;
          sel.simlog_roll = roll;
          sel.simlog_act = (int)entry;
        }
      }
      ScopedTimer t(e, 5);
      if (a_sw > 0) {
        sel.act_idx = 0;
        sel.pcap = std::min(pbe + 1, S.P_max);
        drlgx_launch_step_loop(S, e->stream, sel, act, S.A_max * 3, 1, a_sw);
      }
      a_begin = a_sw;
      if (a_sw < max_n_actions && rest_loops) {
        sel.act_idx = a_sw;
        sel.pcap = std::min(pbe + a_sw + 1, S.P_max);
        drlgx_launch_step_arrow_loop(S, e->stream, sel, act, S.A_max * 3, 1, max_n_actions);
        a_begin = max_n_actions;
      }
    }
    for (int a = a_begin; a < max_n_actions; ++a) {
      LaunchSel sel{roll0, nc, nullptr, na, a};
      sel.map_last_only = 1;
      const int pb = std::min(pbe + a + 1, S.P_max);
      sel.pcap = pb;
      if (drlgx_step_fusable(S, pb) && !e->per_stage) {
        ScopedTimer t(e, 5);
        drlgx_launch_step(S, e->stream, sel, act, S.A_max * 3, 1);
        continue;
      }
      if (drlgx_step_arrow_fusable(e->S) && !e->per_stage) {
        ScopedTimer t(e, 5);
        drlgx_launch_step_arrow(S, e->stream, sel, act, S.A_max * 3, 1);
        continue;
      }
      {
        ScopedTimer t(e, 0);
        drlgx_launch_sim(S, e->stream, sel, act, S.A_max * 3, 1);
      }
      {
        ScopedTimer t(e, 1);
        drlgx_launch_slam(S, e->stream, sel, pb);
      }
      {
        ScopedTimer t(e, 2);
        drlgx_launch_map(S, e->stream, sel);
      }
    }
    drlgx_launch_rewards(S, e->stream, nc, ce, roll0, rewards_dev + c0);
  }
  return check_launch(e);
}

// ---- graph export ------------------------------------------------------------------------------
int drlgx_graph_capacity(const drlgx_engine *e, int *max_nodes, int *max_edges, int *max_frontier) {
  if (!e) return DRLGX_E_INVALID;
  const DrlgxState &S = e->S;
  if (max_nodes) *max_nodes = S.n_envs * (2 * S.L_max + S.P_max + 1);
  if (max_edges) *max_edges = S.n_envs * 2 * (S.M_max + S.P_max + S.L_max + 1);
  if (max_frontier) *max_frontier = S.L_max + 1;
  return DRLGX_OK;
}

int drlgx_graph(drlgx_engine *e, int32_t *node_off_dev, int32_t *edge_off_dev, float *x_dev, int64_t *edge_index_dev,
                float *edge_attr_dev, int32_t *n_frontier_dev, double *frontier_xy_dev, int32_t *nearest_frontier_node_dev) {
  DRLGX_ENTER(e);
  if (!e || !node_off_dev || !edge_off_dev || !x_dev || !edge_index_dev || !edge_attr_dev || !n_frontier_dev ||
      !frontier_xy_dev || !nearest_frontier_node_dev)
    return DRLGX_E_INVALID;
  ScopedTimer t(e, 4);
  drlgx_launch_graph(e->S, e->stream, e->graph_gi, e->graph_gi_stride, node_off_dev, edge_off_dev, x_dev, edge_index_dev,
                     edge_attr_dev, n_frontier_dev, frontier_xy_dev, nearest_frontier_node_dev, e->S.L_max + 1);
  return check_launch(e);
}

// ---- getters -----------------------------------------------------------------------------------
static int fetch(drlgx_engine *e, void *dst, const void *src, size_t bytes) {
  HIPCHK(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
  return DRLGX_OK;
}
#define INST_OK(e, inst) ((e) && (inst) >= 0 && (inst) < (e)->S.n_inst)

int drlgx_get_counts_host(drlgx_engine *e, int inst, int32_t out[5]) {
  DRLGX_ENTER(e);
  if (!INST_OK(e, inst) || !out) return DRLGX_E_INVALID;
  int c[DRLGX_CNT_STRIDE];
  int r = fetch(e, c, e->S.cnt + (size_t)inst * DRLGX_CNT_STRIDE, sizeof(c));
  if (r) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  out[0] = c[C_P]; out[1] = c[C_L]; out[2] = c[C_M]; out[3] = c[C_STEP]; out[4] = c[C_ISAM];
  return DRLGX_OK;
}

int drlgx_counts(drlgx_engine *e, int32_t *counts_dev) {
  DRLGX_ENTER(e);
  if (!e || !counts_dev) return DRLGX_E_INVALID;
  HIPCHK(e, hipMemcpy2DAsync(counts_dev, 5 * sizeof(int32_t), e->S.cnt, DRLGX_CNT_STRIDE * sizeof(int32_t), 5 * sizeof(int32_t),
                             (size_t)e->S.n_envs, hipMemcpyDeviceToDevice, e->stream));
  return DRLGX_OK;
}

int drlgx_get_poses_host(drlgx_engine *e, int inst, double *xytheta, double *information) {
  DRLGX_ENTER(e);
  int32_t c[5];
  int r = drlgx_get_counts_host(e, inst, c);
  if (r) return r;
  const DrlgxState &S = e->S;
  const int P = c[0];
  std::vector<double> ep((size_t)P * 4), pi((size_t)P * 6);
  if ((r = fetch(e, ep.data(), S.est_pose + (size_t)inst * S.P_max * 4, ep.size() * 8))) return r;
  if ((r = fetch(e, pi.data(), S.pose_info + (size_t)inst * S.P_max * 6, pi.size() * 8))) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (int i = 0; i < P; ++i) {
    if (xytheta) {
      xytheta[3 * i] = ep[4 * i];
      xytheta[3 * i + 1] = ep[4 * i + 1];
      xytheta[3 * i + 2] = std::atan2(ep[4 * i + 3], ep[4 * i + 2]);
    }
    if (information) {
      const double *s = &pi[6 * i];
      double *o = information + 9 * i;
      o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
      o[3] = s[1]; o[4] = s[3]; o[5] = s[4];
      o[6] = s[2]; o[7] = s[4]; o[8] = s[5];
    }
  }
  return DRLGX_OK;
}

static int sorted_slots(drlgx_engine *e, int inst, int L, std::vector<int> &keys, std::vector<int> &order) {
  const DrlgxState &S = e->S;
  keys.resize(L);
  int r = fetch(e, keys.data(), S.lm_key + (size_t)inst * S.L_max, (size_t)L * sizeof(int));
  if (r) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  order.resize(L);
  for (int j = 0; j < L; ++j) order[j] = j;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return keys[a] < keys[b]; });
  return DRLGX_OK;
}

int drlgx_get_landmarks_host(drlgx_engine *e, int inst, int32_t *keys, double *xy, double *information) {
  DRLGX_ENTER(e);
  int32_t c[5];
  int r = drlgx_get_counts_host(e, inst, c);
  if (r) return r;
  const DrlgxState &S = e->S;
  const int L = c[1];
  std::vector<int> k, ord;
  if ((r = sorted_slots(e, inst, L, k, ord))) return r;
  std::vector<double> el((size_t)L * 2), li((size_t)L * 3);
  if ((r = fetch(e, el.data(), S.est_lm + (size_t)inst * S.L_max * 2, el.size() * 8))) return r;
  if ((r = fetch(e, li.data(), S.lm_info + (size_t)inst * S.L_max * 3, li.size() * 8))) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (int n = 0; n < L; ++n) {
    int j = ord[n];
    if (keys) keys[n] = k[j];
    if (xy) {
      xy[2 * n] = el[2 * j];
      xy[2 * n + 1] = el[2 * j + 1];
    }
    if (information) {
      information[4 * n] = li[3 * j];
      information[4 * n + 1] = li[3 * j + 1];
      information[4 * n + 2] = li[3 * j + 1];
      information[4 * n + 3] = li[3 * j + 2];
    }
  }
  return DRLGX_OK;
}

int drlgx_get_cov_traces_host(drlgx_engine *e, int inst, double *lm_trace, double *pose_trace) {
  DRLGX_ENTER(e);
  int32_t c[5];
  int r = drlgx_get_counts_host(e, inst, c);
  if (r) return r;
  const DrlgxState &S = e->S;
  const int P = c[0], L = c[1];
  std::vector<int> k, ord;
  if ((r = sorted_slots(e, inst, L, k, ord))) return r;
  std::vector<double> lt(L), pt(P);
  if ((r = fetch(e, lt.data(), S.lm_tr + (size_t)inst * S.L_max, (size_t)L * 8))) return r;
  if ((r = fetch(e, pt.data(), S.pose_tr + (size_t)inst * S.P_max, (size_t)P * 8))) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (lm_trace)
    for (int n = 0; n < L; ++n) lm_trace[n] = lt[ord[n]];
  if (pose_trace)
    for (int i = 0; i < P; ++i) pose_trace[i] = pt[i];
  return DRLGX_OK;
}

int drlgx_vm_shape(const drlgx_engine *e, int *rows, int *cols) {
  if (!e) return DRLGX_E_INVALID;
  if (rows) *rows = e->S.rows;
  if (cols) *cols = e->S.cols;
  return DRLGX_OK;
}

int drlgx_get_virtual_map_host(drlgx_engine *e, int inst, double *prob, double *info, double *cov_trace,
                               uint8_t *updated) {
  DRLGX_ENTER(e);
  if (!INST_OK(e, inst)) return DRLGX_E_INVALID;
  const DrlgxState &S = e->S;
  const size_t V = S.V;
  std::vector<double> pl(3 * V);
  int r;
  if (prob && (r = fetch(e, prob, S.vm_prob + (size_t)inst * V, V * 8))) return r;
  if ((r = fetch(e, pl.data(), S.vm_info + (size_t)inst * 3 * V, 3 * V * 8))) return r;
  if (updated && (r = fetch(e, updated, S.vm_upd + (size_t)inst * S.Vu, V))) return r;
  if (cov_trace && (r = fetch(e, cov_trace, S.vm_tr + (size_t)inst * V, V * 8))) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (info)
    for (size_t v = 0; v < V; ++v) {
      const double a = pl[v], b = pl[V + v], d = pl[2 * V + v];
      info[4 * v] = a; info[4 * v + 1] = b; info[4 * v + 2] = b; info[4 * v + 3] = d;
    }
  return DRLGX_OK;
}

int drlgx_get_ground_truth_host(drlgx_engine *e, int inst, double *vehicle_xytheta, double *landmarks_xy) {
  DRLGX_ENTER(e);
  if (!INST_OK(e, inst)) return DRLGX_E_INVALID;
  const DrlgxState &S = e->S;
  double gp[4];
  int parent = 0, r;
  if ((r = fetch(e, gp, S.gt_pose + (size_t)inst * 4, sizeof(gp)))) return r;
  if ((r = fetch(e, &parent, S.parent + inst, sizeof(int)))) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (vehicle_xytheta) {
    vehicle_xytheta[0] = gp[0];
    vehicle_xytheta[1] = gp[1];
    vehicle_xytheta[2] = std::atan2(gp[3], gp[2]);
  }
  if (landmarks_xy && S.cfg.num_landmarks > 0) {
    if ((r = fetch(e, landmarks_xy, S.gt_lm + (size_t)parent * S.LG * 2, (size_t)S.cfg.num_landmarks * 2 * 8))) return r;
    HIPCHK(e, hipStreamSynchronize(e->stream));
  }
  return DRLGX_OK;
}

int drlgx_get_factors_host(drlgx_engine *e, int inst, int32_t *pose, int32_t *key, double *bearing, double *range) {
  DRLGX_ENTER(e);
  int32_t c[5];
  int r = drlgx_get_counts_host(e, inst, c);
  if (r) return r;
  const DrlgxState &S = e->S;
  const int L = c[1], M = c[2];
  std::vector<int> keys(L), mp(M), ml(M);
  std::vector<double> br((size_t)M * 2);
  if ((r = fetch(e, keys.data(), S.lm_key + (size_t)inst * S.L_max, (size_t)L * 4))) return r;
  if ((r = fetch(e, mp.data(), S.meas_pose + (size_t)inst * S.M_max, (size_t)M * 4))) return r;
  if ((r = fetch(e, ml.data(), S.meas_lm + (size_t)inst * S.M_max, (size_t)M * 4))) return r;
  if ((r = fetch(e, br.data(), S.meas_br + (size_t)inst * S.M_max * 2, (size_t)M * 16))) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (int m = 0; m < M; ++m) {
    if (pose) pose[m] = mp[m];
    if (key) key[m] = keys[ml[m]];
    if (bearing) bearing[m] = br[2 * m];
    if (range) range[m] = br[2 * m + 1];
  }
  return DRLGX_OK;
}

// SLAM2D::adjacency_degree_get (SLAM2D.cpp:198-273) assembled from the exported factor list.
int drlgx_get_adjacency_host(drlgx_engine *e, int inst, double *adjacency, double *features) {
  DRLGX_ENTER(e);
  int32_t c[5];
  int r = drlgx_get_counts_host(e, inst, c);
  if (r) return r;
  const DrlgxState &S = e->S;
  const int P = c[0], L = c[1], M = c[2], N = P + L;
  std::vector<int> k, ord;
  if ((r = sorted_slots(e, inst, L, k, ord))) return r;
  std::vector<int> node_of_slot(L);
  for (int n = 0; n < L; ++n) node_of_slot[ord[n]] = n;
  std::vector<int> mp(M), ml(M);
  std::vector<double> br((size_t)M * 2), odo((size_t)P * 4), lt(L), pt(P);
  if ((r = fetch(e, mp.data(), S.meas_pose + (size_t)inst * S.M_max, (size_t)M * 4))) return r;
  if ((r = fetch(e, ml.data(), S.meas_lm + (size_t)inst * S.M_max, (size_t)M * 4))) return r;
  if ((r = fetch(e, br.data(), S.meas_br + (size_t)inst * S.M_max * 2, (size_t)M * 16))) return r;
  if ((r = fetch(e, odo.data(), S.odo + (size_t)inst * S.P_max * 4, (size_t)P * 32))) return r;
  if ((r = fetch(e, lt.data(), S.lm_tr + (size_t)inst * S.L_max, (size_t)L * 8))) return r;
  if ((r = fetch(e, pt.data(), S.pose_tr + (size_t)inst * S.P_max, (size_t)P * 8))) return r;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (adjacency) std::fill(adjacency, adjacency + (size_t)N * N, 0.0);
  if (features) std::fill(features, features + N, 0.0);
  for (int i = 0; i + 1 < P; ++i) {
    double d = std::sqrt(std::pow(odo[4 * i], 2) + std::pow(odo[4 * i + 1], 2)) + 0.001;
    int a = L + i, b = L + i + 1;
    if (adjacency) adjacency[(size_t)a * N + b] = adjacency[(size_t)b * N + a] = d;
    if (features) {
      features[a] = pt[i];
      features[b] = pt[i + 1];
    }
  }
  for (int m = 0; m < M; ++m) {
    int a = L + mp[m], b = node_of_slot[ml[m]];
    if (adjacency) adjacency[(size_t)a * N + b] = adjacency[(size_t)b * N + a] = br[2 * m + 1];
    if (features) {
      features[a] = pt[mp[m]];
      features[b] = lt[ml[m]];
    }
  }
  return DRLGX_OK;
}

int drlgx_get_landmark_order_host(const drlgx_engine *e, int32_t *order) {
  if (!e || !order) return DRLGX_E_INVALID;
  for (int i = 0; i < e->S.cfg.num_landmarks; ++i) order[i] = e->lm_order[i];
  return DRLGX_OK;
}

// ---- snapshots ---------------------------------------------------------------------------------
// Snapshots are extra instances in the same HBM arrays: one copy kernel moves every field.
int drlgx_snapshot(drlgx_engine *e, int slot) {
  DRLGX_ENTER(e);
  if (!e || slot < 0 || slot >= e->S.cfg.max_snapshots) return DRLGX_E_INVALID;
  const DrlgxState &S = e->S;
  ScopedTimer t(e, 3);
  drlgx_launch_copy(e->fields_dev, (int)e->fields.size(), e->stream, S.n_envs, nullptr, nullptr, 0,
                    2 * S.n_envs + S.n_roll + slot * S.n_envs, 0, e->S.cnt, &S);
  e->snap_pbound[slot] = e->pbound;
  return check_launch(e);
}

int drlgx_restore(drlgx_engine *e, int slot) {
  DRLGX_ENTER(e);
  if (!e || slot < 0 || slot >= e->S.cfg.max_snapshots) return DRLGX_E_INVALID;
  const DrlgxState &S = e->S;
  ScopedTimer t(e, 3);
  drlgx_launch_copy(e->fields_dev, (int)e->fields.size(), e->stream, S.n_envs, nullptr, nullptr,
                    2 * S.n_envs + S.n_roll + slot * S.n_envs, 0, 0, e->S.cnt, &S);
  e->pbound = e->snap_pbound[slot];
  return check_launch(e);
}

// ---- development aid: in-kernel phase stamps of block 0 ------------------------------------------
int drlgx_debug_phase_clocks_host(drlgx_engine *e, int arm, int64_t out[64]) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->S.prof_block = (arm >> 8) & 0xffff;  // (arm >> 8: the workgroup whose phases are stamped; 0 by default)
  // (arm & 2: the second bank of 64 stamps - per-wave stamps of one sweep block step)
  // (arm & 4: 1024 stamps - out must hold them: banks 0, 1 and the per-workgroup start / end stamps of k_step from 128 on)
  // (arm & 4 reads the whole buffer from its start, whatever arm & 2 says: the buffer holds exactly 1024 stamps)
  if (out && e->S.prof)
    HIPCHK(e, hipMemcpy(out, e->S.prof + ((arm & 4) ? 0 : ((arm & 2) ? 64 : 0)), ((arm & 4) ? 1024 : 64) * sizeof(long long), hipMemcpyDeviceToHost));
  if (arm && !e->S.prof) {
    long long *p = nullptr;
    int r = dev_alloc(e, &p, 1024);
    if (r) return r;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    e->S.prof = p;
  } else if (!arm) {
    e->S.prof = nullptr;
  }
  return DRLGX_OK;
}

// ---- incremental belief update: how many SLAM updates took the rank-k path / the full solve ------------------
int drlgx_inc_stats_host(drlgx_engine *e, int64_t out[2], int reset) {
  DRLGX_ENTER(e);
  if (!e || !out) return DRLGX_E_INVALID;
  out[0] = out[1] = -1;  // -1: the incremental path is disabled (DRLGX_INCREMENTAL=0 or the panels exceed the memory budget)
  if (!e->S.inc_stats) return DRLGX_OK;
  unsigned long long v[2] = {0, 0};
  HIPCHK(e, hipMemcpyAsync(v, e->S.inc_stats, sizeof(v), hipMemcpyDeviceToHost, e->stream));
  if (reset) HIPCHK(e, hipMemsetAsync(e->S.inc_stats, 0, sizeof(v), e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  out[0] = (int64_t)v[0];
  out[1] = (int64_t)v[1];
  return DRLGX_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int drlgx_timing_enable(drlgx_engine *e, int on) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  e->timing = on != 0;
  e->per_stage = on == 2;
  return DRLGX_OK;
}

int drlgx_timing_read_host(drlgx_engine *e, double ms[DRLGX_N_TIMERS], int64_t launches[DRLGX_N_TIMERS]) {
  DRLGX_ENTER(e);
  if (!e) return DRLGX_E_INVALID;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (auto &sp : e->spans) {
    float t = 0;
    if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess && sp.id >= 0 && sp.id < DRLGX_N_TIMERS) {
      e->t_ms[sp.id] += t;
      e->t_n[sp.id] += 1;
    }
    e->free_events.push_back(sp.a);
    e->free_events.push_back(sp.b);
  }
  e->spans.clear();
  for (int i = 0; i < DRLGX_N_TIMERS; ++i) {
    if (ms) ms[i] = e->t_ms[i];
    if (launches) launches[i] = e->t_n[i];
    e->t_ms[i] = 0;
    e->t_n[i] = 0;
  }
  return DRLGX_OK;
}

}  // extern "C"
