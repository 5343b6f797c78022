// Fused belief step: simulate -> SLAM -> virtual map for one instance in ONE kernel (one 512-thread workgroup per
// instance).  The three stages are independent between instances, so fusing them removes two launches per step and,
// more importantly, the two grid-wide joins: an instance no longer waits for the slowest workgroup of the previous stage
// (launch + ramp + tail cost ~3-5 us per kernel at 256 workgroups).  Inside the workgroup nothing one stage hands the
// next goes through HBM:
//   * wave 0 runs the (single-wave) simulator; waves 1-7 run the SLAM FRONT END on the state before the step beside it
//     (old factors linearised, tables, block assembly; kslam::SlamCtx::front behind a 7-wave software barrier);
//   * the simulator leaves what it appends (factors, landmarks, counts) in LDS for the SLAM back end (kslam::SimBox, the
//     mailbox `sub_cnt`), its two random streams are written back by the other seven waves;
//   * the SLAM stage writes the pose estimates / information blocks into the map stage's LDS pose tables and leaves the
//     landmark estimates where the map stage reads them; the map stage's ladder tables are fetched before the SLAM stage.
//
// This translation unit is a unity build of the three stage files (their kernels stay available for the paths that use
// them alone: reset, the look-ahead base solve, capacities beyond the LDS-resident SLAM kernel).  It is compiled with
// -ffp-contract=off (simulator and map decisions must round like the CPU reference); k_slam.hip re-enables contraction
// for its own functions.
#include "drlgx_dev.h"

#include "k_sim.hip"
#include "k_slam.hip"
#include "k_map.hip"

namespace kstep {

// LDS of the simulator wave (two mt19937 streams, the normal variates and the in-range list of a measure() call); the SLAM
// stage of k_step carves behind it because its front end runs WHILE the simulator wave works
__host__ __device__ inline size_t sim_lds_bytes(int LG, int P_max) { return drlgx_sim_lds_bytes(LG, P_max); }

template <int FT>
__device__ __forceinline__ void step_once(const DrlgxState &S, const LaunchSel &sel, const double *odom, int odom_stride, int n_measure,
                                          int lds_bytes, int map_chunk) {
  static_assert(kslam::kThreads == kmap::kThreads, "the fused kernel runs both stages with one workgroup size");
  extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
  const int tid = drlgx_tid();
  const int bi = drlgx_bid();
  if (S.prof && tid == 0 && bi < 448) S.prof[128 + 2 * bi] = wall_clock64();  // (dev aid: per-workgroup start / end)
  // ---- what the SLAM front end needs is read before the simulator wave starts to change it ----
  const int pc = sel.cap(S.P_max);  // the launch's pose bound sizes the per-pose LDS tables (LaunchSel::pcap)
  const size_t sim_bytes = sim_lds_bytes(S.LG, pc);
  int *sub_cnt = reinterpret_cast<int *>(step_smem + sim_bytes - 16);
  // what the simulator wave appends, left in LDS for the SLAM stage (ksim::measure)
  double *sim_dyn = reinterpret_cast<double *>(step_smem + 2 * DRLGX_MT_STRIDE * sizeof(uint32_t));
  double *lmbox = reinterpret_cast<double *>(step_smem + sim_bytes - 16 - (size_t)2 * S.LG * 8);
  const kslam::SimBox box{(n_measure == 2 && !sel.simlog) ? sim_dyn : nullptr, reinterpret_cast<const int *>(sim_dyn + 2 * S.LG + 2), lmbox};
  // (the map stage's ladder tables: fetched now, stored to its LDS after the SLAM stage)
  kmap::LadderEntry lo{S.lo_ntab <= kslam::kThreads, 0.0, 0u, -1, 0, 0};
  if (lo.have && tid < S.lo_ntab) {
    lo.pv = S.lo_pv[tid];
    lo.tr = reinterpret_cast<const uint32_t *>(S.lo_tr)[tid];
  }
  kslam::SlamCtx ctx;
  bool pre = false, accepted = false, inc_try = false;
  double od3[3] = {0, 0, 0};
  int P0 = 0, L0 = 0, M0 = 0, isam = 0, pan_L0 = -1, pan_M0 = -1;
  // (what the simulator wave reads that does not depend on the pose count is requested with the prelude's loads: ksim::SimPre)
  ksim::SimPre spre;
  spre.have = tid < 64 && !sel.simlog && sel.on(bi);
  if (sel.on(bi)) {
    const int inst = sel.base + bi;
    const int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
    P0 = cnt[C_P]; L0 = cnt[C_L]; M0 = cnt[C_M]; isam = cnt[C_ISAM];
    // (the covariance panel's header with the same round trip: what inc_precheck decides on)
    int meta0 = 0, meta1 = 0;
    if (S.jc) {
      const int4 mv = *reinterpret_cast<const int4 *>(kslam::inc_meta(S, inst));  // {valid, poses, landmarks, factors}
      meta0 = mv.x & 0x7fffffff;
      meta1 = mv.y;
      pan_L0 = mv.z;
      pan_M0 = mv.w;
    }
    const double *od = odom + (size_t)bi * odom_stride + (size_t)sel.act_idx * 3;
    od3[0] = od[0]; od3[1] = od[1]; od3[2] = od[2];
    if (spre.have) ksim::sim_preload(S, inst, tid, n_measure, spre);
    // the simulator's own acceptance test (ksim::move_accepted): a rejected move appends nothing - no front end then
    accepted = ksim::move_accepted(S, od3[0], od3[1], P0);
    // between relinearisations the SLAM stage is a rank-k covariance update (k_inc.hip): no front end to run ahead
    inc_try = accepted && kslam::inc_precheck(S, inst, P0 + 1, tid, sub_cnt, meta0, meta1, isam);
    // room for the landmarks / factors a step may add (more: slam_finish starts over); the front end only runs ahead when
    // the factor records fit the LDS (else slam_finish takes the workspace variant after the simulator)
    const int Lb = min(S.L_max, L0 + 48), Mb = min(S.M_max, M0 + 48);
    pre = accepted && !inc_try && (3 * (P0 + 1) + 1 + 15) / 16 <= kslam::kDenseTiles && kslam::SlamCtx::big_fits(sim_bytes, lds_bytes, P0 + 1, Lb, Mb);
    if (pre) ctx.setup<true>(S, step_smem, sim_bytes, lds_bytes, inst, P0 + 1, Lb, Mb);
  }
  if (tid == 0) {
    sub_cnt[0] = 0;
    sub_cnt[1] = -1;  // [1..3]: the simulator wave's final counts (stay -1 when it rejects the move)
  }
  __syncthreads();
  if (tid < 64) {
    // wave 0: the (single-wave) simulator; the other seven waves: the part of the SLAM update that does not depend on it
    uint32_t *l0 = reinterpret_cast<uint32_t *>(step_smem);
    uint32_t *l1 = l0 + DRLGX_MT_STRIDE;
    double *dyn = sim_dyn;  // 5008 B: 16-aligned
    if (sel.simlog)  // the simulator ran ahead for the whole action list (k_presim): this action's entry is replayed
      ksim::replay_step_body(S, sel, odom, odom_stride, tid, sub_cnt + 1);
    else
      ksim::sim_step_body(S, sel, odom, odom_stride, n_measure, l0, l1, dyn, tid, nullptr, nullptr, nullptr, true, sub_cnt + 1,
                          n_measure == 2 ? lmbox : nullptr, sel.on(bi) ? P0 : -1, L0, M0, spre);
  } else if (pre) {
    ctx.front<true>(S, tid, P0, L0, M0, P0, L0, isam + 1, false, od3, kslam::SubBarrier{sub_cnt, kslam::kThreads / 64 - 1, 0});
  } else if (inc_try) {
    // the half of the incremental update that does not need this step's measurements: panel -> LDS, the new pose
    // (the LDS plan is a few scalar operations: evaluated here and again after the simulator rather than kept in registers)
    kslam::IncCtx ix;
    bool inc_lds = false;
    if (kslam::inc_plan(S, sel.base + bi, P0 + 1, lds_bytes, sim_bytes, ix, inc_lds, pc, pan_L0, pan_M0)) {
      const kslam::SubBarrier sb{sub_cnt, kslam::kThreads / 64 - 1, 0};
      if (inc_lds) kslam::inc_pre<true, true>(S, ix, tid, od3, sb);
      else kslam::inc_pre<false, true>(S, ix, tid, od3, sb);
    }
  }
  __syncthreads();
  DRLGX_PROF(S, 32);
  if (accepted && tid >= 64 && !sel.simlog) {
    // the two random streams go back to HBM from their LDS images (the simulator wave left the counters in them): seven
    // waves, in the shadow of the few threads that linearise this step's factors
    uint32_t *g = S.mt + (size_t)(sel.base + bi) * 2 * DRLGX_MT_STRIDE;
    const uint32_t *img = reinterpret_cast<const uint32_t *>(step_smem);
    for (int i = tid - 64; i < 2 * DRLGX_MT_STRIDE; i += kslam::kThreads - 64) g[i] = img[i];
  }
  // The pose estimates / information blocks go straight to where the map stage keeps them (its first 10 P_max doubles; its
  // pose tables - 19 P_max doubles - must end inside the simulator's region, free by now, because the map stage fills the
  // last of them while it reads the landmarks).  The landmark estimates stay in the SLAM stage's LDS: the map stage reads
  // them before the first array it writes beyond its pose tables (the information stage, in its phase A) is touched - as
  // long as they lie below its cell masks, which it clears first.
  const bool hand = (size_t)pc * 19 * sizeof(double) + 16 <= sim_bytes - 16;
  const double *lm_lds = nullptr;
  bool inc_done = false;
  if (inc_try) {
    // (sub_cnt[1] < 0 cannot happen: `accepted` is the simulator's own test; the counts would be the old ones and the
    // structure check of inc_post would refuse them)
    kslam::IncCtx ix;
    bool inc_lds = false;
    if (kslam::inc_plan(S, sel.base + bi, P0 + 1, lds_bytes, sim_bytes, ix, inc_lds, pc, pan_L0, pan_M0)) {
      double *hp = hand ? reinterpret_cast<double *>(step_smem) : nullptr;
      inc_done = inc_lds ? kslam::inc_post<true, 0>(S, ix, sub_cnt[2], sub_cnt[3], box, tid, hp, pc)
                         : kslam::inc_post<false, 0>(S, ix, sub_cnt[2], sub_cnt[3], box, tid, hp, pc);
      if (inc_done) lm_lds = ix.thl;  // (the map stage's inputs are in LDS, as after slam_finish)
      else __syncthreads();
    }
  }
  if (!inc_done)
    kslam::slam_finish<FT>(S, sel, lds_bytes, sim_bytes, ctx, pre, sub_cnt + 1, hand ? reinterpret_cast<double *>(step_smem) : nullptr, &lm_lds, box, pc);
  __syncthreads();
  const bool handed = hand && lm_lds != nullptr;  // (lm_lds: set once the SLAM stage ran to its end)
  if (sel.on(bi)) {  // the counts as the simulator wave left them (nothing appended: those before the step)
    const bool appended = sub_cnt[1] >= 0;
    lo.P = appended ? sub_cnt[1] : P0;
    lo.L = appended ? sub_cnt[2] : L0;
    lo.flag = appended ? 0 : 1;
  }
  const unsigned char *map_masks = step_smem + kmap::masks_offset(pc, map_chunk);
  if (!handed || reinterpret_cast<const unsigned char *>(lm_lds + 2 * (size_t)S.L_max) > map_masks) lm_lds = nullptr;
  if (!sel.skip_map) kmap::map_body<false>(S, sel, 1, map_chunk, handed, lm_lds, lo);
  if (S.prof && tid == 0 && bi < 448) S.prof[129 + 2 * bi] = wall_clock64();
}

template <int FT>
__global__ __launch_bounds__(kslam::kThreads) void k_step(DrlgxState S, LaunchSel sel, const double *odom, int odom_stride,
                                                          int n_measure, int lds_bytes, int map_chunk) {
  step_once<FT>(S, sel, odom, odom_stride, n_measure, lds_bytes, map_chunk);
}

// The same kernel with the state struct read through a pointer to its device-resident copy (DrlgxState::self_dev): the
// launch of one workgroup per CU that the headline measures pays for its arguments' first touch in full.
template <int FT>
__global__ __launch_bounds__(kslam::kThreads) void k_step_ref(DrlgxStateConst Sp, LaunchSel sel, const double *odom, int odom_stride,
                                                              int n_measure, int lds_bytes, int map_chunk) {
  if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0) drlgx_warm_state(Sp);
  step_once<FT>(*(const DrlgxState *)Sp, sel, odom, odom_stride, n_measure, lds_bytes, map_chunk);
}

// A whole action LIST per workgroup (the look-ahead's rollouts, EMPlanner2D::simulations_reward's loop, Planner2D.cpp:1432-1460):
// instance bi runs its actions [sel.act_idx, min(a_end, n_act[bi])) back to back - the fused step above once per action, the
// state of one action handed to the next through HBM / L2 exactly as between launches (a workgroup barrier orders them).
// Against one launch per action index this removes the per-launch tails (every launch lasted as long as its slowest
// instance) and lets the hardware deal the candidates to the CUs as they finish: a candidate with few actions makes room
// for the next one instead of idling through the other candidates' later actions.
template <int FT>
__global__ __launch_bounds__(kslam::kThreads) void k_step_loop(DRLGX_KS_PARAM, LaunchSel sel, const double *odom, int odom_stride,
                                                               int n_measure, int lds_bytes, int map_chunk, int a_end) {
  const DrlgxState &S = DRLGX_KS_REF;
  const int bi = blockIdx.x;
  const int n_mine = sel.n_act ? min(a_end, sel.n_act[bi]) : a_end;
  if (sel.active && !sel.active[bi]) return;
  for (int a = sel.act_idx; a < n_mine; ++a) {
    LaunchSel one = sel;
    one.act_idx = a;
    // (the pose bound of THIS action, as a launch per action index would pass it: LDS plans - and with them the choice between
    // equivalent code paths whose roundings differ - then agree with that form bit for bit; sel.pcap = the first action's bound)
    one.pcap = min(sel.pcap + (a - sel.act_idx), S.P_max);
    step_once<FT>(S, one, odom, odom_stride, n_measure, lds_bytes, map_chunk);
    __syncthreads();  // (also a workgroup-scope fence: the next action reads what this one wrote)
  }
}

// The same fusion around the pose-chain solver (trajectories beyond the LDS-resident dense solve): simulate -> k_slam_arrow's
// body -> virtual map in one kernel.  Only the variant whose landmark system is swept in LDS (<= 63 landmarks).
__device__ __forceinline__ void step_arrow_once(const DrlgxState &S, const LaunchSel &sel, const double *odom, int odom_stride, int n_measure,
                                                int lds_bytes, int map_chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
  const int tid = drlgx_tid();
  if (tid < 64) {
    uint32_t *l0 = reinterpret_cast<uint32_t *>(step_smem);
    uint32_t *l1 = l0 + DRLGX_MT_STRIDE;
    double *dyn = reinterpret_cast<double *>(step_smem + 2 * DRLGX_MT_STRIDE * sizeof(uint32_t));
    if (sel.simlog) ksim::replay_step_body(S, sel, odom, odom_stride, tid);
    else ksim::sim_step_body(S, sel, odom, odom_stride, n_measure, l0, l1, dyn, tid);
  }
  __syncthreads();
  kslam::arrow_body<0>(S, sel, lds_bytes);  // (or, between relinearisations, the incremental update: arrow_body's first lines)
  __syncthreads();
  if (!sel.skip_map) kmap::map_body<false>(S, sel, 1, map_chunk);
}
__global__ __launch_bounds__(kslam::kThreads) void k_step_arrow(DRLGX_KS_PARAM, LaunchSel sel, const double *odom, int odom_stride,
                                                                int n_measure, int lds_bytes, int map_chunk) {
  const DrlgxState &S = DRLGX_KS_REF;
  step_arrow_once(S, sel, odom, odom_stride, n_measure, lds_bytes, map_chunk);
}
// ... and a whole action list per workgroup, as k_step_loop (the look-ahead of trajectories beyond the dense solver's reach)
__global__ __launch_bounds__(kslam::kThreads) void k_step_arrow_loop(DRLGX_KS_PARAM, LaunchSel sel, const double *odom, int odom_stride,
                                                                     int n_measure, int lds_bytes, int map_chunk, int a_end) {
  const DrlgxState &S = DRLGX_KS_REF;
  const int bi = blockIdx.x;
  const int n_mine = sel.n_act ? min(a_end, sel.n_act[bi]) : a_end;
  if (sel.active && !sel.active[bi]) return;
  for (int a = sel.act_idx; a < n_mine; ++a) {
    LaunchSel one = sel;
    one.act_idx = a;
    one.pcap = min(sel.pcap + (a - sel.act_idx), S.P_max);
    step_arrow_once(S, one, odom, odom_stride, n_measure, lds_bytes, map_chunk);
    __syncthreads();
  }
}

}  // namespace kstep

bool drlgx_step_fusable(const DrlgxState &S, int p_bound) {
  int chunk = 0;
  const int Pb = p_bound < S.P_max ? p_bound : S.P_max;
  const size_t nf = std::max<size_t>(kslam::slam_dim(Pb), 16 * kslam::kFastTiles);
  // (the SLAM stage sits behind the simulator's LDS: its front end runs beside the simulator wave)
  return drlgx_slam_in_lds(Pb, S.L_max, S.M_max) &&
         kstep::sim_lds_bytes(S.LG, Pb) + kslam::slam_small_bytes(Pb, S.L_max, S.M_max) + kslam::sweep_region_doubles(nf) * 8 <= (size_t)kslam::kLdsBudget &&
         drlgx_map_lds_bytes(S, &chunk, Pb) <= (size_t)kslam::kLdsBudget;
}

// the fused step around the pose-chain solver: its LDS-swept landmark system (<= 63 landmarks) and the same map / simulator
// conditions as above
bool drlgx_step_arrow_fusable(const DrlgxState &S) {
  int chunk = 0;
  return 2 * S.L_max + 1 <= 16 * kslam::kFastTilesArrow && drlgx_map_lds_bytes(S, &chunk) <= (size_t)kslam::kLdsBudget &&
         (size_t)(2 * DRLGX_MT_STRIDE * 4 + (2 * S.LG + 2) * 8 + S.LG * 4) <= (size_t)kslam::kLdsBudget;
}

void drlgx_launch_step_arrow(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure) {
  int chunk = 0;
  (void)drlgx_map_lds_bytes(S, &chunk, sel.pcap);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kstep::k_step_arrow)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, kslam::kLdsBudget);
  hipLaunchKernelGGL(kstep::k_step_arrow, dim3(sel.n), dim3(kslam::kThreads), kslam::kLdsBudget, st, DRLGX_KS_ARG(S), sel, odom, odom_stride, n_measure,
                     kslam::kLdsBudget, chunk);
}

void drlgx_launch_step_arrow_loop(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure, int a_end) {
  int chunk = 0;
  (void)drlgx_map_lds_bytes(S, &chunk, std::min(sel.pcap + (a_end - 1 - sel.act_idx), S.P_max));  // (sized for the range's last action)
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kstep::k_step_arrow_loop)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, kslam::kLdsBudget);
  hipLaunchKernelGGL(kstep::k_step_arrow_loop, dim3(sel.n), dim3(kslam::kThreads), kslam::kLdsBudget, st, DRLGX_KS_ARG(S), sel, odom, odom_stride, n_measure,
                     kslam::kLdsBudget, chunk, a_end);
}

void drlgx_launch_step_loop(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure, int a_end) {
  int chunk = 0;
  (void)drlgx_map_lds_bytes(S, &chunk, std::min(sel.pcap + (a_end - 1 - sel.act_idx), S.P_max));  // (sized for the range's last action)
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kstep::k_step_loop<kslam::kFastTiles>)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, kslam::kLdsBudget);
  hipLaunchKernelGGL((kstep::k_step_loop<kslam::kFastTiles>), dim3(sel.n), dim3(kslam::kThreads), kslam::kLdsBudget, st, DRLGX_KS_ARG(S), sel, odom,
                     odom_stride, n_measure, kslam::kLdsBudget, chunk, a_end);
}

void drlgx_launch_step(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure) {
  int chunk = 0;
  (void)drlgx_map_lds_bytes(S, &chunk, sel.pcap);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kstep::k_step<kslam::kFastTiles>),
                       reinterpret_cast<const void *>(&kstep::k_step_ref<kslam::kFastTiles>)};
  drlgx_ensure_lds_attr(attr_set, fns, 2, kslam::kLdsBudget);
  // (DRLGX_STATE_PTR=0: the struct by value, the A/B of profiles/r05_ab_state_pointer.txt / r06_ab_state_const.txt)
  static const bool by_value = [] { const char *sp = getenv("DRLGX_STATE_PTR"); return sp && sp[0] == '0'; }();
  if (!by_value)
    hipLaunchKernelGGL((kstep::k_step_ref<kslam::kFastTiles>), dim3(sel.n), dim3(kslam::kThreads), kslam::kLdsBudget, st, (DrlgxStateConst)S.self_dev, sel,
                       odom, odom_stride, n_measure, kslam::kLdsBudget, chunk);
  else
    hipLaunchKernelGGL((kstep::k_step<kslam::kFastTiles>), dim3(sel.n), dim3(kslam::kThreads), kslam::kLdsBudget, st, S, sel, odom,
                       odom_stride, n_measure, kslam::kLdsBudget, chunk);
}
