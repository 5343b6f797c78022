// Fused belief step: simulate -> SLAM -> virtual map for one instance in ONE kernel (one 512-thread workgroup per
// instance).  The three stages are independent between instances, so fusing them removes two launches per step and,
// more importantly, the two grid-wide joins: an instance no longer waits for the slowest workgroup of the previous stage
// (launch + ramp + tail cost ~3-5 us per kernel at 256 workgroups).  Wave 0 runs the (single-wave) simulator while the
// other waves wait at the barrier they would otherwise have spent between kernels.
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

template <int FT>
__global__ __launch_bounds__(kslam::kThreads) void k_step(DrlgxState S, LaunchSel sel, const double *odom, int odom_stride,
                                                          int n_measure, int lds_bytes, int map_chunk) {
  static_assert(kslam::kThreads == kmap::kThreads, "the fused kernel runs both stages with one workgroup size");
  extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
  const int tid = threadIdx.x;
  if (tid < 64) {
    uint32_t *l0 = reinterpret_cast<uint32_t *>(step_smem);
    uint32_t *l1 = l0 + DRLGX_MT_STRIDE;
    double *dyn = reinterpret_cast<double *>(step_smem + 2 * DRLGX_MT_STRIDE * sizeof(uint32_t));  // 5008 B: 16-aligned
    ksim::sim_step_body(S, sel, odom, odom_stride, n_measure, l0, l1, dyn, tid);
  }
  __syncthreads();
  kslam::slam_body<FT>(S, sel, lds_bytes);
  __syncthreads();
  kmap::map_body(S, sel, 1, map_chunk);
}

// The same fusion around the pose-chain solver (trajectories beyond the LDS-resident dense solve): simulate -> k_slam_arrow's
// body -> virtual map in one kernel.  Only the variant whose landmark system is swept in LDS (<= 63 landmarks).
__global__ __launch_bounds__(kslam::kThreads) void k_step_arrow(DrlgxState S, LaunchSel sel, const double *odom, int odom_stride,
                                                                int n_measure, int lds_bytes, int map_chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char step_smem[];
  const int tid = threadIdx.x;
  if (tid < 64) {
    uint32_t *l0 = reinterpret_cast<uint32_t *>(step_smem);
    uint32_t *l1 = l0 + DRLGX_MT_STRIDE;
    double *dyn = reinterpret_cast<double *>(step_smem + 2 * DRLGX_MT_STRIDE * sizeof(uint32_t));
    ksim::sim_step_body(S, sel, odom, odom_stride, n_measure, l0, l1, dyn, tid);
  }
  __syncthreads();
  kslam::arrow_body<0>(S, sel, lds_bytes);
  __syncthreads();
  kmap::map_body(S, sel, 1, map_chunk);
}

}  // namespace kstep

bool drlgx_step_fusable(const DrlgxState &S, int p_bound) {
  int chunk = 0;
  return drlgx_slam_in_lds(p_bound < S.P_max ? p_bound : S.P_max, S.L_max, S.M_max) && drlgx_map_lds_bytes(S, &chunk) <= (size_t)kslam::kLdsBudget &&
         (size_t)(2 * DRLGX_MT_STRIDE * 4 + (2 * S.LG + 2) * 8 + S.LG * 4) <= (size_t)kslam::kLdsBudget;
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
  (void)drlgx_map_lds_bytes(S, &chunk);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kstep::k_step_arrow)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, kslam::kLdsBudget);
  hipLaunchKernelGGL(kstep::k_step_arrow, dim3(sel.n), dim3(kslam::kThreads), kslam::kLdsBudget, st, S, sel, odom, odom_stride, n_measure,
                     kslam::kLdsBudget, chunk);
}

void drlgx_launch_step(const DrlgxState &S, hipStream_t st, LaunchSel sel, const double *odom, int odom_stride, int n_measure) {
  int chunk = 0;
  (void)drlgx_map_lds_bytes(S, &chunk);
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&kstep::k_step<kslam::kFastTiles>)};
  drlgx_ensure_lds_attr(attr_set, fns, 1, kslam::kLdsBudget);
  hipLaunchKernelGGL((kstep::k_step<kslam::kFastTiles>), dim3(sel.n), dim3(kslam::kThreads), kslam::kLdsBudget, st, S, sel, odom,
                     odom_stride, n_measure, kslam::kLdsBudget, chunk);
}
