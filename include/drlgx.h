/* drlgx — C ABI of the MI355X-native exploration belief-step / GCN hot path.
 *
 * This is the drop-in boundary: plain C types, opaque handle, int return codes, no
 * torch types.  The reference has no C ABI today — its boundary is two pybind11 modules
 * (`ss2d`: /root/reference/src/SS2D.cpp:18-258, `planner2d`: /root/reference/src/Planner2D.cpp:9-106);
 * each entry point below names the reference interface it replaces.  The Python shims in
 * drl_graph_exploration_amd/ (ss2d.py, planner2d.py, vecenv.py, networks.py) sit on top of it.
 *
 * Conventions
 *  - "dev" pointers are DEVICE (HBM) pointers owned by the caller (e.g. torch tensor data_ptr());
 *    "host" pointers are ordinary host memory.  Every function that takes host pointers says so
 *    in its name (…_host) or parameter comment; those functions synchronise the engine stream.
 *  - All kernels of an engine are enqueued on ONE HIP stream (drlgx_set_stream; default: a private
 *    stream).  Non-_host calls are asynchronous with respect to the host.
 *  - Instance = one belief state.  Instances [0, n_envs) are live environments; the engine owns
 *    additional scratch instances for look-ahead rollouts (one "base" per env + n_rollouts).
 *  - Return value: 0 = OK, <0 = DRLGX_E_* error (see drlgx_strerror).  Nothing aborts.
 *  - There is NO CPU fallback: if no HIP device is present drlgx_create fails with DRLGX_E_NODEVICE.
 */
#ifndef DRLGX_H
#define DRLGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRLGX_OK 0
#define DRLGX_E_INVALID (-1)   /* bad argument */
#define DRLGX_E_NODEVICE (-2)  /* no HIP device / runtime error */
#define DRLGX_E_CAPACITY (-3)  /* an instance overflowed max_poses / max_landmarks / max_factors */
#define DRLGX_E_HIP (-4)       /* HIP runtime error (message in drlgx_last_error) */
#define DRLGX_E_NUMERIC (-5)   /* non-SPD system met by the SLAM solve (gtsam: IndeterminantLinearSystemException) */

#define DRLGX_ALG_EM_AOPT 0 /* Planner2D.h OptimizationAlgorithm::EM_AOPT (trace) */
#define DRLGX_ALG_EM_DOPT 1 /* OptimizationAlgorithm::EM_DOPT (determinant)      */

/* Parameters of one engine.  Mirrors the ini sections read by scripts/envs/pyss2d.py:10-55 and
 * scripts/envs/pyplanner2d.py:24-54 (values: scripts/envs/exploration_env.ini). Angles in radians,
 * already wrapped through Rot2(x).theta() as the reference setters do (Simulation2D.h:52-55,152). */
typedef struct drlgx_config {
  /* [Sensor Model]  BearingRangeSensorModel::Parameter (include/em_exploration/Simulation2D.h:47-76) */
  double bearing_noise, range_noise, min_bearing, max_bearing, min_range, max_range;
  /* [Control Model]  SimpleControlModel::Parameter (Simulation2D.h:145-160) */
  double translation_noise, rotation_noise;
  /* [Environment]  unpadded box, Environment::Parameter (Simulation2D.h:243-268) */
  double env_min_x, env_max_x, env_min_y, env_max_y, safe_distance;
  /* map box = environment padded by ext = 20 m (pyss2d.py:48-55) */
  double map_min_x, map_max_x, map_min_y, map_max_y;
  /* [Virtual Map]  VirtualMap::Parameter (include/em_exploration/VirtualMap.h:17-38) */
  double resolution, sigma0;
  int32_t num_samples; /* only 1 is supported (the shipped value); >1 re-averages identical maps */
  /* [Simulator] */
  double sigma_x0, sigma_y0, sigma_theta0;
  int32_t num_landmarks; /* Simulator.num: ground-truth landmarks sampled per env */
  /* [Planner]  EMPlanner2D::Parameter (include/em_exploration/Planner2D.h; src/Planner2D.cpp:26-41) */
  double angle_weight, distance_weight0, distance_weight1, occupancy_threshold, max_edge_length;
  int32_t algorithm; /* DRLGX_ALG_* */
  /* capacities (new: the reference grows std::vectors) */
  int32_t max_poses;     /* P_max per instance (>= 2; any value whose tables fit the 160 KB LDS: drlgx_create checks) */
  int32_t max_landmarks; /* L_max observed landmarks per instance (likewise; no fixed cap) */
  int32_t max_factors;   /* M_max bearing-range factors per instance */
  int32_t max_actions;   /* A_max actions per look-ahead candidate */
  int32_t max_snapshots; /* device-resident snapshot slots of the live environments (0..16) */
} drlgx_config;

typedef struct drlgx_engine drlgx_engine;

/* ---- life cycle ----------------------------------------------------------------------------- */

/* Create an engine with n_envs environments and n_rollouts look-ahead scratch instances on HIP
 * device `device`.  Replaces the constructors ss2d.Simulator2D / SLAM2D / VirtualMap and
 * planner2d.EMPlanner2D (src/SS2D.cpp:173-188,190-211,226-246; src/Planner2D.cpp:75-92). */
int drlgx_create(const drlgx_config *cfg, int n_envs, int n_rollouts, int device, drlgx_engine **out);
int drlgx_destroy(drlgx_engine *e);
/* Use the caller's HIP stream (hipStream_t) for all subsequent work: NULL = HIP's default (null)
 * stream, (void*)-1 = back to the engine's private non-blocking stream (the initial state). */
int drlgx_set_stream(drlgx_engine *e, void *hip_stream);
int drlgx_synchronize(drlgx_engine *e);
const char *drlgx_strerror(int code);
const char *drlgx_last_error(const drlgx_engine *e);
/* Device-side status word of the last kernels: 0 or a DRLGX_E_* code (synchronises). */
int drlgx_status_host(drlgx_engine *e);
/* The same, with `bytes` of device memory copied to the host under the same synchronisation (bytes = 0: drlgx_status_host).  A batched
 * trainer needs a handful of small device results per vector step on the host - counts, offsets, the read-out of the policy
 * (scripts/policy.py:100-127, 392-400 read them one Python attribute at a time); each separate read drains the stream again. */
int drlgx_status_fetch_host(drlgx_engine *e, const void *src_dev, size_t bytes, void *dst_host);

/* ---- SS2D life cycle (scripts/envs/pyss2d.py:58-206) ----------------------------------------- */

/* SS2D.__init__ for `n` environments: seed the three mt19937 streams with seeds[i]
 * (Simulator2D.cpp:436-443), place the vehicle at start_xytheta[i], rejection-sample the landmarks
 * (Simulator2D::addLandmarks, Simulator2D.cpp:445-464), add the prior (SLAM2D::addPrior,
 * SLAM2D.cpp:44-57), first measure + optimize.  env_ids / seeds / start_xytheta are HOST arrays
 * (n, n, n*3).  Resets the virtual map to its untouched state (VirtualMap::initialize). */
int drlgx_reset_host(drlgx_engine *e, int n, const int32_t *env_ids, const uint32_t *seeds,
                     const double *start_xytheta);

/* SS2D.simulate(odom, core=True) for every environment with active[i] != 0 (NULL = all):
 * Simulator2D::move + SLAM2D::addOdometry, two noisy Simulator2D::measure calls (the first only
 * consumes noise: pyss2d.py:182), SLAM2D::addMeasurement, SLAM2D::optimize,
 * VirtualMap::updateProbability + updateInformation.  odom_dev: DEVICE double[n_envs*3]
 * (x, y, theta); active_dev: DEVICE uint8[n_envs] or NULL. */
int drlgx_step(drlgx_engine *e, const double *odom_dev, const uint8_t *active_dev);

/* ExplorationEnv.step (scripts/envs/exploration_env.py:98-105: `for a in actions: self._sim.simulate(a)`) one action
 * index at a time for all envs: env i executes actions[i][action_index] (DEVICE double [n_envs][max_actions][3], the plans
 * drlgx_line_plan emits) while action_index < n_actions[i].  map_last_only != 0: the virtual map - a pure function of the
 * SLAM state - is rebuilt at each env's last action only (and the marginal covariances only where the map needs them);
 * the state after the plan is the same, intermediate getters of the map see the previous rebuild. */
int drlgx_step_plan(drlgx_engine *e, const double *actions_dev, const int32_t *n_actions_dev, int action_index,
                    int map_last_only);
/* The same loop for ALL action indices in one call: env i executes actions[i][0 .. n_actions[i]); max_n_actions: a host-side
 * bound of the plan lengths (<= max_actions).  A workgroup runs its env's whole plan inside one launch wherever a fused step
 * kernel serves the pose counts the plans reach (one launch for the actions the dense-solver step serves, one for the rest
 * around the pose-chain solver: the same per-action choice as drlgx_step_plan makes), else one drlgx_step_plan per action
 * index; bit-equal either way. */
int drlgx_step_plans(drlgx_engine *e, const double *actions_dev, const int32_t *n_actions_dev, int max_n_actions,
                     int map_last_only);

/* ---- staged form of the belief step -----------------------------------------------------------
 * The reference's pybind classes are driven call by call (scripts/envs/pyss2d.py:102-138 SS2D.__init__, :171-206
 * SS2D.simulate); the object-level shims (drl_graph_exploration_amd/ss2d.py) map every such call onto one of these.
 * drlgx_step == stage_move; stage_measure (discarded); stage_measure + stage_add_measurements; stage_optimize;
 * stage_update_map, fused.  All buffers are DEVICE memory; active_dev: uint8[n_envs] or NULL (= all). */
/* SS2D.__init__ up to SLAM2D::addPrior (src/SS2D.cpp:173-176 Simulator2D(...), initialize_vehicle, random_landmarks;
 * :191 SLAM2D::add_prior): like drlgx_reset_host but WITHOUT the first measure / optimise / map reductions. */
int drlgx_stage_reset_host(drlgx_engine *e, int n, const int32_t *env_ids, const uint32_t *seeds, const double *start_xytheta);
/* SLAM2D::add_prior(VehicleBeliefState(pose, information)) with a FULL information matrix (src/SS2D.cpp:191,
 * SLAM2D.cpp:44-57: noiseModel::Gaussian::Information): replaces the prior information of one env - the diagonal of the
 * ini file's sigma_x0 / sigma_y0 / sigma_theta0 that the resets install - after drlgx_stage_reset_host and before the first
 * optimise.  HOST, row major 3 x 3, symmetric (DRLGX_E_INVALID otherwise). */
int drlgx_stage_set_prior_information_host(drlgx_engine *e, int env, const double *information9);
/* SLAM2D::add_prior(VehicleBeliefState(pose, information)) with a POSE that is not the simulator's initial vehicle pose
 * (src/SS2D.cpp:193, SLAM2D.cpp:44-57: the pose of the prior factor AND the initial estimate of x0; the reference's own caller,
 * pyss2d.py:124-135, passes the vehicle's pose - what the resets install): replaces it for one env, between
 * drlgx_stage_reset_host and the first measurement (one pose, no factor: DRLGX_E_INVALID otherwise).  HOST, (x, y, theta). */
int drlgx_stage_set_prior_pose_host(drlgx_engine *e, int env, const double *xytheta);
/* Simulator2D::move(odom, true) (src/SS2D.cpp:181) + SLAM2D::add_odometry (:193). odom_dev: double[n_envs*3]. */
int drlgx_stage_move(drlgx_engine *e, const double *odom_dev, const uint8_t *active_dev);
/* Simulator2D::measure() (src/SS2D.cpp:182): the noisy (bearing, range) of every ground-truth landmark that passes the
 * sensor gates, in the reference's iteration order; advances the sensor RNG; adds nothing to the graph.
 * keys_dev: int32[n_envs*num_landmarks], bearing_range_dev: double[n_envs*num_landmarks*2], count_dev: int32[n_envs]. */
int drlgx_stage_measure(drlgx_engine *e, const uint8_t *active_dev, int32_t *keys_dev, double *bearing_range_dev, int32_t *count_dev);
/* SLAM2D::add_measurement(key, m) (src/SS2D.cpp:194) for count[i] listed measurements of env i at its newest pose. */
int drlgx_stage_add_measurements(drlgx_engine *e, const uint8_t *active_dev, const int32_t *keys_dev, const double *bearing_range_dev,
                                 const int32_t *count_dev);
/* SLAM2D::optimize(update_covariance = true) (src/SS2D.cpp:198). */
int drlgx_stage_optimize(drlgx_engine *e, const uint8_t *active_dev);
/* rebuild != 0: VirtualMap::update_probability(slam, sensor) + update_information(map, sensor) (src/SS2D.cpp:228-235);
 * rebuild == 0: only the utility / explored sums of the stored planes (the state a freshly constructed VirtualMap is in). */
int drlgx_stage_update_map(drlgx_engine *e, const uint8_t *active_dev, int rebuild);

/* FastMarginals2::update / propagate (src/em_exploration/FastMarginals.cpp:188-321) fed as the EM planner feeds it
 * (src/em_exploration/Planner2D.cpp:652-737 updateNodeInformation_EM, :472-551 updateTrajectory_EM): for n_cand
 * candidates (environment, action list) the 3x3 marginal covariance of EVERY pose after appending one predicted
 * (noise-free) pose per action and the noise-free bearing-range factors from the new poses to the landmarks of the
 * estimated map that pass the sensor gates - an EKF-style propagate / update, Sigma' = Sigma - Sigma A^T (I + A Sigma
 * A^T)^-1 A Sigma, without re-solving and without touching the environments' state.
 * cand_env_dev int32[n_cand]; actions_dev double[n_cand*max_actions*3]; n_actions_dev int32[n_cand];
 * cov_out_dev double[n_cand*out_stride_poses*9] (row-major 3x3 per pose: the P old poses, then one per action),
 * out_stride_poses >= max_poses + max_actions; n_out_dev int32[n_cand] = P + n_actions.  At most 256 predicted
 * measurements per candidate (DRLGX_E_CAPACITY beyond). */
int drlgx_fm2_update(drlgx_engine *e, int n_cand, const int32_t *cand_env_dev, const double *actions_dev, const int32_t *n_actions_dev,
                     double *cov_out_dev, int out_stride_poses, int32_t *n_out_dev);

/* Simulator2D::addLandmarks(landmarks, num_random, params) with a non-empty list (Simulator2D.cpp:445-464; the ini file's
 * optional [Landmarks] section, scripts/envs/pyss2d.py:107-115): the n_fixed listed points take the ground-truth keys
 * 0 .. n_fixed - 1 of EVERY env, the remaining cfg.num_landmarks - n_fixed are sampled as before (cfg.num_landmarks is the
 * total).  HOST array xy [n_fixed][2]; takes effect at the next (staged) reset; n_fixed = 0 restores pure sampling. */
int drlgx_set_fixed_landmarks_host(drlgx_engine *e, int n_fixed, const double *xy);
/* EMPlanner2D(parameter, ...) / EMPlanner2D::setParameter (src/Planner2D.cpp:73-77): the planner constants the kernels
 * read (line-plan edge length, utility weights, occupancy threshold, distance angle weight, algorithm) can be replaced
 * after drlgx_create - the reference constructs its planner after the simulator / SLAM objects. */
int drlgx_set_planner_parameter(drlgx_engine *e, double angle_weight, double distance_weight0, double distance_weight1,
                                double occupancy_threshold, double max_edge_length, int algorithm);

/* EMPlanner2D::calculateUtility (static; src/em_exploration/Planner2D.cpp:354-366) for every
 * environment: U = sum_v tr(info_v^-1) + dist * (w0 - (w0 - w1) * known / V).
 * dist_dev: DEVICE double[n_envs] or NULL (= 0); out_dev: DEVICE double[n_envs]. */
int drlgx_utility(drlgx_engine *e, const double *dist_dev, double *out_dev);
/* EMPlanner2D::calculateUncertainty_EM (Planner2D.cpp:321-341): weighted trace (EM_AOPT) or
 * sum 1[p>0.49]/det(info) (EM_DOPT).  out_dev: DEVICE double[n_envs]. */
int drlgx_uncertainty_em(drlgx_engine *e, int algorithm, double *out_dev);
/* VirtualMap::explored (src/em_exploration/VirtualMap.cpp:47-59). out_dev: DEVICE double[n_envs]. */
int drlgx_explored(drlgx_engine *e, double *out_dev);

/* The metric trio of the reference's evaluation script (scripts/test.py:136-142) for every environment, on the device:
 * out_dev: DEVICE double[n_envs*3] = landmark error (scripts/envs/exploration_env.py:170-177, with its sigma0 argument),
 * map entropy (scripts/test.py:61-74; its per-map-size constant is 0.5 ln 0.5 x the number of padding cells),
 * max localisation uncertainty = max_i tr(marginalCovariance(x_i)) (exploration_env.py:190-194). */
int drlgx_metrics(drlgx_engine *e, double sigma0, double *out_dev);
/* VirtualMap::toCovArray (src/SS2D.cpp:239, src/em_exploration/VirtualMap.cpp:140-151) for every environment:
 * length_dev / angle_dev: DEVICE double[n_envs*rows*cols], row-major grids (min(sqrt(larger eigenvalue of the cell
 * covariance), sigma0) and the direction of its eigenvector). */
int drlgx_cov_array(drlgx_engine *e, double *length_dev, double *angle_dev);

/* ---- planner calls used by the DRL loop ------------------------------------------------------ */

/* EMPlanner2D::line_planner (Planner2D.cpp:937-1041), frontier-goal branch, for n_cand goals.
 * cand_env_dev: DEVICE int32[n_cand]; goal_dev: DEVICE double[n_cand*2];
 * actions_dev: DEVICE double[n_cand*max_actions*3] (x, y, theta); n_actions_dev: DEVICE int32[n_cand]. */
int drlgx_line_plan(drlgx_engine *e, int n_cand, const int32_t *cand_env_dev, const double *goal_dev,
                    double *actions_dev, int32_t *n_actions_dev);

/* EMPlanner2D::simulations_reward (Planner2D.cpp:1416-1468) for n_cand candidates
 * (any n_cand >= 0; more than n_rollouts candidates run in successive waves of n_rollouts; n_rollouts must be
 * >= 1): deep-copies the belief and simulator state of env cand_env[i] (including
 * the RNG states), re-solves at the best estimate (SLAM2D::set_copy_isam, SLAM2D.cpp:490-497), rolls
 * the action list forward with one noisy move/measure + SLAM2D::copy_optimize + virtual-map rebuild
 * per action and returns reward = U_before(0) - U_after(dist).  Live environments are not modified.
 * rewards_dev: DEVICE double[n_cand]. */
int drlgx_lookahead(drlgx_engine *e, int n_cand, const int32_t *cand_env_dev, const double *actions_dev,
                    const int32_t *n_actions_dev, double *rewards_dev);

/* drlgx_lookahead with a host-side bound on the plan lengths: max_n_actions >= every n_actions_dev[i] (the caller
 * usually knows it from stepping the chosen plan; 0 < max_n_actions <= max_actions).  Action indices beyond it are
 * not launched at all (drlgx_lookahead launches all max_actions of them and lets the workgroups exit).  Where the fused
 * step kernels serve the pose counts the rollouts reach, a candidate's WHOLE action list runs inside one workgroup (two
 * launches per wave of candidates instead of one per action index) and the rollouts' simulator - whose draws do not depend
 * on the SLAM state - is run ahead for the whole list and replayed; same rewards bit for bit (DRLGX_LOOKAHEAD_LOOP=0 /
 * DRLGX_LOOKAHEAD_PRESIM=0 switch the two off for A/B runs). */
int drlgx_lookahead_bounded(drlgx_engine *e, int n_cand, const int32_t *cand_env_dev, const double *actions_dev,
                            const int32_t *n_actions_dev, int max_n_actions, double *rewards_dev);

/* ---- graph export for the policy (a14-a17) --------------------------------------------------- */

/* ExplorationEnv.frontier + graph_matrix (scripts/envs/exploration_env.py:196-348) on top of
 * SLAM2D::adjacency_degree_get (SLAM2D.cpp:198-273) and DeepQ.data_process (scripts/policy.py:211-232),
 * batched: one graph per environment, concatenated PyG-Batch style.
 *   node order per graph: landmarks by ground-truth key, poses by index, frontier nodes.
 * Outputs (DEVICE, caller-sized with the capacities from drlgx_graph_capacity):
 *   node_off[n_envs+1], edge_off[n_envs+1]      int32 prefix offsets
 *   x[total_nodes*5] float32                    features (trace, dist, bearing diff, prob, type)
 *   edge_index[2*total_edges] int64             (row array then col array, global node ids)
 *   edge_attr[total_edges] float32
 *   n_frontier[n_envs] int32, frontier_xy[n_envs*max_frontier*2] double
 *   nearest_frontier_node[n_envs] int32         local node id of the robot's nearest frontier
 * Edge order inside a graph is DeepQ.data_process's (row-major first-seen, both directions). */
int drlgx_graph_capacity(const drlgx_engine *e, int *max_nodes, int *max_edges, int *max_frontier);
int drlgx_graph(drlgx_engine *e, int32_t *node_off_dev, int32_t *edge_off_dev, float *x_dev,
                int64_t *edge_index_dev, float *edge_attr_dev, int32_t *n_frontier_dev, double *frontier_xy_dev,
                int32_t *nearest_frontier_node_dev);

/* ---- state export (getters of the pybind surface; HOST outputs, synchronise) ------------------ */

/* counts per instance: out[0]=poses, out[1]=landmarks, out[2]=factors, out[3]=step, out[4]=isam update count */
int drlgx_get_counts_host(drlgx_engine *e, int inst, int32_t out[5]);
/* the same five counts for every live env, asynchronously on the engine stream: counts_dev[n_envs*5] (device).
 * Replaces the per-env Python reads `slam.key_size()`, `slam.map.get_landmark_size()`, `sim.step` in a batched loop. */
int drlgx_counts(drlgx_engine *e, int32_t *counts_dev);
/* Environment.iter_trajectory / get_current_vehicle of SLAM2D.map (src/SS2D.cpp:141-171):
 * xytheta[P*3], information[P*9] (may be NULL). */
int drlgx_get_poses_host(drlgx_engine *e, int inst, double *xytheta, double *information);
/* Environment.iter_landmarks of SLAM2D.map, sorted by ground-truth key: keys[L], xy[L*2], information[L*4]. */
int drlgx_get_landmarks_host(drlgx_engine *e, int inst, int32_t *keys, double *xy, double *information);
/* marginal-covariance traces (SLAM2D::features_out, SLAM2D.h:74-76): landmarks by key, then poses. */
int drlgx_get_cov_traces_host(drlgx_engine *e, int inst, double *lm_trace, double *pose_trace);
/* VirtualMap.to_array / to_cov_trace / iter_virtual_landmarks (src/SS2D.cpp:226-246):
 * prob[V], info[V*4] (2x2 row-major), cov_trace[V], updated[V]; any may be NULL. rows/cols via drlgx_vm_shape. */
int drlgx_vm_shape(const drlgx_engine *e, int *rows, int *cols);
int drlgx_get_virtual_map_host(drlgx_engine *e, int inst, double *prob, double *info, double *cov_trace,
                               uint8_t *updated);
/* Simulator2D.vehicle / Simulator2D.environment (ground truth): vehicle_xytheta[3], landmarks_xy[num*2] by key. */
int drlgx_get_ground_truth_host(drlgx_engine *e, int inst, double *vehicle_xytheta, double *landmarks_xy);
/* SLAM2D::adjacency_out / features_out (dense, (L+P)^2 and L+P doubles) for one environment. */
int drlgx_get_adjacency_host(drlgx_engine *e, int inst, double *adjacency, double *features);
/* factor list: pose index, landmark key, bearing, range (M each). */
int drlgx_get_factors_host(drlgx_engine *e, int inst, int32_t *pose, int32_t *key, double *bearing, double *range);
/* libstdc++ iteration order of the ground-truth landmark hash map (Simulator2D.cpp:331-344): order[num]. */
int drlgx_get_landmark_order_host(const drlgx_engine *e, int32_t *order);

/* Snapshot / restore of live environments (device-to-device; used by benchmarks and by
 * ExplorationEnv-style "copy" semantics).  slot in [0, n_snapshots) — allocated lazily. */
int drlgx_snapshot(drlgx_engine *e, int slot);
int drlgx_restore(drlgx_engine *e, int slot);

/* Per-kernel timing (HIP events on the engine stream): enable (on = 1: spans around whatever is launched; on = 2: the
 * belief step is launched as its three stage kernels instead of the fused kernel, so that each stage gets its own span),
 * then read accumulated milliseconds and launch counts.  timer ids: 0 sim, 1 slam, 2 map, 3 copy/prepare, 4 graph,
 * 5 fused belief step (sim + slam + map), 7 = an EMPTY span recorded once per drlgx_step (the event-pair overhead a caller
 * subtracts from the per-launch averages).  */
#define DRLGX_N_TIMERS 8
int drlgx_timing_enable(drlgx_engine *e, int on);
int drlgx_timing_read_host(drlgx_engine *e, double ms[DRLGX_N_TIMERS], int64_t launches[DRLGX_N_TIMERS]);

/* Development aid: in-kernel phase stamps (wall_clock64, 100 MHz) of ONE workgroup of the belief kernels.  arm = 0 disarms;
 * arm & 1 arms; arm >> 8 = the workgroup (instance index inside the launch) that stamps, 0 by default.  Slots: 0-7 the SLAM
 * back end, 8-13 the simulator, 14 / 24-35 the SLAM front end, 16-21 / 40-43 / 48-63 the map stage.  out (may be NULL)
 * receives the last stamps: 64 values - the first bank; with arm & 2 the second bank (per-wave cycle stamps of one block
 * step of the sweep); with arm & 4 all 1024 values (out must hold them): both banks and, from 128 on, the start / end stamp
 * of every workgroup of the last k_step launch (scripts/phase_profile*.py). */
int drlgx_debug_phase_clocks_host(drlgx_engine *e, int arm, int64_t *out /* 64 values; 1024 with arm & 4 (read from the buffer's start) */);

/* Development aid: the tile of C per workgroup the GCN's fp32 GEMM dispatcher picks for an m x n product computed in k_slices
 * K-slices (transpose_a: A is stored [K x M], the weight-gradient products), as 1000 * columns + rows: 64064 = the 64x64
 * kernels, 128096..128160 / 64096..64160 = the tall-tile kernel k_gemm_wide with 8 / 4 waves (csrc/k_gcn.hip).  The numerics
 * tests use it to prove that they cover every compiled tile. */
int drlgx_debug_gemm_tile_rows(int m, int n, int k_slices, int transpose_a);

/* Development aid: picks the form of the stand-alone virtual-map kernel (VirtualMap::updateProbability / updateInformation,
 * VirtualMap.cpp:61-84,256-316) for every later launch of this process: 1 = the two-workgroups-per-CU form, 0 = the
 * resident form, -1 (the default; the environment variable DRLGX_MAP_COMPACT sets the initial value, read once) = by launch
 * size: two per CU when there are more instances than CUs.  Returns the previous value.  Both forms are bit-equal; the
 * parity test switches between them with this call. */
int drlgx_debug_map_form(int form);

/* The incremental belief update (csrc/k_inc.hip): between relinearisations of the iSAM2 policy (SLAM2D.cpp:10-12: every
 * 10th update, |delta| >= 0.1) SLAM2D::optimize (SLAM2D.cpp:374-430) only gains the new pose's odometry factor and the
 * step's bearing-range factors, so the engine applies the covariance-form update of FastMarginals2::propagate / update
 * (FastMarginals.cpp:188-321) to a per-instance covariance panel instead of re-solving; updates that relinearise, re-based
 * look-ahead copies (SLAM2D::set_copy_isam, SLAM2D.cpp:490-497) and anything the panel cannot serve take the full solve.
 * On by default; DRLGX_INCREMENTAL=0 in the environment of drlgx_create disables it (every update a full solve), as does a
 * panel memory need beyond DRLGX_INC_MAX_GB (default 32).  out[0] = updates served by the rank-k path, out[1] = full
 * solves, since creation or the last call with reset != 0; both -1 when the path is disabled. */
int drlgx_inc_stats_host(drlgx_engine *e, int64_t out[2], int reset);

/* ---- GCN policy (scripts/Networks.py:12-70 over PyG GCNConv(improved=True)) ------------------- */

/* Forward of GCN / PolicyGCN trunk: H1 = relu(Â X W1 + b1), H2 = relu(Â H1 W2 + b2) [* dropout mask],
 * out = H2 Wf^T + bf, with Â = D^-1/2 (A_w + 2I) D^-1/2 built from (edge_index, edge_attr).
 * All pointers DEVICE, fp32 (edge_index int64 as PyG).  hidden = W1 cols, out_dim = Wf rows.
 * ws_dev: workspace of drlgx_gcn_workspace_bytes(); it keeps what the backward call needs (AX, b1, AH1, H2, both CSRs). */
size_t drlgx_gcn_workspace_bytes(int n_nodes, int n_edges, int hidden, int out_dim);
int drlgx_gcn_forward(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim,
                      const float *x, const int64_t *edge_index, const float *edge_attr, const float *W1,
                      const float *b1, const float *W2, const float *b2, const float *Wf, const float *bf,
                      const float *dropout_mask /* [n_nodes*hidden] or NULL */, float *out /* [n_nodes*out_dim] */,
                      void *ws_dev);
/* The same forward for a BATCH of graphs whose boundaries the caller knows (a PyG `Batch` / what drlgx_graph and
 * drlgx_replay_collate emit): graph g owns nodes [node_off[g], node_off[g+1]) and edges [edge_off[g], edge_off[g+1]),
 * DEVICE int32 [n_graphs + 1]; every edge connects two nodes of its own graph (others are ignored); no graph has more
 * than max_edges_per_graph edges (a host-side bound the caller knows: the export's capacity, the collation's counts).
 * Same results bit for bit; the graph normalisation / CSR build is one launch (one workgroup per graph, its edges
 * sorted in LDS) instead of eight; bounds beyond 16 384 edges per graph take the generic build. */
int drlgx_gcn_forward_batched(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim,
                              const float *x, const int64_t *edge_index, const float *edge_attr, const float *W1,
                              const float *b1, const float *W2, const float *b2, const float *Wf, const float *bf,
                              const float *dropout_mask, float *out, void *ws_dev, int n_graphs,
                              const int32_t *node_off, const int32_t *edge_off, int max_edges_per_graph);
/* Backward: given d(out) returns gradients of all six parameter tensors (accumulated = overwritten). */
int drlgx_gcn_backward(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim,
                       const float *x, const int64_t *edge_index, const float *edge_attr, const float *W1,
                       const float *W2, const float *Wf, const float *dropout_mask, const float *d_out,
                       float *dW1, float *db1, float *dW2, float *db2, float *dWf, float *dbf, void *ws_dev);

/* ---- DQN update (scripts/policy.py:137-178, :234-253): the pieces between the two GCN calls ---------------- */

/* Mini-batch collation of replay graphs held in a device pool = torch_geometric DataLoader(s_j_batch, batch_size=BATCH)
 * (scripts/policy.py:146-153): graph g is rows [node_start, +node_cnt) of pool_x ([rows][in_dim] f32) and columns
 * [edge_start, +edge_cnt) of pool_ei ([2][pool_edges] i64) / pool_ea, its node ids start at loc.  desc_dev int64
 * [5][n_graphs] = node_start, node_cnt, edge_start, edge_cnt, loc.  Outputs (DEVICE, sized by the caller from the counts):
 * x_out [N][in_dim], ei_out [2][n_edges_total] (ids shifted by the cumulative node counts), ea_out, batch_out [N], and the
 * graph boundaries in the form drlgx_gcn_forward_batched takes.  pool_q / q_out (optional): one float per pooled node
 * gathered the same way - the target network's read-out, which depends on the graph and the (frozen) target weights only and
 * is therefore evaluated once per stored graph and target refresh instead of once per mini-batch (scripts/policy.py:154-156);
 * with x_out NULL only this value is gathered. */
int drlgx_replay_collate(void *hip_stream, int n_graphs, const int64_t *desc_dev, const float *pool_x, int in_dim,
                         const int64_t *pool_ei, int64_t pool_edges, const float *pool_ea, float *x_out, int64_t *ei_out,
                         int64_t n_edges_total, float *ea_out, int64_t *batch_out,
                         int32_t *node_off_out /* [n_graphs + 1] or NULL */, int32_t *edge_off_out /* [n_graphs + 1] or NULL */,
                         const float *pool_q /* [rows] or NULL */, float *q_out /* [N] or NULL */);
/* The two collations of one DQN mini-batch in one launch (scripts/policy.py:146-156): the current states s_j as
 * drlgx_replay_collate emits them, and - for the next states s_j1, desc2_dev int64 [5][n_graphs] - only the cached per-node
 * value pool_q gathered into q2_out [sum of their node counts] (the target network's read-out over the collated s_j1). */
int drlgx_replay_collate_pair(void *hip_stream, int n_graphs, const int64_t *desc_dev, const float *pool_x, int in_dim,
                              const int64_t *pool_ei, int64_t pool_edges, const float *pool_ea, float *x_out, int64_t *ei_out,
                              int64_t n_edges_total, float *ea_out, int64_t *batch_out, int32_t *node_off_out, int32_t *edge_off_out,
                              const int64_t *desc2_dev, const float *pool_q, float *q2_out);
/* TD targets, scripts/policy.py:154-175: sample i takes max(q1[lo_i:hi_i]) (float32, the target network's read-out over
 * the collated next states; the caller resolves the reference's slicing into [lo, hi)), and
 * a_batch[pos_i] = 1, y_batch[pos_i] = r_i + gamma max  (r_i alone when terminal_i) in float64; both vectors
 * [n_nodes_total] are zeroed first.  meta_dev int64 [4][n_samples] = lo, hi, pos, terminal; r_dev double [n_samples]. */
int drlgx_dqn_targets(void *hip_stream, int n_samples, const float *q1, const int64_t *meta_dev, const double *r_dev,
                      double gamma, int64_t n_nodes_total, double *a_batch, double *y_batch);
/* DeepQ.cost (scripts/policy.py:234-239) and d(cost)/d(pred): loss_out[0] = sum (pred a - y)^2 / batch (float64),
 * d_pred float32 [n_nodes] - what loss.backward() hands to the network (scripts/policy.py:249). */
int drlgx_dqn_loss_grad(void *hip_stream, int n_nodes, const float *pred, const double *action, const double *y,
                        double batch, double *loss_out, float *d_pred);
/* `param.grad.data.clamp_(-c, c)` + torch.optim.Adam.step() (scripts/policy.py:250-253; lr as given, no weight decay /
 * amsgrad) for up to 8 fp32 tensors in one launch.  HOST arrays of DEVICE pointers; step = 1 for the first update;
 * grad_clamp <= 0 disables the clamp. */
int drlgx_adam_step(void *hip_stream, int n_tensors, float *const *params, const float *const *grads,
                    float *const *exp_avg, float *const *exp_avg_sq, const int64_t *sizes, double lr, double beta1,
                    double beta2, double eps, int64_t step, double grad_clamp);
/* ... with the gradient multiplied by grad_scale in front of the clamp: the 1 / world-size of a SUM all-reduced gradient
 * (scripts/policy.py has one process; the build's data-parallel trainer averages over ranks - SURVEY.md 8e) folded in. */
int drlgx_adam_step_scaled(void *hip_stream, int n_tensors, float *const *params, const float *const *grads,
                    float *const *exp_avg, float *const *exp_avg_sq, const int64_t *sizes, double lr, double beta1,
                    double beta2, double eps, int64_t step, double grad_clamp, double grad_scale);

/* Replay graphs never change once stored, so what drlgx_gcn_forward_batched derives from a graph - weighted degrees, self
 * weights, both CSRs with the normalised weights D^-1/2 (A_w + 2I) D^-1/2 (GCNConv(improved=True), scripts/Networks.py:15-16)
 * and Â X - is computed ONCE per stored export and kept beside the pooled graphs, graph-local (row starts / ends relative to
 * the graph's first edge, neighbour ids to its first node).  DEVICE arrays over the pool's node rows / edge columns. */
typedef struct drlgx_csr_cache {
  float *deg, *selfw, *ax;                         /* [nodes], [nodes], [nodes][8] */
  int32_t *ptr_dst, *end_dst, *ptr_src, *end_src;  /* [nodes] */
  int32_t *nbr_dst, *nbr_src;                      /* [edges] */
  float *wn_dst, *wn_src;                          /* [edges] */
} drlgx_csr_cache;
/* Fill the cache for one export of n_graphs graphs (boundaries node_off / edge_off, DEVICE int32 [n_graphs + 1], relative to
 * the export's first node / edge).  x, edge_index (second row at + edge_row_stride), edge_attr and every array of `cache`
 * point AT the export's first node / edge.  DRLGX_E_CAPACITY: a graph has more edges than the per-graph kernel sorts
 * (16 384) - the caller keeps that export uncached and collates it the generic way. */
int drlgx_replay_cache_csr(void *hip_stream, int n_graphs, const int32_t *node_off, const int32_t *edge_off, int max_edges_per_graph,
                           const float *x, int in_dim, const int64_t *edge_index, int64_t edge_row_stride, const float *edge_attr,
                           const drlgx_csr_cache *cache);
/* Mini-batch collation straight into the graph part of a GCN workspace (ws_dev of drlgx_gcn_workspace_bytes(n_nodes,
 * n_edges, ...)): the cached arrays of graph g (desc_dev as drlgx_replay_collate; `cache` points at the POOL's first node /
 * edge) land at its cumulative offsets with row bounds / neighbour ids shifted - bit for bit what
 * drlgx_gcn_forward_batched builds for the collated batch.  desc2_dev / pool_q / q2_out: as drlgx_replay_collate_pair
 * (or all NULL).  Then drlgx_gcn_forward_prebuilt runs the forward without touching x / edge_index / edge_attr. */
int drlgx_gcn_collate_csr(void *hip_stream, int n_graphs, const int64_t *desc_dev, const drlgx_csr_cache *cache, int n_nodes,
                          int n_edges, int hidden, int out_dim, void *ws_dev, int32_t *node_off_out, int32_t *edge_off_out,
                          const int64_t *desc2_dev, const float *pool_q, float *q2_out);
int drlgx_gcn_forward_prebuilt(void *hip_stream, int n_nodes, int n_edges, int in_dim, int hidden, int out_dim, const float *W1,
                               const float *b1, const float *W2, const float *b2, const float *Wf, const float *bf,
                               const float *dropout_mask, float *out, void *ws_dev);

/* One DQN update as TWO host calls (scripts/policy.py:139-178 and :234-249).  Nothing new runs on the device - the launches of
 * drlgx_replay_collate_pair + drlgx_dqn_targets (prepare), and of drlgx_gcn_forward_batched + drlgx_dqn_loss_grad +
 * drlgx_gcn_backward (forward_backward), are issued back to back on the caller's stream with every intermediate in ONE
 * caller-owned DEVICE arena of drlgx_dqn_arena_bytes(): the 256-env trainer was bound by the host work of those launches.
 * The arena's layout is a function of the capacities (cap_nodes / cap_edges of the collated current states, cap_nodes1 of the
 * next states' read-out) only; both calls of an update - and drlgx_dqn_arena_views, for read-backs - must pass the same ones.
 * views[12] = x, edge_index [2][n_edges], edge_attr, batch, node_off, edge_off, q1, a_batch, y_batch, out, d_out, loss.
 * params / grads: HOST arrays of six DEVICE pointers (W1 b1 W2 b2 Wf bf and their gradients, written not accumulated).
 * cache (or NULL): the pool's drlgx_csr_cache - prepare then collates the cached graph data straight into the GCN workspace
 * (drlgx_gcn_collate_csr) instead of x / edge_index / edge_attr, and forward_backward (same `cache`-ness) skips the build. */
size_t drlgx_dqn_arena_bytes(int n_graphs, int64_t cap_nodes, int64_t cap_edges, int64_t cap_nodes1, int in_dim, int hidden,
                             int out_dim);
int drlgx_dqn_arena_views(void *arena_dev, int n_graphs, int64_t cap_nodes, int64_t cap_edges, int64_t cap_nodes1, int in_dim,
                          int hidden, int out_dim, void **views);
int drlgx_dqn_prepare(void *hip_stream, int n_graphs, const int64_t *desc_dev, const int64_t *desc1_dev, const float *pool_x,
                      int in_dim, const int64_t *pool_ei, int64_t pool_edges, const float *pool_ea, const float *pool_q,
                      int64_t n_nodes, int64_t n_edges, int64_t n_nodes1, const int64_t *meta_dev, const double *r_dev,
                      double gamma, void *arena_dev, int64_t cap_nodes, int64_t cap_edges, int64_t cap_nodes1, int hidden,
                      int out_dim, const drlgx_csr_cache *cache);
int drlgx_dqn_forward_backward(void *hip_stream, int n_graphs, int64_t n_nodes, int64_t n_edges, int max_edges_per_graph,
                               int in_dim, int hidden, int out_dim, const float *const *params, const float *dropout_mask,
                               double batch, float *const *grads, void *arena_dev, int64_t cap_nodes, int64_t cap_edges,
                               int64_t cap_nodes1, int graph_prebuilt);

/* ---- env wrapper / actor-critic heads ------------------------------------------------------------------------- */

/* The normalisation of ExplorationEnv.rewards_all_goals (scripts/envs/exploration_env.py:151-161) for every env: raw
 * look-ahead rewards of env e's frontiers at [cand_first[e], +n_frontier[e]) -> np.interp onto [-1, 0] (nearest frontier
 * is the first arg-max, loop_clo 0) or [-1, 1] (loop_clo 1).  All pointers DEVICE. */
int drlgx_normalise_rewards(void *hip_stream, int n_envs, const double *raw, const int64_t *cand_first,
                            const int32_t *n_frontier, double *out, uint8_t *loop_clo);
/* PolicyGCN head (scripts/Networks.py:47-50): masked_select + torch_geometric.utils.softmax over every graph's selected
 * nodes.  q / mask over all N nodes, graph boundaries node_off int32 [n_graphs + 1]; p_out holds the selected nodes of
 * all graphs in node order (sum(mask) values).  Backward: d_q over all N nodes (0 outside the mask). */
int drlgx_segment_softmax(void *hip_stream, int n_graphs, const int32_t *node_off, const float *q, const uint8_t *mask,
                          float *p_out);
int drlgx_segment_softmax_backward(void *hip_stream, int n_graphs, const int32_t *node_off, const float *p,
                                   const float *d_p, const uint8_t *mask, float *d_q);
/* ValueGCN head (scripts/Networks.py:66-70): global_mean_pool(h, batch).mean(dim=1) of the [N][n_cols] read-out, one
 * value per graph; backward d_h [N][n_cols]. */
int drlgx_mean_pool(void *hip_stream, int n_graphs, const int32_t *node_off, const float *h, int n_cols, float *v_out);
int drlgx_mean_pool_backward(void *hip_stream, int n_graphs, const int32_t *node_off, const float *d_v, int n_cols,
                             float *d_h);

#ifdef __cplusplus
}
#endif
#endif /* DRLGX_H */
