// Small kernels around the belief step: instance copies for the look-ahead (deep copies of
// SLAM2D / Simulator2D state, Planner2D.cpp:1417-1420), SLAM2D::set_copy_isam re-basing
// (SLAM2D.cpp:490-497), rewards / utility (Planner2D.cpp:354-366, 1463-1464) and the closed-form
// line planner (Planner2D.cpp:937-1041).
#include "drlgx_dev.h"
#include "drlgx_fields.h"

namespace {

// the incremental update's covariance panel (csrc/k_inc.hip) of instance s -> d: the live rows / columns only.  Block
// `part` of kCopySplit copies every kCopySplit-th group of 8 rows, 32 threads x 16 bytes per row, four rows' loads in
// flight per thread
constexpr int kCopySplit = 4;
__device__ __forceinline__ void copy_panel_part(const DrlgxState &S, int s, int d, int part) {
  const int *ms = S.jc_meta + (size_t)s * 4;
  int *md = S.jc_meta + (size_t)d * 4;
  const int valid = ms[0], P = ms[1], L = ms[2], M = ms[3];
  if (threadIdx.x == 0 && part == 0) {
    md[0] = valid; md[1] = P; md[2] = L; md[3] = M;
  }
  if (valid != 1) return;
  const double *ps = S.jc + (size_t)s * S.jc_stride;
  double *pd = S.jc + (size_t)d * S.jc_stride;
  const int npair = (3 + 2 * L + 1) >> 1, rows = 3 * P + 2 * L;
  const int cp0 = threadIdx.x & 31, r0 = (threadIdx.x >> 5) + 8 * part, rs = 8 * kCopySplit;
  auto roff = [&](int q) -> size_t { return (size_t)(q < 3 * P ? q : q - 3 * P + 3 * S.P_max) * S.jc_ld; };
  for (int cp = cp0; cp < npair; cp += 32)
    for (int q = r0; q < rows; q += 4 * rs) {
      const int q1 = q + rs, q2 = q + 2 * rs, q3 = q + 3 * rs;
      const size_t o0 = roff(q), o1 = roff(min(q1, rows - 1)), o2 = roff(min(q2, rows - 1)), o3 = roff(min(q3, rows - 1));
      const double2 v0 = reinterpret_cast<const double2 *>(ps + o0)[cp], v1 = reinterpret_cast<const double2 *>(ps + o1)[cp];
      const double2 v2 = reinterpret_cast<const double2 *>(ps + o2)[cp], v3 = reinterpret_cast<const double2 *>(ps + o3)[cp];
      reinterpret_cast<double2 *>(pd + o0)[cp] = v0;
      if (q1 < rows) reinterpret_cast<double2 *>(pd + o1)[cp] = v1;
      if (q2 < rows) reinterpret_cast<double2 *>(pd + o2)[cp] = v2;
      if (q3 < rows) reinterpret_cast<double2 *>(pd + o3)[cp] = v3;
    }
  if (part == 0)
    for (int e = threadIdx.x; e < 6 * P; e += 256) S.jd[(size_t)d * S.P_max * 6 + e] = S.jd[(size_t)s * S.P_max * 6 + e];
}

// copy every field of instance src[i] to dst[i] whose class is not in skip_mask and - panel != 0 - its covariance panel:
// one launch for everything an instance copy (snapshot / restore, env -> base -> rollout) moves
__global__ __launch_bounds__(256) void k_copy_instances(const DrlgxField *fields, int n_fields, const int32_t *src,
                                                        const int32_t *dst, int src_off, int dst_off, int skip_mask, const int *cnt,
                                                        DrlgxState S, int panel) {
  // grid = (instances, fields [+ kCopySplit]): every (instance, field) slice is streamed by its own workgroup with
  // 16-byte accesses when the slice is 16-byte aligned (all large fields are)
  // (x = field, y = instance: consecutive workgroups stream slices of DIFFERENT arrays - with x = instance the workgroups in flight
  // all walked the same field at the same offsets of 256 instances: 20.8 against 18.0 us per 78 MB restore, profiles/r06_ab_copy.txt)
  // (a 1-D grid dealt so that instance i's slices are copied on XCD i % 8 - where the belief kernels' workgroup of instance i runs
  // next - changed neither kernel's time)
  const int i = blockIdx.y;
  const int f = blockIdx.x;
  const int s = (src ? src[i] : i) + src_off, d = (dst ? dst[i] : i) + dst_off;
  if (f >= n_fields) {
    if (panel) copy_panel_part(S, s, d, f - n_fields);
    return;
  }
  // (one workgroup for ALL the fields of a few bytes per instance was measured: 21.4 against 17.8 us - seven dependent round trips)
  if (fields[f].cls & skip_mask) return;
  const size_t stride = fields[f].stride;
  const char *sb = fields[f].base + (size_t)s * stride;
  char *db = fields[f].base + (size_t)d * stride;
  // the live part: per-pose / per-landmark / per-factor arrays end at the source instance's counts
  size_t live = fields[f].pad > 0 ? (size_t)fields[f].pad : stride;  // (pad: a sub-slice of the instance's slice - one plane of the virtual map's information)
  if (fields[f].unit && cnt) {
    const int *c = cnt + (size_t)s * DRLGX_CNT_STRIDE;
    const int k = fields[f].unit == 1 ? c[C_P] : fields[f].unit == 2 ? c[C_L] : c[C_M];
    const size_t b = ((size_t)(k > 0 ? k : 0) * (size_t)fields[f].unit_bytes + 15) & ~(size_t)15;
    live = b < stride ? b : stride;
  }
  if ((stride & 15) == 0) {
    const uint4 *sp = reinterpret_cast<const uint4 *>(sb);
    uint4 *dp = reinterpret_cast<uint4 *>(db);
    const size_t nq = live / 16;
    size_t k = threadIdx.x;
    for (; k + 256 < nq; k += 512) {  // (two loads in flight per thread)
      const uint4 v0 = sp[k], v1 = sp[k + 256];
      dp[k] = v0; dp[k + 256] = v1;
    }
    if (k < nq) dp[k] = sp[k];
  } else {
    const uint32_t *sp = reinterpret_cast<const uint32_t *>(sb);
    uint32_t *dp = reinterpret_cast<uint32_t *>(db);
    const size_t nw = (live < stride ? live : stride) / 4;
    for (size_t k = threadIdx.x; k < nw; k += 256) dp[k] = sp[k];
  }
}

// What a status read brings to the host, packed for ONE copy: [status word, every env's pose count] (head bytes, a multiple of 16),
// then the caller's bytes (drlgx_status_fetch_host).  Separate copies - a word, a strided column of the counters, the caller's
// buffer - cost more in submission than this launch: 40 us per read on an idle stream against ~15 for a single copy.
__global__ __launch_bounds__(256) void k_fetch_pack(const int *status, const int *cnt, int n_envs, int head, const unsigned char *src,
                                                    size_t bytes, unsigned char *out) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
  int *h = reinterpret_cast<int *>(out);
  for (size_t i = t; i < (size_t)n_envs + 1; i += nt) h[i] = i == 0 ? status[0] : cnt[(i - 1) * DRLGX_CNT_STRIDE + C_P];
  unsigned char *o = out + head;
  if (((reinterpret_cast<uintptr_t>(src) | bytes) & 15) == 0) {
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *o4 = reinterpret_cast<uint4 *>(o);
    for (size_t i = t; i < bytes / 16; i += nt) o4[i] = s4[i];
  } else {
    for (size_t i = t; i < bytes; i += nt) o[i] = src[i];
  }
}

// SLAM2D::set_copy_isam: theta := calculateBestEstimate(), delta := 0, fresh ISAM2 (update count 0)
__global__ __launch_bounds__(64) void k_rebase(DrlgxState S, int base0, int n) {
  const int inst = base0 + blockIdx.x;
  int *cnt = S.cnt + (size_t)inst * DRLGX_CNT_STRIDE;
  const int P = cnt[C_P], L = cnt[C_L];
  for (int i = threadIdx.x; i < P; i += 64) {
    for (int k = 0; k < 4; ++k)
      S.th_pose[((size_t)inst * S.P_max + i) * 4 + k] = S.est_pose[((size_t)inst * S.P_max + i) * 4 + k];
    for (int k = 0; k < 3; ++k) S.d_pose[((size_t)inst * S.P_max + i) * 3 + k] = 0.0;
  }
  for (int j = threadIdx.x; j < L; j += 64) {
    for (int k = 0; k < 2; ++k) {
      S.th_lm[((size_t)inst * S.L_max + j) * 2 + k] = S.est_lm[((size_t)inst * S.L_max + j) * 2 + k];
      S.d_lm[((size_t)inst * S.L_max + j) * 2 + k] = 0.0;
    }
  }
  if (threadIdx.x == 0) {
    cnt[C_ISAM] = 0;
    cnt[C_NEWP] = 0;
    cnt[C_NEWL] = 0;
    cnt[C_FLAG] = 0;
    if (S.jc_meta) S.jc_meta[(size_t)inst * 4] = 0;  // the re-based system is solved in full (and leaves a fresh covariance panel)
  }
}

// after copying base -> rollout: result_ (est_pose) is the ORIGINAL estimate of the live env
// (SLAM2D copy keeps result_; set_copy_isam does not touch it), distance accumulator reset.
__global__ __launch_bounds__(64) void k_fix_rollouts(DrlgxState S, const int32_t *cand_env, int roll0) {
  const int c = blockIdx.x;
  const int env = cand_env[c], inst = roll0 + c;
  const int P = S.cnt[(size_t)inst * DRLGX_CNT_STRIDE + C_P];
  for (int e = threadIdx.x; e < P * 4; e += 64)
    S.est_pose[(size_t)inst * S.P_max * 4 + e] = S.est_pose[(size_t)env * S.P_max * 4 + e];
  if (threadIdx.x == 0) {
    S.parent[inst] = env;
    S.red[(size_t)inst * DRLGX_RED_STRIDE + R_DIST] = 0.0;
  }
}

__device__ __forceinline__ double utility_of(const DrlgxState &S, int inst, double dist) {
  const double *red = S.red + (size_t)inst * DRLGX_RED_STRIDE;
  const drlgx_config &cfg = S.cfg;
  const double pk = red[R_KNOWN] / (double)S.V;
  const double dw = cfg.distance_weight0 - (cfg.distance_weight0 - cfg.distance_weight1) * pk;
  return red[R_UTR] + dist * dw;
}

__global__ void k_rewards(DrlgxState S, int n_cand, const int32_t *cand_env, int roll0, double *rewards) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cand) return;
  const int inst = roll0 + c;
  const double u0 = utility_of(S, cand_env[c], 0.0);
  const double u1 = utility_of(S, inst, S.red[(size_t)inst * DRLGX_RED_STRIDE + R_DIST]);
  rewards[c] = u0 - u1;
}

// mode 0: calculateUtility(dist); 1: explored(); 2: uncertainty_EM trace-weighted; 3: uncertainty_EM det
__global__ void k_utility(DrlgxState S, const double *dist, double *out, int mode) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S.n_envs) return;
  const double *red = S.red + (size_t)e * DRLGX_RED_STRIDE;
  if (mode == 0) out[e] = utility_of(S, e, dist ? dist[e] : 0.0);
  else if (mode == 1) out[e] = red[R_EXPL] / (double)S.count_explored;
  else if (mode == 2) out[e] = red[R_UWTR];
  else out[e] = red[R_UDET];
}

// EMPlanner2D::line_planner (Planner2D.cpp:937-1041), goal = frontier point
__global__ void k_line_plan(DrlgxState S, int n_cand, const int32_t *cand_env, const double *goal, double *actions,
                            int32_t *n_actions) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cand) return;
  const int env = cand_env[c];
  const int P = S.cnt[(size_t)env * DRLGX_CNT_STRIDE + C_P];
  const double *ep = S.est_pose + ((size_t)env * S.P_max + (P - 1)) * 4;
  const double rx = ep[0], ry = ep[1];
  double rth = atan2(ep[3], ep[2]);
  const double gx = goal[2 * c], gy = goal[2 * c + 1];
  double gth = atan2(gy - ry, gx - rx);
  const double PI = 3.14159265358979323846;
  if (rth < 0) rth = PI * 2 + rth;
  if (gth < 0) gth = PI * 2 + gth;
  const double dr = 180 * PI / 180;
  double diff = gth - rth;
  double *out = actions + (size_t)c * S.A_max * 3;
  int n = 0;
  double d, sign;
  if (diff > PI) {
    d = 2 * PI - diff;
    sign = -1.0;
  } else if (diff > -PI && diff < 0) {
    d = fabs(diff);
    sign = -1.0;
  } else if (diff <= -PI) {
    d = 2 * PI - fabs(diff);
    sign = 1.0;
  } else {
    d = diff;
    sign = 1.0;
  }
  auto push = [&](double x, double y, double th) {
    if (n < S.A_max) {
      // a Pose2 action: theta() = atan2(sin, cos)
      out[3 * n] = x;
      out[3 * n + 1] = y;
      out[3 * n + 2] = atan2(sin(th), cos(th));
    }
    n++;
  };
  {
    const int q = (int)(d / dr);
    const double rem = d - dr * q;
    for (int i = 0; i < q; ++i) push(0, 0, sign * dr);
    push(0, 0, sign * rem);
  }
  const double dist = sqrt(pow(rx - gx, 2) + pow(ry - gy, 2));
  const int dq = (int)(dist / S.cfg.max_edge_length);
  const double drem = dist - dq * S.cfg.max_edge_length;
  for (int i = 0; i < dq; ++i) push(S.cfg.max_edge_length, 0, 0);
  push(drem, 0, 0);
  n_actions[c] = n;
  if (n > S.A_max) atomicMin(S.status, DRLGX_E_CAPACITY);
}

// The metric trio of the reference's evaluation script, one 256-thread workgroup per environment:
//   out[3 e + 0] landmark error   (exploration_env.py:170-177): (sum_j |gt[key_j] - est_j| + sigma0 (n_gt - L)) / n_gt
//   out[3 e + 1] map entropy      (scripts/test.py:61-74): -sum_v p ln p + 0.5 ln(0.5) (V - interior cells)
//   out[3 e + 2] max localisation uncertainty (exploration_env.py:190-194): max over the poses of tr(marginal covariance)
__global__ __launch_bounds__(256) void k_metrics(DrlgxState S, double sigma0, double *out) {
  __shared__ double red[3][4];
  const int e = blockIdx.x, tid = threadIdx.x;
  const int *cnt = S.cnt + (size_t)e * DRLGX_CNT_STRIDE;
  const int P = cnt[C_P], L = cnt[C_L];
  const double *gt = S.gt_lm + (size_t)S.parent[e] * S.LG * 2;
  const double *el = S.est_lm + (size_t)e * S.L_max * 2;
  const int *key = S.lm_key + (size_t)e * S.L_max;
  double err = 0, ent = 0, mx = 0;
  for (int j = tid; j < L; j += 256) {
    const double dx = gt[2 * key[j]] - el[2 * j], dy = gt[2 * key[j] + 1] - el[2 * j + 1];
    err += sqrt(dx * dx + dy * dy);
  }
  const double *prob = S.vm_prob + (size_t)e * S.V;
  for (int v = tid; v < S.V; v += 256) ent += prob[v] * log(prob[v]);
  const double *ptr = S.pose_tr + (size_t)e * S.P_max;
  for (int i = tid; i < P; i += 256) mx = fmax(mx, ptr[i]);
  for (int o = 32; o > 0; o >>= 1) {
    err += __shfl_down(err, o);
    ent += __shfl_down(ent, o);
    mx = fmax(mx, __shfl_down(mx, o));
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = err;
    red[1][tid >> 6] = ent;
    red[2][tid >> 6] = mx;
  }
  __syncthreads();
  if (tid == 0) {
    const int n_gt = S.cfg.num_landmarks;
    const double es = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double hs = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    out[3 * e] = (es + sigma0 * (n_gt - L)) / n_gt;
    out[3 * e + 1] = -hs + 0.5 * log(0.5) * (double)(S.V - S.count_explored);
    out[3 * e + 2] = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
  }
}

// VirtualMap::toCovArray (VirtualMap.cpp:140-151): per cell the larger eigenvalue of the 2x2 covariance (its square
// root clipped at sigma0) and the direction of its eigenvector.  Eigen's SelfAdjointEigenSolver returns a rotation
// (cos > 0) with the columns sorted by eigenvalue, so the eigenvector's sign is fixed by: x component > 0 when the
// larger eigenvalue belongs to the first axis, y component > 0 otherwise (recalled from Eigen; used for rendering only).
__global__ __launch_bounds__(256) void k_cov_array(DrlgxState S, double *length, double *angle) {
  const int e = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
  if (v >= S.V) return;
  const double *inf = S.vm_info + (size_t)e * 3 * S.V;
  double a, b, d;
  inv2_llt_s(inf[v], inf[S.V + v], inf[2 * S.V + v], a, b, d);  // covariance()
  const double half = 0.5 * (a - d), r = sqrt(half * half + b * b);
  const double lmax = 0.5 * (a + d) + r;
  double vx, vy;
  if (a >= d) {  // (lmax - d, b) is parallel to the eigenvector and its x component is >= 0
    vx = half + r; vy = b;
    if (vx == 0.0 && vy == 0.0) { vx = 1.0; vy = 0.0; }
  } else {       // (b, lmax - a): y component >= 0
    vx = b; vy = r - half;
  }
  length[(size_t)e * S.V + v] = fmin(sqrt(lmax), S.cfg.sigma0);
  angle[(size_t)e * S.V + v] = atan2(vy, vx);
}

}  // namespace

void drlgx_launch_metrics(const DrlgxState &S, hipStream_t st, double sigma0, double *out) {
  hipLaunchKernelGGL(k_metrics, dim3(S.n_envs), dim3(256), 0, st, S, sigma0, out);
}
void drlgx_launch_cov_array(const DrlgxState &S, hipStream_t st, double *length, double *angle) {
  hipLaunchKernelGGL(k_cov_array, dim3((S.V + 255) / 256, S.n_envs), dim3(256), 0, st, S, length, angle);
}
void drlgx_launch_copy(const DrlgxField *fields_dev, int n_fields, hipStream_t st, int n, const int32_t *src,
                       const int32_t *dst, int src_off, int dst_off, int skip_mask, const int *cnt, const DrlgxState *panel) {
  if (n <= 0) return;
  const bool with_panel = panel && panel->jc;
  const dim3 cgrid(n_fields + (with_panel ? kCopySplit : 0), n);
  hipLaunchKernelGGL(k_copy_instances, cgrid, dim3(256), 0, st, fields_dev, n_fields, src, dst,
                     src_off, dst_off, skip_mask, cnt, with_panel ? *panel : DrlgxState{}, with_panel ? 1 : 0);
}
void drlgx_launch_fetch_pack(const DrlgxState &S, hipStream_t st, const void *src, size_t bytes, void *out) {
  const size_t head = drlgx_fetch_head_bytes(S.n_envs);
  const int blocks = (int)std::min<size_t>(256, (head + bytes + 4095) / 4096);
  hipLaunchKernelGGL(k_fetch_pack, dim3(std::max(blocks, 1)), dim3(256), 0, st, S.status, S.cnt, S.n_envs, (int)head,
                     reinterpret_cast<const unsigned char *>(src), bytes, reinterpret_cast<unsigned char *>(out));
}
void drlgx_launch_rebase(const DrlgxState &S, hipStream_t st, int base0, int n) {
  hipLaunchKernelGGL(k_rebase, dim3(n), dim3(64), 0, st, S, base0, n);
}
void drlgx_launch_fix_rollouts(const DrlgxState &S, hipStream_t st, int n_cand, const int32_t *cand_env, int roll0) {
  hipLaunchKernelGGL(k_fix_rollouts, dim3(n_cand), dim3(64), 0, st, S, cand_env, roll0);
}
void drlgx_launch_rewards(const DrlgxState &S, hipStream_t st, int n_cand, const int32_t *cand_env, int roll0,
                          double *rewards) {
  hipLaunchKernelGGL(k_rewards, dim3((n_cand + 127) / 128), dim3(128), 0, st, S, n_cand, cand_env, roll0, rewards);
}
void drlgx_launch_utility(const DrlgxState &S, hipStream_t st, const double *dist, double *out, int mode) {
  hipLaunchKernelGGL(k_utility, dim3((S.n_envs + 127) / 128), dim3(128), 0, st, S, dist, out, mode);
}
void drlgx_launch_line_plan(const DrlgxState &S, hipStream_t st, int n_cand, const int32_t *cand_env, const double *goal,
                            double *actions, int32_t *n_actions) {
  hipLaunchKernelGGL(k_line_plan, dim3((n_cand + 127) / 128), dim3(128), 0, st, S, n_cand, cand_env, goal, actions,
                     n_actions);
}
