// Graph export for the policy network, batched over environments (one 256-thread workgroup per env).
//
// Restates, on the device: ExplorationEnv.frontier + graph_matrix
// (scripts/envs/exploration_env.py:196-348), SLAM2D::adjacency_degree_get
// (src/em_exploration/SLAM2D.cpp:198-273) and DeepQ.data_process (scripts/policy.py:211-232), and
// concatenates the per-env graphs the way torch_geometric's DataLoader/Batch does.
//   node order : landmarks by ground-truth key | poses by index | frontier nodes (order of first use)
//   edge order : data_process's row-major first-seen order = undirected pairs (i<j) sorted by (i,j),
//                each emitted as (i,j) then (j,i)
//   features   : [trace of marginal covariance, distance to robot, bearing difference in [0,2pi),
//                 occupancy probability at the node's cell, type (-1 / 0 / +1)]
// Three launches: build (per env: frontiers, attachments, counts) -> scan (offsets) -> emit.
// Compiled with -ffp-contract=off: thresholds (p<0.45, 0.49<p<0.51) and nearest-frontier ties must
// resolve exactly as the numpy reference does.
#include "drlgx_dev.h"

namespace {

constexpr int kT = 256;

struct GraphBufs {
  int *gi;     // [n_envs][gi_stride]: 0 N, 1 E, 2 F, 3 nf(all frontier cells), 4.. node_of_slot[L_max], slot_of_node[L_max],
               //                       lm_front[L_max], fsel[L_max+1]
  int gi_stride;
};

__device__ __forceinline__ void cell_xy(const DrlgxState &S, int v, double &x, double &y) {
  const int row = v / S.cols, col = v - row * S.cols;
  // ExplorationEnv.index2coor (exploration_env.py:364-367)
  x = (col + 0.5) * S.cfg.resolution + S.cfg.map_min_x;
  y = (row + 0.5) * S.cfg.resolution + S.cfg.map_min_y;
}
// ExplorationEnv.coor2index (exploration_env.py:369-372): int(round(.)) with Python's round-half-even
__device__ __forceinline__ int coor2cell(const DrlgxState &S, double x, double y) {
  int j = (int)rint((x - S.cfg.map_min_x) / S.cfg.resolution - 0.5);
  int i = (int)rint((y - S.cfg.map_min_y) / S.cfg.resolution - 0.5);
  // numpy would wrap negative indices / raise beyond the grid; nodes live inside the map in practice: clamp
  i = min(max(i, 0), S.rows - 1);
  j = min(max(j, 0), S.cols - 1);
  return i * S.cols + j;
}
__device__ __forceinline__ double dist2d(double x1, double y1, double x2, double y2) {
  return sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));  // ExplorationEnv.points2dist
}

__global__ __launch_bounds__(kT) void k_graph_build(DrlgxState S, GraphBufs G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int e = blockIdx.x, tid = threadIdx.x;
  const int *cnt = S.cnt + (size_t)e * DRLGX_CNT_STRIDE;
  const int P = cnt[C_P], L = cnt[C_L], M = cnt[C_M];
  const int V = S.V, rows = S.rows, cols = S.cols;
  int *gi = G.gi + (size_t)e * G.gi_stride;
  int *node_of_slot = gi + 4, *slot_of_node = node_of_slot + S.L_max, *lm_front = slot_of_node + S.L_max,
      *fsel = lm_front + S.L_max;
  // LDS: flags (V bytes) | per-thread counts (kT ints) | frontier cell list (V shorts) | best cell per query
  unsigned char *flag = smem_raw;
  int *tcount = reinterpret_cast<int *>(smem_raw + ((V + 15) & ~15));
  unsigned short *fcell = reinterpret_cast<unsigned short *>(tcount + kT + 4);
  int *best = reinterpret_cast<int *>(fcell + ((V + 3) & ~3));  // [L_max + 1]
  const double *prob = S.vm_prob + (size_t)e * V;
  const int *lm_key = S.lm_key + (size_t)e * S.L_max;
  const double *est_lm = S.est_lm + (size_t)e * S.L_max * 2;
  const double *est_pose = S.est_pose + (size_t)e * S.P_max * 4;
  const drlgx_config &cfg = S.cfg;
  const double ext = 20.0;  // ExplorationEnv.ext

  // node order of the landmarks: rank of the ground-truth key (gtsam keyVector is sorted, App. A.4)
  for (int j = tid; j < L; j += kT) {
    int r = 0;
    for (int k = 0; k < L; ++k) r += (lm_key[k] < lm_key[j]) ? 1 : 0;
    node_of_slot[j] = r;
    slot_of_node[r] = j;
  }
  // frontier cells (exploration_env.py:306-325): free (p < 0.45), >= 2 unknown (0.49 < p < 0.51) cells in the
  // clipped 3x3 neighbourhood, centre inside the unpadded box
  for (int v = tid; v < V; v += kT) {
    unsigned char f = 0;
    if (prob[v] < 0.45) {
      const int ci = v / cols, cj = v - ci * cols;
      const int i0 = ci - 1 >= 0 ? ci - 1 : 0, i1 = ci + 1 < rows ? ci + 1 : rows - 1;
      const int j0 = cj - 1 >= 0 ? cj - 1 : 0, j1 = cj + 1 < cols ? cj + 1 : cols - 1;
      int count = 0;
      for (int ni = i0; ni <= i1; ++ni)
        for (int nj = j0; nj <= j1; ++nj) {
          const double p = prob[ni * cols + nj];
          if (0.49 < p && p < 0.51) count++;
        }
      if (count >= 2) {
        double x, y;
        cell_xy(S, v, x, y);
        if (cfg.map_min_x + ext <= x && x <= cfg.map_max_x - ext && cfg.map_min_y + ext <= y && y <= cfg.map_max_y - ext)
          f = 1;
      }
    }
    flag[v] = f;
  }
  __syncthreads();
  // ordered compaction (np.nonzero order = ascending cell index): contiguous chunk per thread + LDS scan
  const int cpt = (V + kT - 1) / kT;
  const int v0 = tid * cpt, v1 = min(V, v0 + cpt);
  int c = 0;
  for (int v = v0; v < v1; ++v) c += flag[v];
  tcount[tid] = c;
  __syncthreads();
  for (int o = 1; o < kT; o <<= 1) {
    int add = (tid >= o) ? tcount[tid - o] : 0;
    __syncthreads();
    tcount[tid] += add;
    __syncthreads();
  }
  const int nf = tcount[kT - 1];
  {
    int pos = tcount[tid] - c;
    for (int v = v0; v < v1; ++v)
      if (flag[v]) fcell[pos++] = (unsigned short)v;
  }
  __syncthreads();
  // nearest frontier to the robot (query 0) and to every landmark in node order (exploration_env.py:350-358)
  const double rx = est_pose[4 * (P - 1)], ry = est_pose[4 * (P - 1) + 1];
  for (int q = tid; q <= L; q += kT) {
    double px = rx, py = ry;
    if (q > 0) {
      const int s = slot_of_node[q - 1];
      px = est_lm[2 * s];
      py = est_lm[2 * s + 1];
    }
    double bd = __longlong_as_double(0x7ff0000000000000LL);  // +inf
    int bc = -1;
    for (int k = 0; k < nf; ++k) {
      double x, y;
      cell_xy(S, fcell[k], x, y);
      const double d = dist2d(px, py, x, y);
      if (d < bd) {
        bd = d;
        bc = fcell[k];
      }
    }
    best[q] = bc;
  }
  __syncthreads();
  if (tid == 0) {
    int F = 0;
    if (nf > 0) {
      for (int q = 0; q <= L; ++q) {  // dedup in order of first appearance (exploration_env.py:327-338)
        int f = -1;
        for (int k = 0; k < F; ++k)
          if (fsel[k] == best[q]) {
            f = k;
            break;
          }
        if (f < 0) {
          f = F;
          fsel[F++] = best[q];
        }
        if (q > 0) lm_front[q - 1] = f;
      }
    }
    // undirected edges: factors (unique pose-landmark pairs + odometry) + robot-frontier + landmark-frontier.
    // data_process drops exact zeros (policy.py:218-219): only a frontier edge of length 0 can be one.
    int und = M + (P - 1);
    if (nf > 0) {
      double fx, fy;
      cell_xy(S, fsel[0], fx, fy);
      if (dist2d(fx, fy, rx, ry) != 0.0) und++;
      for (int a = 0; a < L; ++a) {
        cell_xy(S, fsel[lm_front[a]], fx, fy);
        const int s = slot_of_node[a];
        if (dist2d(fx, fy, est_lm[2 * s], est_lm[2 * s + 1]) != 0.0) und++;
      }
    }
    gi[0] = L + P + F;
    gi[1] = 2 * und;
    gi[2] = F;
    gi[3] = nf;
    if (nf == 0) atomicMin(S.status, DRLGX_E_INVALID);  // the reference raises when no frontier exists
  }
}

// exclusive scan of the per-env node / edge counts -> batch offsets (one 1024-thread workgroup, chunk per thread +
// shuffle scans; the serial version spent 58 us on 256 dependent global loads)
__global__ __launch_bounds__(1024) void k_graph_scan(DrlgxState S, GraphBufs G, int32_t *node_off, int32_t *edge_off) {
  __shared__ int wtot[2][16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n = S.n_envs;
  const int chunk = (n + 1023) / 1024;
  const int i0 = t * chunk, i1 = min(n, i0 + chunk);
  int sn = 0, se = 0;
  for (int i = i0; i < i1; ++i) {
    sn += G.gi[(size_t)i * G.gi_stride + 0];
    se += G.gi[(size_t)i * G.gi_stride + 1];
  }
  int xn = sn, xe = se;
  for (int off = 1; off < 64; off <<= 1) {
    const int yn = __shfl_up(xn, off), ye = __shfl_up(xe, off);
    if (lane >= off) {
      xn += yn;
      xe += ye;
    }
  }
  if (lane == 63) {
    wtot[0][wave] = xn;
    wtot[1][wave] = xe;
  }
  __syncthreads();
  if (wave == 0) {
    int vn = lane < 16 ? wtot[0][lane] : 0, ve = lane < 16 ? wtot[1][lane] : 0;
    const int in_ = vn, ie = ve;
    for (int off = 1; off < 16; off <<= 1) {
      const int yn = __shfl_up(vn, off), ye = __shfl_up(ve, off);
      if (lane >= off) {
        vn += yn;
        ve += ye;
      }
    }
    if (lane < 16) {
      wtot[0][lane] = vn - in_;
      wtot[1][lane] = ve - ie;
    }
    if (lane == 15) {
      node_off[n] = vn;
      edge_off[n] = ve;
    }
  }
  __syncthreads();
  sn = wtot[0][wave] + xn - sn;
  se = wtot[1][wave] + xe - se;
  for (int i = i0; i < i1; ++i) {
    node_off[i] = sn;
    edge_off[i] = se;
    sn += G.gi[(size_t)i * G.gi_stride + 0];
    se += G.gi[(size_t)i * G.gi_stride + 1];
  }
}

__global__ __launch_bounds__(kT) void k_graph_emit(DrlgxState S, GraphBufs G, const int32_t *node_off, const int32_t *edge_off,
                                                   float *x, int64_t *edge_index, float *edge_attr, int32_t *n_frontier,
                                                   double *frontier_xy, int32_t *nearest_node, int max_frontier) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int e = blockIdx.x, tid = threadIdx.x;
  const int *cnt = S.cnt + (size_t)e * DRLGX_CNT_STRIDE;
  const int P = cnt[C_P], L = cnt[C_L], M = cnt[C_M];
  const int V = S.V;
  const int *gi = G.gi + (size_t)e * G.gi_stride;
  const int N = gi[0], F = gi[2];
  const int *node_of_slot = gi + 4, *slot_of_node = node_of_slot + S.L_max, *lm_front = slot_of_node + S.L_max,
            *fsel = lm_front + S.L_max;
  const int n0 = node_off[e], e0 = edge_off[e], e_total = edge_off[S.n_envs];
  const double *prob = S.vm_prob + (size_t)e * V;
  const double *vtr = S.vm_tr + (size_t)e * V;
  const double *est_lm = S.est_lm + (size_t)e * S.L_max * 2;
  const double *est_pose = S.est_pose + (size_t)e * S.P_max * 4;
  const double *lm_tr = S.lm_tr + (size_t)e * S.L_max;
  const double *pose_tr = S.pose_tr + (size_t)e * S.P_max;
  const int *meas_pose = S.meas_pose + (size_t)e * S.M_max;
  const int *meas_lm = S.meas_lm + (size_t)e * S.M_max;
  const double *meas_br = S.meas_br + (size_t)e * S.M_max * 2;
  const double *odo = S.odo + (size_t)e * S.P_max * 4;
  // LDS: obs table [L][P] (factor index + 1), row start offsets [N + 1]
  unsigned short *obs = reinterpret_cast<unsigned short *>(smem_raw);
  int *rstart = reinterpret_cast<int *>(smem_raw + (((size_t)L * P * 2 + 15) & ~(size_t)15));
  for (int k = tid; k < L * P; k += kT) obs[k] = 0;
  __syncthreads();
  for (int m = tid; m < M; m += kT) obs[meas_lm[m] * P + meas_pose[m]] = (unsigned short)(m + 1);
  __syncthreads();
  const double rx = est_pose[4 * (P - 1)], ry = est_pose[4 * (P - 1) + 1];
  double root_theta = atan2(est_pose[4 * (P - 1) + 3], est_pose[4 * (P - 1) + 2]);
  const double PI = 3.14159265358979323846;
  if (root_theta < 0) root_theta = PI * 2 + root_theta;
  // ---- node features (exploration_env.py:236-276) ----
  for (int n = tid; n < N; n += kT) {
    double px, py, f0;
    float type;
    if (n < L) {
      const int s = slot_of_node[n];
      px = est_lm[2 * s]; py = est_lm[2 * s + 1];
      f0 = lm_tr[s];
      type = -1.0f;
    } else if (n < L + P) {
      const int p = n - L;
      px = est_pose[4 * p]; py = est_pose[4 * p + 1];
      // features_matrix_ is only written for keys that appear in a two-key factor (SLAM2D.cpp:226-268)
      f0 = (P > 1 || M > 0) ? pose_tr[p] : 0.0;
      if (P == 1 && M > 0) {
        bool any = false;
        for (int m = 0; m < M; ++m) any |= (meas_pose[m] == p);
        f0 = any ? pose_tr[p] : 0.0;
      }
      type = (p == P - 1) ? 0.0f : -1.0f;
    } else {
      double fx, fy;
      cell_xy(S, fsel[n - L - P], fx, fy);
      px = fx; py = fy;
      f0 = vtr[coor2cell(S, fx, fy)];
      type = 1.0f;
    }
    const double dist = dist2d(px, py, rx, ry);
    // ExplorationEnv.diff_theta (exploration_env.py:378-387)
    double goal_theta = atan2(py - ry, px - rx);
    if (goal_theta < 0) goal_theta = PI * 2 + goal_theta;
    double diff = goal_theta - root_theta;
    if (diff < 0) diff = PI * 2 + diff;
    const double pr = prob[coor2cell(S, px, py)];
    float *o = x + (size_t)(n0 + n) * 5;
    o[0] = (float)f0; o[1] = (float)dist; o[2] = (float)diff; o[3] = (float)pr; o[4] = type;
  }
  // ---- edges: per-row counts of higher neighbours, scan, emit ----
  for (int n = tid; n <= N; n += kT) rstart[n] = 0;
  __syncthreads();
  for (int n = tid; n < N; n += kT) {
    int c = 0;
    if (n < L) {
      const int s = slot_of_node[n];
      for (int p = 0; p < P; ++p) c += obs[s * P + p] ? 1 : 0;
      if (F > 0) {
        double fx, fy;
        cell_xy(S, fsel[lm_front[n]], fx, fy);
        if (dist2d(fx, fy, est_lm[2 * s], est_lm[2 * s + 1]) != 0.0) c++;
      }
    } else if (n < L + P) {
      const int p = n - L;
      if (p + 1 < P) c++;
      if (p == P - 1 && F > 0) {
        double fx, fy;
        cell_xy(S, fsel[0], fx, fy);
        if (dist2d(fx, fy, rx, ry) != 0.0) c++;
      }
    }
    rstart[n + 1] = c;
  }
  __syncthreads();
  if (tid == 0)
    for (int n = 0; n < N; ++n) rstart[n + 1] += rstart[n];
  __syncthreads();
  auto emit = [&](int k, int i, int j, double w) {
    const size_t a = (size_t)e0 + 2 * (size_t)k;
    edge_index[a] = n0 + i;
    edge_index[a + 1] = n0 + j;
    edge_index[(size_t)e_total + a] = n0 + j;
    edge_index[(size_t)e_total + a + 1] = n0 + i;
    edge_attr[a] = (float)w;
    edge_attr[a + 1] = (float)w;
  };
  for (int n = tid; n < N; n += kT) {
    int k = rstart[n];
    if (n < L) {
      const int s = slot_of_node[n];
      for (int p = 0; p < P; ++p) {
        const int m1 = obs[s * P + p];
        if (m1) emit(k++, n, L + p, meas_br[2 * (m1 - 1) + 1]);  // measured range (SLAM2D.cpp:253-254)
      }
      if (F > 0) {
        double fx, fy;
        cell_xy(S, fsel[lm_front[n]], fx, fy);
        const double d = dist2d(fx, fy, est_lm[2 * s], est_lm[2 * s + 1]);
        if (d != 0.0) emit(k++, n, L + P + lm_front[n], d);
      }
    } else if (n < L + P) {
      const int p = n - L;
      if (p + 1 < P)  // odometry length + 0.001 (SLAM2D.cpp:236-239)
        emit(k++, n, n + 1, sqrt(pow(odo[4 * p], 2) + pow(odo[4 * p + 1], 2)) + 0.001);
      if (p == P - 1 && F > 0) {
        double fx, fy;
        cell_xy(S, fsel[0], fx, fy);
        const double d = dist2d(fx, fy, rx, ry);
        if (d != 0.0) emit(k++, n, L + P, d);
      }
    }
  }
  if (tid == 0) {
    n_frontier[e] = F;
    nearest_node[e] = L + P;  // frontier 0 is the robot's nearest frontier (exploration_env.py:216-218)
  }
  for (int f = tid; f < F && f < max_frontier; f += kT) {
    double fx, fy;
    cell_xy(S, fsel[f], fx, fy);
    frontier_xy[((size_t)e * max_frontier + f) * 2] = fx;
    frontier_xy[((size_t)e * max_frontier + f) * 2 + 1] = fy;
  }
}

}  // namespace

void drlgx_launch_graph(const DrlgxState &S, hipStream_t st, int *gi, int gi_stride, int32_t *node_off, int32_t *edge_off,
                        float *x, int64_t *edge_index, float *edge_attr, int32_t *n_frontier, double *frontier_xy,
                        int32_t *nearest_node, int max_frontier) {
  GraphBufs G{gi, gi_stride};
  const size_t lds_a = ((S.V + 15) & ~15) + (size_t)(kT + 4) * 4 + (size_t)((S.V + 3) & ~3) * 2 + (size_t)(S.L_max + 2) * 4;
  const size_t lds_c = (((size_t)S.L_max * S.P_max * 2 + 15) & ~(size_t)15) + (size_t)(2 * S.L_max + S.P_max + 4) * 4;
  static bool attr_set[32] = {false};
  const void *fns[] = {reinterpret_cast<const void *>(&k_graph_build), reinterpret_cast<const void *>(&k_graph_emit)};
  drlgx_ensure_lds_attr(attr_set, fns, 2, 160 * 1024);
  hipLaunchKernelGGL(k_graph_build, dim3(S.n_envs), dim3(kT), lds_a, st, S, G);
  hipLaunchKernelGGL(k_graph_scan, dim3(1), dim3(1024), 0, st, S, G, node_off, edge_off);
  hipLaunchKernelGGL(k_graph_emit, dim3(S.n_envs), dim3(kT), lds_c, st, S, G, node_off, edge_off, x, edge_index, edge_attr,
                     n_frontier, frontier_xy, nearest_node, max_frontier);
}
