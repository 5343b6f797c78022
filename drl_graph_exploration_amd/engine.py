"""Thin object wrapper over the C ABI (include/drlgx.h): one `Engine` = one drlgx_engine handle.

Bulk inputs/outputs are torch CUDA tensors (torch is used only for device memory and streams); the
`*_host` getters return numpy arrays and mirror the getters of the reference's pybind modules.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import DrlgxConfig, start_pose


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_NP_OF = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8,
          torch.bool: np.bool_, torch.int16: np.int16, torch.int8: np.int8}


class Engine(object):
    def __init__(self, cfg, n_envs, n_rollouts=0, device=0):
        if not isinstance(cfg, DrlgxConfig):
            raise TypeError("cfg must be a DrlgxConfig")
        if not torch.cuda.is_available():
            raise _lib.DrlgxError("no HIP device visible: the drlgx engine has no CPU fallback")
        self.L = _lib.lib()
        self.cfg = cfg
        self.n_envs = n_envs
        self.n_rollouts = n_rollouts
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p()
        _lib.check(self.L.drlgx_create(C.byref(cfg), n_envs, n_rollouts, device, C.byref(self.h)))
        r, c = C.c_int32(), C.c_int32()
        self.L.drlgx_vm_shape(self.h, C.byref(r), C.byref(c))
        self.rows, self.cols = r.value, c.value
        self.use_torch_stream()

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.drlgx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        _lib.check(rc, self.h)

    def use_torch_stream(self):
        """Enqueue engine kernels on torch's current stream of the engine's device, so that tensors can be shared
        without syncs.  Called before every call that enqueues work (cheap when the stream has not changed), so the
        engine follows `with torch.cuda.stream(...)` blocks."""
        s = _lib.stream_ptr(self.device)
        if s != getattr(self, "_bound_stream", None):
            self._chk(self.L.drlgx_set_stream(self.h, C.c_void_p(s)))
            self._bound_stream = s

    def synchronize(self):
        self._chk(self.L.drlgx_synchronize(self.h))

    def status(self):
        return self.L.drlgx_status_host(self.h)

    def check_status(self):
        st = self.status()
        if st != 0:
            _lib.check(st, self.h)

    def fetch(self, *tensors):
        """check_status() with the given device tensors brought to the host under the same synchronisation (one stream drain and
        one copy instead of a `.cpu()` / `.item()` each): returns one numpy array per tensor, same dtype and shape."""
        self.use_torch_stream()
        for t in tensors:
            if t.device != self.device:
                raise ValueError("Engine.fetch reads tensors on the engine's device (%s), not %s" % (self.device, t.device))
        parts = [t.detach().contiguous().view(-1).view(torch.uint8) for t in tensors]
        n = sum(p.numel() for p in parts)
        if n == 0:
            self.check_status()
            return [np.empty(tuple(t.shape), dtype=_NP_OF[t.dtype]) for t in tensors]
        packed = parts[0] if len(parts) == 1 else torch.cat(parts)
        host = getattr(self, "_fetch_host", None)
        if host is None or host.numel() < n:
            host = self._fetch_host = torch.empty(max(2 * n, 1 << 16), dtype=torch.uint8, pin_memory=True)
        st = self.L.drlgx_status_fetch_host(self.h, _p(packed), n, C.c_void_p(host.data_ptr()))
        if st != 0:
            _lib.check(st, self.h)
        raw = host.numpy()
        out, off = [], 0
        for t, p in zip(tensors, parts):
            k = p.numel()
            out.append(raw[off:off + k].copy().view(_NP_OF[t.dtype]).reshape(tuple(t.shape)))
            off += k
        return out

    # ---- life cycle
    def reset(self, env_ids, seeds, starts=None, los=None):
        """SS2D.__init__ for the listed envs. starts: (n,3) x,y,theta; or `los` to use the reference's
        legacy numpy start-pose stream (pyss2d.py:89-95)."""
        self.use_torch_stream()
        env_ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        if starts is None:
            starts = np.array([start_pose(int(lo), self.cfg.map_max_x) for lo in los], dtype=np.float64)
        starts = np.ascontiguousarray(starts, dtype=np.float64)
        self._chk(self.L.drlgx_reset_host(self.h, len(env_ids), env_ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                          seeds.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          starts.ctypes.data_as(C.POINTER(C.c_double))))

    def step(self, odom, active=None):
        """SS2D.simulate(core=True) for all envs. odom: CUDA float64 [n_envs,3]; active: CUDA uint8 [n_envs]."""
        self.use_torch_stream()
        assert odom.is_cuda and odom.dtype == torch.float64 and odom.is_contiguous()
        self._chk(self.L.drlgx_step(self.h, _p(odom), _p(active)))

    # ---- staged belief step (one call per call of the reference's SS2D.__init__ / SS2D.simulate)

    def step_plan(self, actions, n_actions, action_index, map_last_only=True):
        """One action index of every env's plan (drlgx_step_plan): actions [n_envs, max_actions, 3] f64, n_actions [n_envs]
        i32 on the device; with `map_last_only` the virtual map is rebuilt at each env's last action only."""
        self.use_torch_stream()
        self._chk(self.L.drlgx_step_plan(self.h, _p(actions), _p(n_actions), int(action_index), 1 if map_last_only else 0))

    def step_plans(self, actions, n_actions, max_n_actions, map_last_only=True):
        """Every env's whole plan (drlgx_step_plans): `for k in range(max_n_actions): step_plan(..., k)` in one call."""
        self.use_torch_stream()
        self._chk(self.L.drlgx_step_plans(self.h, _p(actions), _p(n_actions), int(max_n_actions), 1 if map_last_only else 0))

    def stage_reset(self, env_ids, seeds, starts):
        env_ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        starts = np.ascontiguousarray(starts, dtype=np.float64)
        self.use_torch_stream()
        self._chk(self.L.drlgx_stage_reset_host(self.h, len(env_ids), env_ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                                seeds.ctypes.data_as(C.POINTER(C.c_uint32)), starts.ctypes.data_as(C.POINTER(C.c_double))))

    def stage_set_prior_information(self, env, information):
        """A full 3 x 3 prior information for one env (after stage_reset, before the first optimise)."""
        info = np.ascontiguousarray(information, dtype=np.float64).reshape(9)
        self._chk(self.L.drlgx_stage_set_prior_information_host(self.h, int(env), info.ctypes.data_as(C.POINTER(C.c_double))))

    def stage_set_prior_pose(self, env, xytheta):
        """The pose of the prior factor / the initial estimate of x0 for one env (after stage_reset, before the first measurement)."""
        p = np.ascontiguousarray(xytheta, dtype=np.float64).reshape(3)
        self._chk(self.L.drlgx_stage_set_prior_pose_host(self.h, int(env), p.ctypes.data_as(C.POINTER(C.c_double))))

    def stage_move(self, odom, active=None):
        self.use_torch_stream()
        self._chk(self.L.drlgx_stage_move(self.h, _p(odom), _p(active)))

    def stage_measure(self, active=None):
        """One Simulator2D::measure per env: (keys [n, LG] i32, bearing_range [n, LG, 2] f64, count [n] i32)."""
        self.use_torch_stream()
        lg = max(self.cfg.num_landmarks, 1)
        keys = torch.zeros(self.n_envs, lg, dtype=torch.int32, device=self.device)
        br = torch.zeros(self.n_envs, lg, 2, dtype=torch.float64, device=self.device)
        cnt = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self._chk(self.L.drlgx_stage_measure(self.h, _p(active), _p(keys), _p(br), _p(cnt)))
        return keys, br, cnt

    def stage_add_measurements(self, keys, br, cnt, active=None):
        self.use_torch_stream()
        self._chk(self.L.drlgx_stage_add_measurements(self.h, _p(active), _p(keys), _p(br), _p(cnt)))

    def stage_optimize(self, active=None):
        self.use_torch_stream()
        self._chk(self.L.drlgx_stage_optimize(self.h, _p(active)))

    def stage_update_map(self, active=None, rebuild=True):
        self.use_torch_stream()
        self._chk(self.L.drlgx_stage_update_map(self.h, _p(active), int(bool(rebuild))))

    def set_fixed_landmarks(self, xy):
        """The listed ground-truth landmarks of `Simulator2D.random_landmarks(landmarks, num, params)` (keys 0 .. k - 1 of every
        env; cfg.num_landmarks is the total): effective at the next reset."""
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        self._chk(self.L.drlgx_set_fixed_landmarks_host(self.h, len(xy), xy.ctypes.data_as(C.POINTER(C.c_double))))

    def set_planner_parameter(self, angle_weight, distance_weight0, distance_weight1, occupancy_threshold, max_edge_length, algorithm):
        self._chk(self.L.drlgx_set_planner_parameter(self.h, float(angle_weight), float(distance_weight0), float(distance_weight1),
                                                     float(occupancy_threshold), float(max_edge_length), int(algorithm)))
        for k, v in (("angle_weight", angle_weight), ("distance_weight0", distance_weight0), ("distance_weight1", distance_weight1),
                     ("occupancy_threshold", occupancy_threshold), ("max_edge_length", max_edge_length), ("algorithm", int(algorithm))):
            setattr(self.cfg, k, v)

    def fm2_update(self, cand_env, actions, n_actions):
        """FastMarginals2-style covariance look-ahead (drlgx_fm2_update): returns (cov [C, max_poses + max_actions, 3, 3]
        float64, n_poses [C] int32); rows beyond n_poses[c] are undefined."""
        self.use_torch_stream()
        n = cand_env.numel()
        stride = self.cfg.max_poses + self.cfg.max_actions
        cov = torch.zeros(n, stride, 3, 3, dtype=torch.float64, device=self.device)
        n_out = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._chk(self.L.drlgx_fm2_update(self.h, n, _p(cand_env), _p(actions), _p(n_actions), _p(cov), stride, _p(n_out)))
        return cov, n_out

    def utility(self, dist=None):
        self.use_torch_stream()
        out = torch.empty(self.n_envs, dtype=torch.float64, device=self.device)
        self._chk(self.L.drlgx_utility(self.h, _p(dist), _p(out)))
        return out

    def uncertainty_em(self, algorithm):
        self.use_torch_stream()
        out = torch.empty(self.n_envs, dtype=torch.float64, device=self.device)
        self._chk(self.L.drlgx_uncertainty_em(self.h, int(algorithm), _p(out)))
        return out

    def explored(self):
        self.use_torch_stream()
        out = torch.empty(self.n_envs, dtype=torch.float64, device=self.device)
        self._chk(self.L.drlgx_explored(self.h, _p(out)))
        return out

    def metrics(self, sigma0=1.0):
        """[n_envs, 3] float64 CUDA tensor: landmark error, map entropy, max localisation uncertainty - the columns
        scripts/test.py:136-142 writes per executed action."""
        self.use_torch_stream()
        out = torch.empty(self.n_envs, 3, dtype=torch.float64, device=self.device)
        self._chk(self.L.drlgx_metrics(self.h, float(sigma0), _p(out)))
        return out

    def cov_array(self):
        """VirtualMap.to_cov_array for every env: (length, angle) [n_envs, rows, cols] float64 CUDA tensors."""
        self.use_torch_stream()
        ln = torch.empty(self.n_envs, self.rows, self.cols, dtype=torch.float64, device=self.device)
        an = torch.empty_like(ln)
        self._chk(self.L.drlgx_cov_array(self.h, _p(ln), _p(an)))
        return ln, an

    def line_plan(self, cand_env, goals):
        self.use_torch_stream()
        n = cand_env.numel()
        A = self.cfg.max_actions
        actions = torch.zeros(n, A, 3, dtype=torch.float64, device=self.device)
        n_act = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._chk(self.L.drlgx_line_plan(self.h, n, _p(cand_env), _p(goals), _p(actions), _p(n_act)))
        return actions, n_act

    def lookahead(self, cand_env, actions, n_actions, max_n_actions=None):
        """Look-ahead rewards; `max_n_actions` (host int >= every n_actions[i]) skips the launches of the action indices
        no plan reaches (None: all cfg.max_actions indices are launched)."""
        self.use_torch_stream()
        n = cand_env.numel()
        rewards = torch.empty(n, dtype=torch.float64, device=self.device)
        if max_n_actions is None:
            self._chk(self.L.drlgx_lookahead(self.h, n, _p(cand_env), _p(actions), _p(n_actions), _p(rewards)))
        else:
            self._chk(self.L.drlgx_lookahead_bounded(self.h, n, _p(cand_env), _p(actions), _p(n_actions), int(max_n_actions),
                                                     _p(rewards)))
        return rewards

    def graph_capacity(self):
        """(max nodes, max edges, max frontiers per env) of one batched export."""
        if not hasattr(self, "_gcap"):
            a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
            self._chk(self.L.drlgx_graph_capacity(self.h, C.byref(a), C.byref(b), C.byref(c)))
            self._gcap = (a.value, b.value, c.value)
        return self._gcap

    def graph(self, plan=False):
        """Batched ExplorationEnv.graph_matrix + DeepQ.data_process for all envs (one PyG-style batch).
        Returns a dict of CUDA tensors: x [N,5] f32, edge_index [2,E] i64, edge_attr [E] f32, node_off / edge_off
        [n_envs+1] i32, batch [N] i64, n_frontier [n_envs] i32, frontier_xy [n_envs,Fmax,2] f64,
        nearest_frontier_node [n_envs] i32 (local node id); max_graph_edges = the largest graph's edge count (host int);
        node_off_h / edge_off_h / n_frontier_h = host copies (numpy).  Raises on a non-zero status word (check_status).
        plan=True: the line plan to EVERY frontier slot rides along (drlgx_line_plan over all n_envs x Fmax slots, the unused ones
        planned towards the origin and ignored): actions_pad [n_envs, Fmax, max_actions, 3] f64, n_act_pad [n_envs, Fmax] i32 and
        its host copy n_act_pad_h - the plans' lengths reach the host in the export's own synchronisation."""
        self.use_torch_stream()
        if not hasattr(self, "_gcap"):
            a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
            self._chk(self.L.drlgx_graph_capacity(self.h, C.byref(a), C.byref(b), C.byref(c)))
            self._gcap = (a.value, b.value, c.value)
        mn, me, mf = self._gcap
        dev = self.device
        node_off = torch.empty(self.n_envs + 1, dtype=torch.int32, device=dev)
        edge_off = torch.empty(self.n_envs + 1, dtype=torch.int32, device=dev)
        x = torch.empty(mn, 5, dtype=torch.float32, device=dev)
        ei = torch.empty(2 * me, dtype=torch.int64, device=dev)
        ea = torch.empty(me, dtype=torch.float32, device=dev)
        nfr = torch.empty(self.n_envs, dtype=torch.int32, device=dev)
        fxy = torch.zeros(self.n_envs, mf, 2, dtype=torch.float64, device=dev)
        near = torch.empty(self.n_envs, dtype=torch.int32, device=dev)
        self._chk(self.L.drlgx_graph(self.h, _p(node_off), _p(edge_off), _p(x), _p(ei), _p(ea), _p(nfr), _p(fxy), _p(near)))
        extra = {}
        if plan:
            if getattr(self, "_pad_env", None) is None:
                self._pad_env = torch.arange(self.n_envs, device=dev, dtype=torch.int32).repeat_interleave(mf).contiguous()
            acts, n_act = self.line_plan(self._pad_env, fxy.view(-1, 2))
            extra = dict(actions_pad=acts.view(self.n_envs, mf, -1, 3), n_act_pad=n_act.view(self.n_envs, mf))
        # the batch's boundaries and frontier counts on the host too, with the status word, in ONE synchronisation (keys *_h)
        if plan:
            node_off_h, edge_off_h, nfr_h, n_act_h = self.fetch(node_off, edge_off, nfr, extra["n_act_pad"])
            extra["n_act_pad_h"] = n_act_h
        else:
            node_off_h, edge_off_h, nfr_h = self.fetch(node_off, edge_off, nfr)
        N, E = int(node_off_h[-1]), int(edge_off_h[-1])
        counts = (node_off[1:] - node_off[:-1]).to(torch.int64)
        batch = torch.repeat_interleave(torch.arange(self.n_envs, device=dev), counts, output_size=N)
        return dict(x=x[:N], edge_index=ei[:2 * E].view(2, E), edge_attr=ea[:E], node_off=node_off, edge_off=edge_off,
                    batch=batch, n_frontier=nfr, frontier_xy=fxy, nearest_frontier_node=near,
                    max_graph_edges=int(np.diff(edge_off_h).max()) if self.n_envs else 0,
                    node_off_h=node_off_h, edge_off_h=edge_off_h, n_frontier_h=nfr_h, **extra)

    def snapshot(self, slot=0):
        self.use_torch_stream()
        self._chk(self.L.drlgx_snapshot(self.h, slot))

    def restore(self, slot=0):
        self.use_torch_stream()
        self._chk(self.L.drlgx_restore(self.h, slot))

    def inc_stats(self, reset=False):
        """(SLAM updates served by the incremental rank-k path, full solves) since creation / the last reset;
        (-1, -1) when the incremental path is disabled (DRLGX_INCREMENTAL=0 at creation)."""
        out = (C.c_int64 * 2)()
        self._chk(self.L.drlgx_inc_stats_host(self.h, out, 1 if reset else 0))
        return int(out[0]), int(out[1])

    def timing_enable(self, on=True):
        """on: False / True (spans around what is launched, fused step = 'step') / 2 (per-stage kernels)."""
        self._chk(self.L.drlgx_timing_enable(self.h, int(on)))

    def timing_read(self):
        ms = (C.c_double * _lib.N_TIMERS)()
        n = (C.c_int64 * _lib.N_TIMERS)()
        self._chk(self.L.drlgx_timing_read_host(self.h, ms, n))
        names = ["sim", "slam", "map", "copy", "graph", "step", "t6", "t7"]
        return {names[i]: (ms[i], n[i]) for i in range(_lib.N_TIMERS)}

    # ---- getters (host)
    def counts(self, inst):
        out = (C.c_int32 * 5)()
        self._chk(self.L.drlgx_get_counts_host(self.h, inst, out))
        return dict(poses=out[0], landmarks=out[1], factors=out[2], step=out[3], isam_count=out[4])

    def counts_dev(self):
        """[n_envs, 5] int32 CUDA tensor: poses, landmarks, factors, step, isam update count (no sync)."""
        self.use_torch_stream()
        out = torch.empty(self.n_envs, 5, dtype=torch.int32, device=self.device)
        self._chk(self.L.drlgx_counts(self.h, _p(out)))
        return out

    def poses(self, inst):
        P = self.counts(inst)["poses"]
        xyt = np.zeros((P, 3))
        info = np.zeros((P, 3, 3))
        dp = C.POINTER(C.c_double)
        self._chk(self.L.drlgx_get_poses_host(self.h, inst, xyt.ctypes.data_as(dp), info.ctypes.data_as(dp)))
        return xyt, info

    def landmarks(self, inst):
        n = self.counts(inst)["landmarks"]
        keys = np.zeros(n, dtype=np.int32)
        xy = np.zeros((n, 2))
        info = np.zeros((n, 2, 2))
        dp = C.POINTER(C.c_double)
        self._chk(self.L.drlgx_get_landmarks_host(self.h, inst, keys.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  xy.ctypes.data_as(dp), info.ctypes.data_as(dp)))
        return keys, xy, info

    def cov_traces(self, inst):
        c = self.counts(inst)
        lm = np.zeros(c["landmarks"])
        ps = np.zeros(c["poses"])
        dp = C.POINTER(C.c_double)
        self._chk(self.L.drlgx_get_cov_traces_host(self.h, inst, lm.ctypes.data_as(dp), ps.ctypes.data_as(dp)))
        return lm, ps

    def virtual_map(self, inst):
        V = self.rows * self.cols
        prob = np.zeros(V)
        info = np.zeros((V, 2, 2))
        tr = np.zeros(V)
        upd = np.zeros(V, dtype=np.uint8)
        dp = C.POINTER(C.c_double)
        self._chk(self.L.drlgx_get_virtual_map_host(self.h, inst, prob.ctypes.data_as(dp), info.ctypes.data_as(dp),
                                                    tr.ctypes.data_as(dp), upd.ctypes.data_as(C.POINTER(C.c_uint8))))
        return prob.reshape(self.rows, self.cols), info, tr.reshape(self.rows, self.cols), upd

    def ground_truth(self, inst):
        veh = np.zeros(3)
        lms = np.zeros((max(self.cfg.num_landmarks, 1), 2))
        dp = C.POINTER(C.c_double)
        self._chk(self.L.drlgx_get_ground_truth_host(self.h, inst, veh.ctypes.data_as(dp), lms.ctypes.data_as(dp)))
        return veh, lms[:self.cfg.num_landmarks]

    def adjacency(self, inst):
        c = self.counts(inst)
        N = c["poses"] + c["landmarks"]
        A = np.zeros((N, N))
        X = np.zeros(N)
        dp = C.POINTER(C.c_double)
        self._chk(self.L.drlgx_get_adjacency_host(self.h, inst, A.ctypes.data_as(dp), X.ctypes.data_as(dp)))
        return A, X

    def factors(self, inst):
        m = self.counts(inst)["factors"]
        pose = np.zeros(m, dtype=np.int32)
        key = np.zeros(m, dtype=np.int32)
        b = np.zeros(m)
        r = np.zeros(m)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        self._chk(self.L.drlgx_get_factors_host(self.h, inst, pose.ctypes.data_as(ip), key.ctypes.data_as(ip),
                                                b.ctypes.data_as(dp), r.ctypes.data_as(dp)))
        return pose, key, b, r

    def landmark_order(self):
        o = np.zeros(max(self.cfg.num_landmarks, 1), dtype=np.int32)
        self._chk(self.L.drlgx_get_landmark_order_host(self.h, o.ctypes.data_as(C.POINTER(C.c_int32))))
        return o[:self.cfg.num_landmarks]
