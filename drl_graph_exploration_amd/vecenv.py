"""`VecExplorationEnv` — the batched counterpart of the reference's `ExplorationEnv`
(scripts/envs/exploration_env.py:20-422): `n_envs` independent exploration environments stepped in lock-step on one
GPU through the drlgx engine (one workgroup per environment per kernel).

Method names and meaning follow the reference class; everything that was a scalar / list per environment becomes a
device tensor over environments (or over (environment, frontier) candidates):

    reset()              ExplorationEnv.reset      (:389-422)  SS2D.__init__, 4 x (1, 1, pi/2), regenerate (+50) if no landmark
    graph_matrix()       frontier + graph_matrix + DeepQ.data_process (:196-348, policy.py:211-232) -> one PyG-style batch
    actions_all_goals()  (:134-143)  line plan to every frontier of every env
    rewards_all_goals()  (:145-162)  look-ahead reward of every plan + the per-decision np.interp normalisation / loop_clo
    step(...)            (:98-105)   executes one chosen plan per env (ragged lengths -> active masks)
    status() / done()    (:164-168, :107-110)

There is no CPU path here: the engine raises if no HIP device is visible.
"""
import math

import numpy as np
import torch

from .config import default_config, start_pose
from .engine import Engine


def normalise_rewards(raw, cand_env, cand_first, n_envs, n_frontier=None):
    """exploration_env.py:151-161 for every env at once. raw [C] f64 look-ahead rewards of the (env, frontier)
    candidates in env-major order, cand_env [C] i64, cand_first [n_envs] i64 (index of each env's frontier 0 = the
    vehicle's nearest frontier). Returns (normalised [C], loop_clo [n_envs] bool): the nearest frontier is the
    (first) arg-max -> np.interp to [-1, 0], loop_clo False; otherwise [-1, 1], loop_clo True.
    Device tensors go through drlgx_normalise_rewards (one wave per env); the tensor-op form below is the host mirror the
    CPU tests check against np.interp."""
    if raw.is_cuda:
        import ctypes as C
        from . import _lib
        if n_frontier is None:
            n_frontier = torch.diff(torch.cat([cand_first, torch.tensor([raw.numel()], device=raw.device)])).to(torch.int32)
        raw = raw.contiguous()
        out = torch.empty_like(raw)
        loop = torch.empty(n_envs, dtype=torch.uint8, device=raw.device)
        vp = C.c_void_p
        _lib.check(_lib.lib().drlgx_normalise_rewards(vp(_lib.stream_ptr(raw.device)), n_envs, vp(raw.data_ptr()),
                                                      vp(cand_first.contiguous().data_ptr()), vp(n_frontier.contiguous().data_ptr()),
                                                      vp(out.data_ptr()), vp(loop.data_ptr())))
        return out, loop.bool()
    e = cand_env
    lo = torch.full((n_envs,), float("inf"), dtype=raw.dtype, device=raw.device).scatter_reduce(0, e, raw, reduce="amin")
    hi = torch.full((n_envs,), -float("inf"), dtype=raw.dtype, device=raw.device).scatter_reduce(0, e, raw, reduce="amax")
    nearest = raw[cand_first.clamp(max=max(raw.numel() - 1, 0))]
    loop_clo = nearest < hi  # np.nanargmax returns the first maximum
    top = torch.where(loop_clo, 1.0, 0.0).to(raw.dtype)
    span = hi - lo
    slope = (top + 1.0) / torch.where(span > 0, span, torch.ones_like(span))
    r = slope[e] * (raw - lo[e]) - 1.0  # np.interp: slope * (x - xp[0]) + fp[0]
    r = torch.where(raw >= hi[e], top[e], r)  # x >= xp[-1] -> fp[-1] (also the degenerate lo == hi case)
    return r, loop_clo


class VecExplorationEnv(object):
    def __init__(self, map_size, n_envs, env_index=0, test=True, num_landmarks=None, algorithm=0, device=0,
                 n_rollouts=None, max_poses=None, starts=None, seed=None):
        self.map_size = map_size
        self.n_envs = n_envs
        self.test = test
        if max_poses is None:
            # the reference's own 40 m evaluation episodes take 96-197 primitive steps (data/test_result/40_DQN_GCN.csv)
            # on top of the 5 poses of reset().  The capacity costs memory only: the engine picks the SLAM kernel per
            # launch from the trajectories' current length (fused LDS-resident step up to 42 poses, the pose-chain
            # ordered solve of csrc/k_slam_arrow.hip beyond), refreshed at every status check
            max_poses = 256
        self.cfg = default_config(map_size, num_landmarks=num_landmarks, algorithm=algorithm, max_poses=max_poses)
        if n_rollouts is None:
            # one rollout instance per (env, frontier) candidate up to 4096; more candidates run in waves
            n_rollouts = min(n_envs * (self.cfg.max_landmarks + 1), 4096)
        self.engine = Engine(self.cfg, n_envs, n_rollouts, device)
        self.device = self.engine.device
        self.np_random = np.random.RandomState(seed)
        self.env_index = np.arange(n_envs, dtype=np.int64) + int(env_index)  # one TEST seed per env
        self._starts = None if starts is None else np.asarray(starts, dtype=np.float64).reshape(n_envs, 3)
        self._max_steps = 5000
        self.dist = torch.zeros(n_envs, dtype=torch.float64, device=self.device)
        self.loop_clo = torch.zeros(n_envs, dtype=torch.bool, device=self.device)
        self._graph = None
        self._cand_env = None
        self._scan = torch.tensor([(1.0, 1.0, math.pi / 2.0)] * n_envs, dtype=torch.float64, device=self.device)
        self.reset()

    def close(self):
        self.engine.close()

    # ------------------------------------------------------------------ reset (exploration_env.py:389-422)
    def reset(self, ids=None):
        """Re-create the listed environments (all by default)."""
        n = self.n_envs
        todo = np.arange(n, dtype=np.int64) if ids is None else np.asarray(ids, dtype=np.int64).reshape(-1)
        self.dist[torch.as_tensor(todo, device=self.device)] = 0.0
        while len(todo):
            if self.test:
                seeds = self.env_index[todo].astype(np.uint32)
                los = self.env_index[todo]
            else:
                hi = np.iinfo(np.int32).max
                seeds = np.array([self.np_random.randint(0, hi) for _ in todo], dtype=np.uint32)
                los = np.array([self.np_random.randint(0, hi) for _ in todo], dtype=np.int64)
            if self._starts is not None:
                starts = self._starts[todo]
            else:
                starts = np.array([start_pose(int(lo), self.cfg.map_max_x) for lo in los], dtype=np.float64)
            self.engine.reset(todo.astype(np.int32), seeds, starts=starts)
            active = torch.zeros(n, dtype=torch.uint8, device=self.device)
            active[torch.as_tensor(todo, device=self.device)] = 1
            for _ in range(4):
                self.engine.step(self._scan, active)
            (n_lm,) = self.engine.fetch(self.engine.counts_dev()[:, 1])  # (with the status check, one synchronisation)
            empty = np.array([i for i in todo if n_lm[i] < 1], dtype=np.int64)
            self.env_index[empty] += 50  # "regenerate a environment"
            todo = empty
        self._graph = None
        return self._get_obs()

    def _get_obs(self):
        return None  # the occupancy grids stay on the device; use obs(i) for one env's VirtualMap.to_array()

    def obs(self, i):
        return self.engine.virtual_map(int(i))[0]

    # ------------------------------------------------------------------ graph export
    def graph_matrix(self, plan=False):
        """All envs' graphs as one batch (see Engine.graph); node order per env = [landmarks (hash order), poses,
        frontiers], edge order = DeepQ.data_process's. Candidate c = (env, frontier) in env-major order.
        plan=True (the trainers): the line plans to all frontiers come out of the same call and the same synchronisation
        (Engine.graph(plan=True)); actions_all_goals() then has nothing left to do."""
        g = self.engine.graph(plan=plan)  # (one synchronisation: the status word, the batch's boundaries and the frontier counts)
        nfr_h = g["n_frontier_h"].astype(np.int64)
        n_cand = int(nfr_h.sum())
        self._graph = g
        # the candidate arrays are host arithmetic on the frontier counts: one upload instead of a dozen small launches
        first_h = np.cumsum(nfr_h) - nfr_h
        env_h = np.repeat(np.arange(self.n_envs, dtype=np.int64), nfr_h)
        fidx_h = np.arange(n_cand, dtype=np.int64) - first_h[env_h]
        node_h = g["node_off_h"][1:].astype(np.int64)[env_h] - nfr_h[env_h] + fidx_h  # node_off[e+1] - n_frontier[e] + f
        mf = g["frontier_xy"].shape[1]
        up = torch.as_tensor(np.concatenate([env_h, fidx_h, node_h, env_h * mf + fidx_h, first_h]), device=self.device)
        env_t, fidx, node_t, slot_t, first = (up[:n_cand], up[n_cand:2 * n_cand], up[2 * n_cand:3 * n_cand], up[3 * n_cand:4 * n_cand],
                                              up[4 * n_cand:])
        self._cand_env = env_t.to(torch.int32)
        self._cand_first, self._cand_first_h = first, first_h
        self._cand_fidx = fidx
        self._cand_node = node_t
        self._goals = g["frontier_xy"].view(-1, 2)[slot_t].contiguous()
        self._n_act_h = None
        self._actions = self._n_act = None
        if plan:
            self._actions = g["actions_pad"].view(self.n_envs * mf, -1, 3)[slot_t].contiguous()
            self._n_act = g["n_act_pad"].view(-1)[slot_t].contiguous()
            self._n_act_h = g["n_act_pad_h"].reshape(-1)[env_h * mf + fidx_h]
        return g

    @property
    def candidates(self):
        """(cand_env [C] i32, cand_node [C] i64 global node ids, first candidate of each env [n_envs] i64)."""
        return self._cand_env, self._cand_node, self._cand_first

    def actions_all_goals(self):
        """Line plan to every frontier: (actions [C, max_actions, 3] f64, n_actions [C] i32)."""
        if self._graph is None:
            self.graph_matrix()
        if self._actions is not None:  # (graph_matrix(plan=True) planned already)
            return self._actions, self._n_act
        self._actions, self._n_act = self.engine.line_plan(self._cand_env, self._goals)
        # the plans' lengths on the host: the look-ahead and the step launch the action indices some plan reaches
        (self._n_act_h,) = self.engine.fetch(self._n_act)
        return self._actions, self._n_act

    def rewards_all_goals(self, all_actions=None, return_raw=False):
        """Look-ahead reward per candidate, normalised per env like exploration_env.py:151-161:
        nearest frontier is the arg-max -> interp to [-1, 0], loop_clo False; else [-1, 1], loop_clo True."""
        actions, n_act = all_actions if all_actions is not None else (self._actions, self._n_act)
        n_act_h = self._n_act_h if all_actions is None else None
        if n_act_h is not None:
            kmax = max(int(n_act_h.max()), 1) if n_act_h.size else 1
        else:
            kmax = max(int(n_act.max().item()), 1) if n_act.numel() else 1  # host bound: unreached action indices are not launched
        raw = self.engine.lookahead(self._cand_env, actions, n_act, max_n_actions=kmax)
        r, self.loop_clo = normalise_rewards(raw, self._cand_env.long(), self._cand_first, self.n_envs, self._graph["n_frontier"])
        return (r, raw) if return_raw else r

    # ------------------------------------------------------------------ step (exploration_env.py:98-105)
    def step(self, choice, check=True):
        """Execute, for every env, the plan of its chosen candidate. `choice` [n_envs] = frontier index within the env
        (int tensor / array) — `all_actions[key_size + action_index]` of policy.py:120.  A host array costs no synchronisation
        (the plans' lengths are on the host since actions_all_goals); check=False leaves the status check to the caller's next
        `engine.fetch` / `engine.check_status` (the trainers read the step's results with one)."""
        kmax = None
        if not torch.is_tensor(choice) and self._n_act_h is not None:
            c_h = self._cand_first_h + np.asarray(choice, dtype=np.int64)
            kmax = int(self._n_act_h[c_h].max())
            c = torch.as_tensor(c_h, device=self.device)
        else:
            c = self._cand_first + torch.as_tensor(choice, device=self.device).long()
        self.last_pick = c  # the chosen candidates' indices (device, int64)
        acts = self._actions[c]  # [n_envs, A, 3]
        nact = self._n_act[c]
        return self.step_actions(acts, nact, kmax=kmax, check=check)

    def step_actions(self, acts, nact, map_every_action=False, kmax=None, check=True):
        """Execute `nact[i]` actions of `acts[i]` in every env.  The virtual map is a pure function of the SLAM state, so it
        is rebuilt at each env's last action only unless `map_every_action` (per-step metrics of the map) asks otherwise.
        kmax: a host bound of nact (None: read from the device)."""
        if kmax is None:
            kmax = int(nact.max().item())
        if acts.shape[1] < self.cfg.max_actions:  # drlgx_step_plan strides the plans by the engine's max_actions
            acts = torch.cat([acts, acts.new_zeros(acts.shape[0], self.cfg.max_actions - acts.shape[1], 3)], dim=1)
        acts = acts.contiguous()
        nact = nact.to(torch.int32).contiguous()
        self.engine.step_plans(acts, nact, kmax, map_last_only=not map_every_action)
        live = torch.arange(acts.shape[1], device=self.device)[None, :] < nact[:, None]
        self.dist += (torch.sqrt(acts[:, :, 0] ** 2 + acts[:, :, 1] ** 2) * live).sum(dim=1)
        if check:
            self.engine.check_status()
        self._graph = None
        return self._get_obs(), self.done(), {}

    def status(self):
        return self.engine.explored()

    def done(self):
        """exploration_env.py:107-110: `_done or step > max_steps or status() > 0.85` - the reference's terminal
        condition and nothing else."""
        c = self.engine.counts_dev()
        return (self.status() > 0.85) | (c[:, 3] > self._max_steps)

    def truncated(self):
        """Engine-side condition with no counterpart in the reference: envs whose trajectory is within one plan
        (max_actions poses) of the pose capacity.  Callers reset them WITHOUT treating the transition as terminal
        (the reference's episodes end on `explored > 0.85` only); with the default capacity of 256 poses this does not
        occur on the reference's maps."""
        c = self.engine.counts_dev()
        return c[:, 0] + self.cfg.max_actions + 1 > self.cfg.max_poses

    # ------------------------------------------------------------------ reporting
    def metrics(self, sigma0=1.0):
        """[n_envs, 3] float64 device tensor: get_landmark_error(), scripts/test.py's map_entropy(obs) and
        max_uncertainty_of_trajectory() of every env (scripts/test.py:136-142), computed on the device."""
        return self.engine.metrics(sigma0)

    # (host, one env)
    def get_landmark_size(self, i):
        return self.engine.counts(int(i))["landmarks"]

    def get_landmark_error(self, i, sigma0=1.0):
        """exploration_env.py:170-177."""
        keys, xy, _ = self.engine.landmarks(int(i))
        _, gt = self.engine.ground_truth(int(i))
        err = float(np.sqrt(((gt[keys] - xy) ** 2).sum(axis=1)).sum()) if len(keys) else 0.0
        n_gt = self.cfg.num_landmarks
        return (err + sigma0 * (n_gt - len(keys))) / n_gt

    def max_uncertainty_of_trajectory(self, i):
        """exploration_env.py:190-194."""
        _, X = self.engine.adjacency(int(i))
        return float(np.amax(X[self.get_landmark_size(i):]))
