#!/usr/bin/env python
"""bench.py — env-steps/sec of the MI355X belief-step hot path (BASELINE.json metric).

Workload (BASELINE.json configs[1]): 256 parallel environments per GPU, 40x40 m map
(V = 1600 virtual cells), 100 ground-truth landmarks, every env warmed with a fixed motion script to a
~64-node pose/landmark graph (36 poses + the landmarks seen so far) and snapshotted on the device.
One timed "step" = one belief update (SS2D.simulate(core=True): move, 2 noisy measure calls, factor
append, iSAM2-policy SLAM solve + all block marginals, occupancy + virtual-map rebuild, utility
reductions) for all 256 envs, starting from the snapshot (the device-side restore is inside the timed
region so the graph size stays at the quoted ~64 nodes).  All inputs are resident in HBM.

Multi-GPU: one process per GPU (torch.distributed / RCCL), independent environments per rank, no
data-path collective (weak scaling); time = max over ranks.

Output: ONE JSON line on rank 0 (see the driver contract), with `roofline` for the time-dominant
kernel (per-kernel HIP-event timing on the engine stream), `kernels` for all three and
`cpu_baseline` = the CPU oracle ("port") timed single-threaded on the same box.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS = 256
MAP = 40
NUM_LM = 100
WARM_SCRIPT = [(1, 1, math.pi / 2)] * 4 + [(2, 0, 0), (2, 0, 0), (0, 0, 0.6)] * 10 + [(2, 0, 0)]  # -> 36 poses
STEP_ACTION = (2.0, 0.0, 0.0)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(P, L, M, V):
    """SURVEY.md §8(d) per-env-step algorithmic bytes (fp64, perfect on-chip reuse)."""
    n = 3 * P + 2 * L
    cov = 96 * P + 41 * V + 16          # covariance propagation + utility (a9 + a10)
    occ = 24 * P + 16 * L + 8 * V       # occupancy rebuild (a7/a8)
    slam = 56 * P + 40 * M + 2 * 8 * n + 72 * P + 32 * L  # factors, state r/w, marginal blocks (a5/a6)
    sim = 16 * NUM_LM + 4 * 2 * 626 * 2 + 32 * 2  # GT landmarks, two mt19937 streams r/w, poses (a2-a4)
    return {"map": cov + occ, "slam": slam, "sim": sim}


def make_engine(device_index, seed0):
    from drl_graph_exploration_amd import default_config
    from drl_graph_exploration_amd.engine import Engine
    cfg = default_config(MAP, num_landmarks=NUM_LM, max_poses=41, max_landmarks=100, max_factors=512, max_snapshots=1)
    eng = Engine(cfg, N_ENVS, 0, device=device_index)
    ids = np.arange(N_ENVS)
    eng.reset(ids, seed0 + ids, los=seed0 + ids)
    for act in WARM_SCRIPT:
        eng.step(torch.tensor([act] * N_ENVS, dtype=torch.float64, device=eng.device))
    eng.check_status()
    eng.snapshot(0)
    return eng, cfg


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (a port of the reference arithmetic, not the reference binary) on ONE host thread:
    same config, envs warmed with the same script, belief updates timed on fresh clones."""
    from oracle import oracle as O
    ocfg = O.default_config(MAP, num_landmarks=NUM_LM)
    sims = []
    t_prep = time.time()
    for lo in range(4):
        s = O.OracleSim(ocfg, lo, lo)
        for act in WARM_SCRIPT:
            s.simulate(act)
        sims.append(s)
        if time.time() - t_prep > budget_s:
            break
    n = 0
    t = 0.0
    while t < budget_s:
        for s in sims:
            c = s.clone()
            t0 = time.perf_counter()
            c.simulate(STEP_ACTION)
            t += time.perf_counter() - t0
            n += 1
    return {"value": n / t, "unit": "env-steps/sec", "cores": 1, "kind": "port",
            "sample": "%d belief updates (SS2D.simulate) on %d seeded envs at 36 poses / ~%d landmarks, %.1f s of CPU work, "
                      "single thread, oracle/drlgx_oracle.cpp -O3" % (n, len(sims), sims[0].num_landmarks(), t),
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    eng, cfg = make_engine(local_rank, seed0=rank * N_ENVS)
    odom = torch.tensor([STEP_ACTION] * N_ENVS, dtype=torch.float64, device=dev)
    counts = [eng.counts(i) for i in range(0, N_ENVS, 16)]
    P = float(np.mean([c["poses"] for c in counts]))
    L = float(np.mean([c["landmarks"] for c in counts]))
    M = float(np.mean([c["factors"] for c in counts]))

    def one_step():
        eng.restore(0)
        eng.step(odom)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    eng.check_status()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # second pass with per-kernel HIP events on the engine stream (kept out of the headline region)
    eng.timing_enable(True)
    eng.timing_read()
    for _ in range(args.steps):
        one_step()
    tm = eng.timing_read()
    eng.timing_enable(False)

    if rank == 0:
        V = eng.rows * eng.cols
        ab = algorithmic_bytes(P + 1, L, M, V)
        kernels = {}
        for name in ("sim", "slam", "map", "copy"):
            ms, n = tm[name]
            if n == 0:
                continue
            avg_us = ms / n * 1e3
            b = ab.get(name)
            ent = {"avg_us_per_launch": avg_us, "launches": int(n)}
            if b is not None:
                ent["algorithmic_bytes_per_launch"] = b * N_ENVS
                ent["achieved_GBs"] = b * N_ENVS / (avg_us * 1e-6) / 1e9
                ent["frac_hbm_peak"] = ent["achieved_GBs"] / HBM_PEAK_GBS
            kernels[name] = ent
        dom = max(("sim", "slam", "map"), key=lambda k: kernels[k]["avg_us_per_launch"])
        roofline = {"kernel": {"sim": "k_sim_step", "slam": "k_slam", "map": "k_map"}[dom], "bound": "hbm",
                    "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": kernels[dom]["frac_hbm_peak"], "traffic": None,
                    "avg_us_per_launch": kernels[dom]["avg_us_per_launch"]}
        total_steps = args.steps * N_ENVS * world
        out = {
            "metric": "env-steps/sec (256 parallel envs, ~64-node graphs)", "value": total_steps / elapsed,
            "unit": "env-steps/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: 256 envs/GPU, 40x40 map (V=1600), 100 landmarks, belief update at "
                                   "~%.0f-node graphs (P=%.1f poses, L=%.1f landmarks, M=%.0f factors) from a device snapshot"
                                   % (P + 1 + L, P + 1, L, M),
                       "envs_per_gpu": N_ENVS, "map_size": MAP, "num_landmarks": NUM_LM, "parallelism": "env-sharded x%d" % world},
            "roofline": roofline, "kernels": kernels,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["gpu_over_cpu_thread"] = (out["value"] / world) / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
