"""GPU tests of bench.py's contract: the default single-GPU line and the self-spawned two-rank run (ranks share the one
GPU of the test box over gloo: the control flow, the rank count and the gradient all-reduce section are what is checked -
the measured configuration is one rank per GPU over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_self_spawned():
    r = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-policy"], {"DRLGX_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["ranks_in_process_group"] == 2 and r["collective_backend"] == "gloo"
    assert r["steps"] == 6 and r["scaling"] == "weak" and r["value"] > 0
    t = r["train_allreduce"]
    assert t["ranks"] == 2 and t["allreduce_ms"] > 0 and t["train_step_ms"] > t["allreduce_ms"] * 0.0
    assert t["env_steps_per_sec_while_training"] > 0
    assert "cpu_baseline" not in r  # N = 1 only


def test_bench_single_gpu_line():
    r = _run(["--steps", "10", "--warmup", "2", "--no-policy", "--no-cpu-baseline"])
    assert r["n_gpus"] == 1 and r["dtype"] == "f64" and r["unit"] == "env-steps/sec" and r["vs_baseline"] is None
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1 and r["roofline"]["kernel"] == "k_step"
    assert r["train_allreduce"]["ranks"] == 1 and r["train_allreduce"]["allreduce_ms"] == 0.0
