"""Dev tool: GCN forward (256-graph batch) and 64-graph train step times of bench.policy_bench, for A/B runs of kernel
variants (DRLGX_LIB_DEV=<variant library>)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
eng, cfg = bench.make_engine(0, 0)
r = bench.policy_bench(eng, eng.device, iters=20)
print(os.environ.get("DRLGX_LIB_DEV", "product"), {k: round(r[k], 4) for k in ("gcn_forward_ms", "train_step_ms", "graph_nodes", "train_step_nodes")})
eng.close()
