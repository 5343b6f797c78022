"""Dev tool: run the GCN forward / train step of bench.policy_bench a few times (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
eng, cfg = bench.make_engine(0, 0)
print(bench.policy_bench(eng, eng.device, iters=20))
