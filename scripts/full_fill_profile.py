"""Dev tool: the 2 048-env (eight workgroups per CU) belief-step workload of bench.py's `full_fill` section alone, for
rocprofv3 (kernel stats / PMC passes): python scripts/full_fill_profile.py [envs = 2048] [steps = 40]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.full_fill_bench(0, n_envs=int(sys.argv[1]) if len(sys.argv) > 1 else 2048,
                                       steps=int(sys.argv[2]) if len(sys.argv) > 2 else 40)))
