"""Dev tool: bench.py's two trainer-loop figures alone (dqn_loop, a2c_loop), a few repetitions; ORDER=a2c: the A2C loop only."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    if os.environ.get("ORDER") == "a2c":
        print(json.dumps({"a2c_ms": round(bench.a2c_loop_bench(0)["ms_per_vector_step"], 2)}))
        continue
    d, a = bench.dqn_loop_bench(0), bench.a2c_loop_bench(0)
    print(json.dumps({"dqn_ms": round(d["ms_per_vector_step"], 2), "a2c_ms": round(a["ms_per_vector_step"], 2)}))
