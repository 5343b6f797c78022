#!/bin/bash
# Dev loop for k_map changes on the GPU box: in-kernel phases, belief parity tests, short bench line.
python scripts/phase_profile_map.py 2>&1 | tail -13
timeout 800 python -m pytest tests/test_gpu_belief.py -x -q 2>&1 | tail -4
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-policy --no-train 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print({k:round(v['avg_us_per_launch'],2) for k,v in d['kernels'].items()})"
