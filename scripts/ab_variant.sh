#!/bin/bash
# Dev tool (GPU box): the product library against a kernel-variant library (scripts/build_variant.sh) on the same box, interleaved:
# the bench's step loop, the look-ahead workload, the 2 048-env fill, config-5 updates, the step-vs-poses table.
#   scripts/ab_variant.sh <variant name>      -> stdout
V=$PWD/drl_graph_exploration_amd/libdrlgx_$1.so
[ -f "$V" ] || { echo "no $V"; exit 1; }
for i in 1 2; do
  python scripts/ab_step.py 300 2>/dev/null; DRLGX_LIB_DEV=$V python scripts/ab_step.py 300 2>/dev/null
done
for i in 1 2; do
  python scripts/lookahead_workload.py 12 2>/dev/null; DRLGX_LIB_DEV=$V python scripts/lookahead_workload.py 12 2>/dev/null
done
python scripts/full_fill_profile.py 2>/dev/null; DRLGX_LIB_DEV=$V python scripts/full_fill_profile.py 2>/dev/null
python scripts/config5_updates.py 2>/dev/null | tail -4; DRLGX_LIB_DEV=$V python scripts/config5_updates.py 2>/dev/null | tail -4
PP_SAMPLE=39,49 python scripts/bench_vs_poses.py 100 100 2>/dev/null | tail -8; DRLGX_LIB_DEV=$V PP_SAMPLE=39,49 python scripts/bench_vs_poses.py 100 100 2>/dev/null | tail -8
