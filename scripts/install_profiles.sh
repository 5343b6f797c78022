#!/bin/bash
# Copy the summaries of a scripts/collect_profiles.sh run (gpurun_out/<tag>/) into profiles/ as <tag>_*, and make them the
# current digest-checked counter files (profiles/pmc_traffic.json, profiles/sq_counters.json) that bench.py reads.
set -eu
tag=$1
src=gpurun_out/$tag
for f in kernel_stats.csv pmc_traffic.json sq_counters.txt sq_counters.json bench.json bench_2ranks_gloo.json gcn_kernels_by_grid.csv; do
  cp $src/$f profiles/${tag}_$f
done
cp $src/vs_poses_100lm.txt profiles/${tag}_step_vs_poses_100lm.txt
cp $src/vs_poses_8lm.txt profiles/${tag}_step_vs_poses_8lm.txt
for f in lookahead_kernel_stats.csv lookahead_pmc_traffic.json lookahead_breakdown.txt relinearising_update_phases.txt step_workgroups.txt full_fill.json \
         config5_bench.txt config5_kernel_stats.csv config5_pmc_traffic.json config5_workgroups.txt config5_update_phases.txt config5_updates.txt \
         a2c_loop_split.txt a2c_loop_kernels.txt a2c_loop_idle_gaps.txt a2c_loop_cprofile.txt status_fetch_latency.txt; do
  [ -f $src/$f ] && cp $src/$f profiles/${tag}_$f
done
cp $src/pmc_traffic.json profiles/pmc_traffic.json
cp $src/sq_counters.json profiles/sq_counters.json
ls -la profiles | grep $tag
