#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 passes over the bench workload, summarised into gpurun_out/<tag>/.
#   1. --kernel-trace --stats            -> kernel_stats.csv
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)  -> pmc_traffic.json (with the digest of the kernel sources)
#   3. --pmc SQ_* (one pass, 8 SQ slots) + GRBM_GUI_ACTIVE     -> sq_counters.txt
# usage: scripts/collect_profiles.sh <tag>     (counters are never combined with sys / hip / hsa tracing)
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-policy --no-train"
db() { find "$1" -name '*_results.db' | head -1; }
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -- $BENCH > $out/bench_under_rocprof.json 2> $out/rocprof_stats.err
python scripts/rocpd_summary.py "$(db /tmp/prof_stats)" > $out/kernel_stats.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_fetch -- $BENCH > /dev/null 2> $out/rocprof_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_write -- $BENCH > /dev/null 2> $out/rocprof_write.err
python scripts/rocpd_pmc.py "$(db /tmp/prof_fetch)" "$(db /tmp/prof_write)" > $out/pmc_traffic.json
timeout 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-trace -d /tmp/prof_sq -- $BENCH > /dev/null 2> $out/rocprof_sq.err
timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES \
  --kernel-trace -d /tmp/prof_sq2 -- $BENCH > /dev/null 2> $out/rocprof_sq2.err
python scripts/pmc_sq_summary.py "$(db /tmp/prof_sq)" "$(db /tmp/prof_sq2)" --json $out/sq_counters.json > $out/sq_counters.txt
ls -la $out
# (the counters of THIS tree become the digest-checked files bench.py reads, here on the box too: the bench line below then carries
# roofline.traffic; scripts/install_profiles.sh does the same in the repository afterwards)
[ -s $out/pmc_traffic.json ] && cp $out/pmc_traffic.json profiles/pmc_traffic.json
[ -s $out/sq_counters.json ] && cp $out/sq_counters.json profiles/sq_counters.json
# the bench line itself, the 2-rank control flow on one device (gloo), and the step-vs-trajectory-length tables
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
DRLGX_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 5 > $out/bench_2ranks_gloo.json 2> $out/bench_2ranks_gloo.err
PP_SAMPLE=39,49 timeout 300 python scripts/bench_vs_poses.py 206 100 phases > $out/vs_poses_100lm.txt 2>&1
timeout 300 python scripts/bench_vs_poses.py 206 8 phases > $out/vs_poses_8lm.txt 2>&1
# the look-ahead workload: kernel stats and HBM traffic of its kernels (k_step_loop: a candidate's whole action list per launch)
LA="python scripts/lookahead_workload.py"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_la -- $LA > $out/lookahead_workload.txt 2> $out/rocprof_la.err
python scripts/rocpd_summary.py "$(db /tmp/prof_la)" > $out/lookahead_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_la_fetch -- $LA > /dev/null 2> $out/rocprof_la_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_la_write -- $LA > /dev/null 2> $out/rocprof_la_write.err
python scripts/rocpd_pmc.py "$(db /tmp/prof_la_fetch)" "$(db /tmp/prof_la_write)" "\`python scripts/lookahead_workload.py\` (2 633 candidates, 19 475 rollout updates per look-ahead)" > $out/lookahead_pmc_traffic.json
timeout 200 python scripts/lookahead_breakdown.py > $out/lookahead_breakdown.txt 2>&1
timeout 200 python scripts/phase_profile_relin.py > $out/relinearising_update_phases.txt 2>&1
timeout 200 python scripts/phase_profile_blocks.py > $out/step_workgroups.txt 2>&1
timeout 300 python scripts/full_fill_profile.py > $out/full_fill.json 2> /dev/null
# BASELINE config 5 scale (scripts/bench_config5.py: stage kernels, incremental updates with the panel in HBM / L2): kernel stats and HBM traffic
C5="python scripts/bench_config5.py 256 110"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -- $C5 > $out/config5_bench.txt 2> $out/rocprof_c5.err
python scripts/rocpd_summary.py "$(db /tmp/prof_c5)" > $out/config5_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_c5_fetch -- $C5 > /dev/null 2> $out/rocprof_c5_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_c5_write -- $C5 > /dev/null 2> $out/rocprof_c5_write.err
python scripts/rocpd_pmc.py "$(db /tmp/prof_c5_fetch)" "$(db /tmp/prof_c5_write)" "\`python scripts/bench_config5.py 256 110\` (50 m map, 500 landmarks, ~110 poses, 256 envs; warm-up included in the averages)" > $out/config5_pmc_traffic.json
timeout 200 python scripts/phase_profile_config5_blocks.py > $out/config5_workgroups.txt 2>&1
timeout 200 python scripts/phase_profile_config5.py 123 6 > $out/config5_update_phases.txt 2>&1
timeout 200 python scripts/config5_updates.py > $out/config5_updates.txt 2>&1
# the A2C trainer loop (bench.py's a2c_loop workload): wall clock split into stepping and update, the kernels' totals per vector step
# (42 = 2 warm-up + 40 timed steps), the device's idle gaps by the launches around them, the host's profile
SPLIT=1 NOPROF=1 timeout 200 python scripts/a2c_loop_cprofile.py > $out/a2c_loop_split.txt 2>&1
NOPROF=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_a2c -o a2c -- python scripts/a2c_loop_cprofile.py > /dev/null 2> $out/rocprof_a2c.err
timeout 60 python scripts/rocpd_kernel_summary.py "$(db /tmp/prof_a2c)" 30 42 > $out/a2c_loop_kernels.txt 2>&1 < /dev/null
timeout 60 python scripts/rocpd_gaps.py "$(db /tmp/prof_a2c)" 15 1500 > $out/a2c_loop_idle_gaps.txt 2>&1 < /dev/null
NOGC=1 SORT=cumtime TOP=40 timeout 200 python scripts/a2c_loop_cprofile.py > $out/a2c_loop_cprofile.txt 2>&1
timeout 120 python scripts/fetch_latency.py > $out/status_fetch_latency.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_gcn -- python $OLDPWD/scripts/profile_gcn.py > /dev/null 2>&1)
python scripts/rocpd_summary.py "$(db /tmp/prof_gcn)" --by-grid k_ > $out/gcn_kernels_by_grid.csv 2>/dev/null
ls -la $out
