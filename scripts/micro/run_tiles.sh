cd /tmp && export TMPDIR=/tmp
(cd /root/repo && python -m pytest tests/test_gpu_gcn.py -x -q -m gpu 2>&1 | tail -3) > /root/repo/gpurun_out/tile_d.txt
rm -rf /tmp/prof_gcn; rocprofv3 --kernel-trace --stats -d /tmp/prof_gcn -o p -- python /root/repo/scripts/profile_gcn.py > /tmp/prof.log 2>&1
grep "graph_nodes" /tmp/prof.log >> /root/repo/gpurun_out/tile_d.txt
python /root/repo/scripts/rocpd_summary.py $(find /tmp/prof_gcn -name "*.db" | head -1) --by-grid "k_" | grep -v "k_step\|k_slam\|k_map\|k_sim" >> /root/repo/gpurun_out/tile_d.txt 2>&1
