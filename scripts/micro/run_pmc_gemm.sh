cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/pmc_gcn; rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_gcn -o pmc -- python /root/repo/scripts/profile_gcn.py > /tmp/pmc_gcn.log 2>&1
  python /root/repo/scripts/pmc_gemm_summary.py $(find /tmp/pmc_gcn -name "*.db" | head -1) "k_gemm<false, false, 1" > /root/repo/gpurun_out/pmc_gemm_$i.txt 2>&1
done
