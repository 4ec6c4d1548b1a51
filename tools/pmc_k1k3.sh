cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "MfmaUtil SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "LdsBankConflict MemUnitStalled SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_k3 -o p$i -- python3 tools/kernel_bench.py k3 k1 > gpurun_out/pmc_k3_$i.log 2>&1 || echo "pass $i failed"
done
ls gpurun_out/pmc_k3 | head -30
