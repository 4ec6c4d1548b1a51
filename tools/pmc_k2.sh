# PMC passes over the K2 kernels (SYN-1M shape, fp32 and bf16 tables): VALU / wait cycles, L2 hit
# rate, TA busy.  Usage (GPU box): bash tools/pmc_k2.sh ; results under gpurun_out/pmc_k2/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM MemUnitStalled" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr SQ_INSTS_VALU SQ_WAVES" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  N_ONLY=1000000 timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_k2 -o p$i -- python3 tools/k2_regimes.py --train > gpurun_out/pmc_k2_$i.log 2>&1 || echo "pass $i failed"
done
ls gpurun_out/pmc_k2 | wc -l
