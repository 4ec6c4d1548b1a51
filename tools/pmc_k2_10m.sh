# PMC passes over the K2 kernels on the 10M-row table (HBM regime), fp32 and bf16 tables: vector / wait cycles, L2 hit
# rate, TA busy, read latency.  Usage (GPU box): bash tools/pmc_k2_10m.sh ; results under gpurun_out/pmc_k2_10m/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr SQ_INSTS_VALU SQ_WAVES" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  N_ONLY=10000000 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_k2_10m -o p$i -- python3 tools/k2_regimes.py --train > gpurun_out/pmc_k2_10m_$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py gpurun_out/pmc_k2_10m > gpurun_out/pmc_k2_10m.json
