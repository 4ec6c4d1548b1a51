# PMC passes over the K2 kernels on one small dense graph (tools/k2_small_dense.py); summarised by tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_k2s -o p$i -- python3 tools/k2_small_dense.py n=4057 dens=0.785 > gpurun_out/pmc_k2s_$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py gpurun_out/pmc_k2s > gpurun_out/pmc_k2s_summary.json; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_k2s_summary.json'))
for k,v in d.items():
    if 'node_attn' in k: print(k[:100], json.dumps({a:round(b) for a,b in v.items()}))
PY
