# PMC passes over the dense K2 kernels (tools/k2_dense.py at the DBLP APTPA shape); summarised by tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_dense -o p$i -- python3 tools/k2_dense.py n=4057 dens=12924399 > gpurun_out/pmc_dense_$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py gpurun_out/pmc_dense > gpurun_out/pmc_dense_summary.json
