# HBM-side (L2-miss) bytes per launch of the K2 kernels, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel trace only), then tools/pmc_traffic.py applies the
# gfx950 correction (FETCH_SIZE counts 128-B requests as 64 B -> x2) and writes profiles/k2_traffic.json.
# Usage (GPU box): bash tools/pmc_traffic.sh && python3 tools/pmc_traffic.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_traffic -o $c -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --hbm-regime-nodes 0 --traffic static > gpurun_out/pmc_traffic_$c.log 2>&1 || echo "pass $c failed"
done
ls gpurun_out/pmc_traffic
