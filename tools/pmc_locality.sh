# Locality pass (han_amd.reorder) before / after on a graph whose locality hides behind shuffled node ids:
# epoch time (plain bench runs) and the L2 hit rate of the K2 kernels (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum).
# Usage (GPU box): bash tools/pmc_locality.sh [workload]   -> gpurun_out/locality_<workload>.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WL=${1:-syn-1m-local-shuffled}
for R in none bfs; do
  python3 bench.py --workload $WL --reorder $R --steps 5 --warmup 2 --no-cpu-baseline --hbm-regime-nodes 0 --traffic static > gpurun_out/loc_${WL}_$R.json 2> gpurun_out/loc_${WL}_$R.err || echo "bench $R failed"
  timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_loc_${WL}_$R -o p -- python3 bench.py --workload $WL --reorder $R --steps 2 --warmup 1 --no-cpu-baseline --hbm-regime-nodes 0 --traffic static > gpurun_out/pmc_loc_${WL}_$R.log 2>&1 || echo "pmc $R failed"
done
python3 - <<PY
import json, os, sys
sys.path.insert(0, "tools")
import subprocess
out = {"workload": "$WL", "command": "bash tools/pmc_locality.sh $WL"}
for r in ("none", "bfs"):
    d = json.loads(open("gpurun_out/loc_${WL}_%s.json" % r).read().strip().splitlines()[-1])
    pm = json.loads(subprocess.check_output([sys.executable, "tools/pmc_summary.py", "gpurun_out/pmc_loc_${WL}_%s" % r]))
    k2 = {}
    for name, c in pm.items():
        if "node_attn" in name and "TCC_HIT_sum" in c:
            short = "fwd_train" if ("fwd_kernel" in name and "true" in name.split("<")[1].split(",")[1]) else \
                    ("fwd_eval" if "fwd_kernel" in name else ("bwd_cols" if "bwd_cols" in name else None))
            if short:
                k2[short] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 4)
    out[r] = {"epochs_per_s": d["value"], "ms_per_epoch": d["ms_per_step"], "k2_l2_hit_rate": k2,
              "k2_ms": {k: v["avg_launch_ms"] for k, v in d.get("roofline_k2_all", {}).items()},
              "reorder": d["config"].get("reorder")}
json.dump(out, open("gpurun_out/locality_$WL.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
