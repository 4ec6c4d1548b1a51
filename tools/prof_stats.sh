# rocprofv3 --kernel-trace --stats of the default bench line; prints the top kernels.
# Usage (GPU box): bash tools/prof_stats.sh <tag> [bench args...]  -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-run}; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o b -- python3 bench.py --no-cpu-baseline --hbm-regime-nodes 0 --traffic static "$@" > gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$TAG/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%-100s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -c 300 gpurun_out/prof_$TAG.log
