# Development batch for one gpurun call: steps run in order, a step that hits its timeout stops the batch
# (no further GPU step after a hang).  Usage: bash tools/dev_batch.sh step1 step2 ...   (steps are shell strings)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for step in "$@"; do
  echo "=== $step"
  bash -c "$step"
  rc=$?
  echo "=== rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the batch"; exit $rc; fi
done
