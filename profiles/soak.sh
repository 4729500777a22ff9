#!/bin/bash
# Soak run: the seeded sweeps widened (every kernel family, every forced mode the sweeps cycle through), at HEAD.
# profiles/soak.sh <tag>
tag=${1:-r4y}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_soak.log
echo "# soak at HEAD $(date -u +%FT%TZ)" > $out
step() {
  echo "== $*" >> $out
  bash -c "$*" 2>&1 | tail -3 >> $out
  echo "   rc=${PIPESTATUS[0]}" >> $out
}
step "JINC_SWEEP_SEEDS=${SOAK_PARITY_SEEDS:-2000} timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised"
step "JINC_SWEEP_SEEDS=600 timeout -k 10 400 python -m pytest tests/test_framelane.py tests/test_framelane_pair.py -m gpu -q -k randomised"
step "JINC_RUNS_SWEEP_SEEDS=1500 timeout -k 10 300 python -m pytest tests/test_direct_runs.py -m gpu -q -k randomised"
echo finished >> $out
cat $out
