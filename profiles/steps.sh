#!/bin/bash
# Runs the given commands (one per argument) one after the other on the GPU box, each under its own timeout and with its
# output in gpurun_out/<tag>/<n>.log; a step that fails is recorded and the next one still runs, but a step that was
# KILLED at its limit (rc 124 / 137) ends the call -- no further GPU work after a hang.
# usage: gpurun -- bash profiles/steps.sh <tag> <seconds-per-step> "cmd 1" "cmd 2" ...
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; lim=$2; shift 2
mkdir -p gpurun_out/$tag
n=0; worst=0
for cmd in "$@"; do
  n=$((n+1))
  echo "== step $n: $cmd" | tee -a gpurun_out/$tag/steps.log
  timeout -k 10 "$lim" bash -c "$cmd" > gpurun_out/$tag/$n.log 2>&1
  rc=$?
  echo "   rc=$rc" | tee -a gpurun_out/$tag/steps.log
  tail -n 6 gpurun_out/$tag/$n.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $n was killed at its limit: stopping" | tee -a gpurun_out/$tag/steps.log; exit 124; fi
  [ $rc -ne 0 ] && worst=$rc
done
exit $worst
