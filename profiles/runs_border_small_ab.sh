#!/bin/bash
# Border frame of runs-form plans (drifting, fs 9) in small batches: gather kernel (JINC_RUNS_FL_BORDER_FRAMES=0) against the
# frame-lane kernel -- sub-group form up to 32 frames -- from 3 frames (=3).  profiles/runs_border_small_ab.sh <tag>
tag=${1:-r4y}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_runs_border_small_ab.log
: > $out
run() {
  label=$1; shift
  line=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"], "step-interior", r.get("step_minus_interior_ms"))')" >> $out
}
for round in 1 2; do
for cfg in ${CONFIGS:-N15T4 N3T4 N480T4}; do
for n in ${FRAMES:-4 8 16 32 48}; do
  JINC_RUNS_FL_BORDER_FRAMES=0 run "gather   " --config $cfg --frames $n
  JINC_RUNS_FL_BORDER_FRAMES=3 run "framelane" --config $cfg --frames $n
done
done
done
cat $out
