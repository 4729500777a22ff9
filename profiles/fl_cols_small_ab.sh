#!/bin/bash
# Border columns of periodic plans in SMALL batches: column-strip kernel (JINC_FL_COLS_FRAMES=0) against the frame-lane kernel's
# sub-group form from N frames (JINC_FL_COLS_FRAMES=N).  profiles/fl_cols_small_ab.sh <tag>
tag=${1:-r4y}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_fl_cols_small_ab.log
: > $out
run() {
  label=$1; shift
  line=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"], "step-interior", r.get("step_minus_interior_ms"))')" >> $out
}
for round in 1 2; do
for cfg in ${CONFIGS:-C2 C1 C2YUV}; do
for n in ${FRAMES:-4 8 16 32 48}; do
  JINC_FL_COLS_FRAMES=0 run "colstrip " --config $cfg --frames $n
  JINC_FL_COLS_FRAMES=3 run "framelane" --config $cfg --frames $n
done
done
done
cat $out
