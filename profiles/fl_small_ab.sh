#!/bin/bash
# Calls of 1 .. 12 frames of plans without phase structure: the automatic choice (gather kernel below 16 frames in round 3) against
# the frame-lane kernel's sub-group form forced (kernel mode 16).  profiles/fl_small_ab.sh <tag>
tag=${1:-r4y}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_fl_small_ab.log
: > $out
run() {
  label=$1; shift
  line=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"], "valu", r["valu_frac"])')" >> $out
}
for cfg in ${CONFIGS:-A137 D169 A1875}; do
for n in ${FRAMES:-1 2 3 4 6 8 12}; do
  JINC_FL_SUB=0 run "gather   " --config $cfg --frames $n --kernel-mode 1
  run "auto     " --config $cfg --frames $n
  run "sub_mode16" --config $cfg --frames $n --kernel-mode 16
done
done
cat $out
