#!/bin/bash
# Frame-lane kernel, groups of fewer than 64 frames: 64-frame form (JINC_FL_SUB=0) against the sub-group form (round 4).
# profiles/fl_sub_ab.sh <tag>
tag=${1:-r4y}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_fl_sub_ab.log
: > $out
run() {
  label=$1; shift
  line=$(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"], "valu", r["valu_frac"])')" >> $out
}
for cfg in ${CONFIGS:-A137 D169 N15T4 A1875}; do
for n in ${FRAMES:-8 16 24 32 48}; do
  JINC_FL_SUB=0 run "form64 " --config $cfg --frames $n --kernel-mode 11
  for g in 2 4 8; do
    JINC_FL_SUB=$g run "sub_g$g " --config $cfg --frames $n --kernel-mode 11
  done
done
done
cat $out
