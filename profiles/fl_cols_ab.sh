#!/bin/bash
# Border columns of periodic plans in batches: column-strip kernel (JINC_FL_COLS_FRAMES=0) against the frame-lane kernel.
tag=${1:-r4x}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_fl_cols_ab.log
: > $out
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  line=$(env "${envs[@]}" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"], "step-interior", r["step_minus_interior_ms"], "border_alone", r["border_ms_alone"])')" >> $out
}
for round in 1 2; do
for cfg in C2 C1 C3 C2H D12; do
  run "colstrip " JINC_FL_COLS_FRAMES=0 -- --config $cfg
  run "framelane" JINC_FL_COLS_FRAMES=64 -- --config $cfg
done
done
run "colstrip  64" JINC_FL_COLS_FRAMES=0 -- --config C2 --frames 64
run "framelane 64" JINC_FL_COLS_FRAMES=64 -- --config C2 --frames 64
run "colstrip  128" JINC_FL_COLS_FRAMES=0 -- --config C2 --frames 128
run "framelane 128" JINC_FL_COLS_FRAMES=64 -- --config C2 --frames 128
cat $out
