#!/bin/bash
# Sub-group form of the frame-lane kernel: tile size by the number of workgroups a launch should have (JINC_FL_SUB_MIN_BLOCKS).
# profiles/fl_sub_tiles_ab.sh <tag>
tag=${1:-r4y}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_fl_sub_tiles_ab.log
: > $out
run() {
  label=$1; shift
  line=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"], "valu", r["valu_frac"])')" >> $out
}
for cfg in ${CONFIGS:-A137L16 A137L32 A137L4 D169L16}; do
  for b in ${BLOCKS:-1024 3072 6144 12288}; do
    JINC_FL_SUB_MIN_BLOCKS=$b run "min_blocks_$b" --config $cfg
  done
done
cat $out
