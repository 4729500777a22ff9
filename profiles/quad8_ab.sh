#!/bin/bash
# 8 x 8 support (tap 4 at 2x): window kernel (JINC_QUAD8=0) against the quad form (JINC_QUAD8=1).  profiles/quad8_ab.sh <tag>
tag=${1:-r4y}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_quad8_ab.log
: > $out
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  line=$(env "${envs[@]}" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"])')" >> $out
}
for round in 1 2; do
for cfg in C4 C2T4 C2HT4; do
  run "full_window " JINC_QUAD8=0 -- --config $cfg --kernel-mode 15
  run "window8     " JINC_QUAD8=0 -- --config $cfg
  run "quad8_rg8   " JINC_QUAD8=1 JINC_QUAD_RG=8 -- --config $cfg
  run "quad8_rg4   " JINC_QUAD8=1 JINC_QUAD_RG=4 -- --config $cfg
done
done
for n in 1 4 16; do
  run "window8 frames=$n" JINC_QUAD8=0 -- --config C2T4 --frames $n
  run "quad8   frames=$n" JINC_QUAD8=1 -- --config C2T4 --frames $n
  run "window8 frames=$n" JINC_QUAD8=0 -- --config C4 --frames $n
  run "quad8   frames=$n" JINC_QUAD8=1 -- --config C4 --frames $n
done
cat $out
