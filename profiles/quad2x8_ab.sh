#!/bin/bash
# 8 x 8 support: one period per lane (quad8) against two (JINC_QUAD2X8=1).  profiles/quad2x8_ab.sh <tag>
tag=${1:-r4z}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_quad2x8_ab.log
: > $out
run() {
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  line=$(env "${envs[@]}" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"])')" >> $out
}
for round in 1 2; do
for cfg in C4 C2T4 C2HT4; do
  run "quad8  " JINC_QUAD2X8=0 -- --config $cfg
  run "quad2x8" JINC_QUAD2X8=1 -- --config $cfg
done
done
cat $out
