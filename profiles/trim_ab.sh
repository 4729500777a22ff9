#!/bin/bash
# Same-box A/B of the trimmed support (round 4): full window (kernel mode 15) against the trimmed support in its forms.
#   profiles/trim_ab.sh <tag>
tag=${1:-r4t}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_trim_ab.log
: > $out
run() {  # label, env assignments..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  line=$(env "${envs[@]}" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"])')" >> $out
}
for round in 1 2; do
for cfg in C2 C1 C2YUV C2H C3 T6; do
  run "full_window        " X=1 -- --config $cfg --kernel-mode 15
  run "trimmed_auto       " X=1 -- --config $cfg
  run "trimmed_quad_rg8   " JINC_QUAD_RG=8 -- --config $cfg --kernel-mode 13
  run "trimmed_quad_rg4   " JINC_QUAD_RG=4 -- --config $cfg --kernel-mode 13
  run "trimmed_window     " X=1 -- --config $cfg --kernel-mode 2
  run "trimmed_rows       " X=1 -- --config $cfg --kernel-mode 3
done
done
for n in 1 4 16 64; do
  run "full_window   frames=$n" X=1 -- --config C2 --frames $n --kernel-mode 15
  run "trimmed_auto  frames=$n" X=1 -- --config C2 --frames $n
  run "trimmed_quad4 frames=$n" JINC_QUAD_RG=4 -- --config C2 --frames $n --kernel-mode 13
  run "trimmed_quad8 frames=$n" JINC_QUAD_RG=8 -- --config C2 --frames $n --kernel-mode 13
  run "trimmed_win   frames=$n" X=1 -- --config C2 --frames $n --kernel-mode 2
done
cat $out
