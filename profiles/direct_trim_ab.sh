#!/bin/bash
# Direct kernel: full window (mode 15) against the trimmed support (round 4).  profiles/direct_trim_ab.sh <tag>
tag=${1:-r4w}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_direct_trim_ab.log
: > $out
run() {
  label=$1; shift
  line=$(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"])')" >> $out
}
for round in 1 2; do
for cfg in ${JINC_AB_CONFIGS:-D12 D23 D13 D12H D12T4 T16 U43 C4 C2F}; do
  run "full_window  " --config $cfg --kernel-mode 15
  run "trimmed_auto " --config $cfg
done
done
cat $out
