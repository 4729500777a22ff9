#!/bin/bash
# Direct kernel's row walk: a wave's lanes as 64 consecutive (column group, row chunk) pairs against 64 / H groups x H row chunks
# (JINC_DIRECT_LANE_ROWS = H): do vertically adjacent row chunks meet in the caches?  profiles/direct_lane_rows_ab.sh <tag>
tag=${1:-r4y}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_direct_lane_rows_ab.log
: > $out
run() {
  label=$1; shift
  line=$(python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"].split(":")[0], d["config"]["frames_per_step_per_gpu"], "value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["kernel"], "kernel_ms", r["kernel_ms_per_launch"], "x", r["launches_per_step"], "valu", r["valu_frac"])')" >> $out
}
for round in 1 2; do
for cfg in ${CONFIGS:-D12F D12H D12 D23}; do
  for h in 0 2 4 8; do
    JINC_DIRECT_LANE_ROWS=$h run "lane_rows_$h" --config $cfg
  done
done
done
cat $out
