#!/bin/bash
# Device-resident rates at given frames per call: bash profiles/frames_rates.sh <tag> "<configs>" "<frames list>" [extra bench args]
tag=$1; configs=$2; frames=$3; shift 3
mkdir -p gpurun_out/$tag; out=gpurun_out/$tag/rates.txt
for c in $configs; do
  for f in $frames; do
    timeout 120 python bench.py --config $c --frames $f --steps 50 --warmup 5 --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-6s frames=%-3d %-28s %8.1f Gpix/s  valu %.3f  step %.3f ms' % ('$c', $f, d['config']['kernel'], d['value']/1e3, r['valu_frac'], d['ms_per_step']))" | tee -a $out
  done
done
