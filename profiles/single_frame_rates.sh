#!/bin/bash
# Device-resident rates of single frames and small batches (what one GetFrame call launches): gpurun -- bash profiles/single_frame_rates.sh
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/round2/single_frame_rates.log
mkdir -p gpurun_out/round2; : > $out
for c in C2 C3 C4 N15 N3 N480 N15T4 N15T8 A137 A1875 D169 D12 T16; do
  for f in 1 4 16 64; do
    timeout 120 python bench.py --config $c --frames $f --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-6s frames=%-3d %-28s %8.1f Gpix/s  valu %.3f  step %.3f ms' % ('$c', $f, d['config']['kernel'], d['value']/1e3, r['valu_frac'], d['ms_per_step']))" >> $out
  done
done
cat $out
