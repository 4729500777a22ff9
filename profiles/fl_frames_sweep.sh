#!/bin/bash
# Frames per launch against the frame-lane kernels' rate (tail effect of a one-launch step): gpurun -- bash profiles/fl_frames_sweep.sh
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/fl_frames_sweep.log
: > $out
for c in A137 A1875 D169 N15 N15T4 N480 N15T8; do
  for f in 64 128 256 512; do
    timeout 120 python bench.py --config $c --frames $f --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$c frames=$f', d['config']['kernel'], 'Gpix/s=%.1f'%(d['value']/1e3), 'valu_frac=%.3f'%r['valu_frac'], 'kernel_ms=%.3f'%r['kernel_ms_per_launch'], 'step_ms=%.3f'%d['ms_per_step'])" >> $out
  done
done
cat $out
