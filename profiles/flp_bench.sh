#!/bin/bash
# Frame-pair form against the 64-frame form (kernel mode 11) on the frame-lane configurations: gpurun -- bash profiles/flp_bench.sh [frames]
cd "$GRAFT_REPO_ROOT" || exit 1
fr=${1:-256}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', d['config']['kernel'], 'Gpix/s=%.1f'%(d['value']/1e3), 'valu_frac=%.3f'%r['valu_frac'], 'kernel_ms=%.3f'%r['kernel_ms_per_launch'])"; }
for c in ${CONFIGS:-A137 A1875 N15 N480 N3 C2}; do
  python bench.py --config $c --frames $fr --steps 20 --warmup 3 --no-cpu-baseline --kernel-mode 11 2>/dev/null | tail -1 | line "$c 64-frame form"
  python bench.py --config $c --frames $fr --steps 20 --warmup 3 --no-cpu-baseline --kernel-mode 12 2>/dev/null | tail -1 | line "$c pair form    "
  python bench.py --config $c --frames $fr --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line "$c automatic    "
done
