# (the two knobs this script drives, JINC_AUX_PRIORITY / JINC_BORDER_AFTER, were removed again after the measurement: no effect; log in profiles/round2/border_order_ab.log)
# Border kernels: side-stream priority and queueing order against the interior kernel.  gpurun -- bash profiles/experiments/border_order_ab.sh
cd "$GRAFT_REPO_ROOT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'Gpix/s=%.2f'%(d['value']/1e3), 'valu_frac=%.3f'%r['valu_frac'], 'step_ms=%.3f'%d['ms_per_step'], 'border_ms=%s'%r['border_kernel_ms_per_step'])"; }
for c in C3 C4 C2 T16 T6 D12; do
  for prio in 1 0 -1; do for after in 0 1; do
    JINC_AUX_PRIORITY=$prio JINC_BORDER_AFTER=$after python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line "$c prio=$prio after=$after"
  done; done
done
