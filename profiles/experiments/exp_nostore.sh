cd "$GRAFT_REPO_ROOT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', d['config']['kernel'], 'Gpix/s=%.1f'%(d['value']/1e3), 'valu_frac=%.3f'%r['valu_frac'], 'kernel_ms=%.3f'%r['kernel_ms_per_launch'])"; }
for c in A137 N15 C2; do
  for m in 11 12; do
  python bench.py --config $c --frames 256 --steps 20 --warmup 3 --no-cpu-baseline --kernel-mode $m 2>/dev/null | tail -1 | line "$c mode $m stores on "
  JINC_FL_EXP_NOSTORE=1 python bench.py --config $c --frames 256 --steps 20 --warmup 3 --no-cpu-baseline --kernel-mode $m 2>/dev/null | tail -1 | line "$c mode $m stores OFF"
  done
done
