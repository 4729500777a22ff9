cd "$GRAFT_REPO_ROOT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', d['config']['kernel'], 'Gpix/s=%.1f'%(d['value']/1e3), 'valu_frac=%.3f'%r['valu_frac'], 'kernel_ms=%.3f'%r['kernel_ms_per_launch'])"; }
for c in N15T4 D169; do
  for kb in 64 72 80; do
    JINC_FL_LDS_KB=$kb python bench.py --config $c --frames 128 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line "$c lds=$kb"
  done
done
