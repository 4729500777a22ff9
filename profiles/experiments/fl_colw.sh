# 64-frame frame-lane forms: tile-shape weight (price of a strip's window columns against the staged footprint).
cd "$GRAFT_REPO_ROOT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', d['config']['kernel'], 'Gpix/s=%.1f'%(d['value']/1e3), 'valu_frac=%.3f'%r['valu_frac'], 'kernel_ms=%.3f'%r['kernel_ms_per_launch'])"; }
for c in D169 N15T4 N15T8; do for w in 0 0.5 1 2 4; do JINC_FL_COLW=$w python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | line "$c colw=$w"; done; done
for w in 0 0.5 1 2; do JINC_FL_COLW=$w python bench.py --config N15 --frames 64 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | line "N15@64 colw=$w"; done
for w in 0 0.5 1 2; do JINC_FL_COLW=$w python bench.py --config A137 --frames 64 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | line "A137@64 colw=$w"; done
