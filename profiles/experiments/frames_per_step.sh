cd "$GRAFT_REPO_ROOT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', d['config']['kernel'], 'Gpix/s=%.2f'%(d['value']/1e3), 'valu_frac=%.3f'%r['valu_frac'], 'kernel_ms=%.3f'%r['kernel_ms_per_launch'], 'step_ms=%.3f'%d['ms_per_step'], 'border_ms=%s'%r['border_kernel_ms_per_step'])"; }
for f in 4 8 16 32; do python bench.py --config T16 --frames $f --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line "T16 frames=$f"; done
for f in 16 32 64; do python bench.py --config C3 --frames $f --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line "C3 frames=$f"; done
for f in 8 16; do python bench.py --config C4 --frames $f --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line "C4 frames=$f"; done
for f in 64 256; do python bench.py --config N15T4 --frames $f --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | line "N15T4 frames=$f"; done
