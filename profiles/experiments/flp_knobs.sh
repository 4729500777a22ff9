# Frame-pair form knobs: LDS budget, workgroup size, tile-shape weight.  gpurun -- bash profiles/experiments/flp_knobs.sh
cd "$GRAFT_REPO_ROOT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', d['config']['kernel'], 'Gpix/s=%.1f'%(d['value']/1e3), 'valu_frac=%.3f'%r['valu_frac'], 'kernel_ms=%.3f'%r['kernel_ms_per_launch'])"; }
for c in A137 N15 N480; do
  for kb in 56 64 72 80; do JINC_FLP_LDS_KB=$kb python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | line "$c lds=$kb"; done
  for th in 256 384; do JINC_FLP_THREADS=$th python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | line "$c threads=$th"; done
  for w in 0 0.25 1.0 2.0; do JINC_FLP_COLW=$w python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 | line "$c colw=$w"; done
done
