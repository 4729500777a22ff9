# Batch size from which the frame-lane kernels beat the single-frame kernels (gather / quasi-periodic).
cd "$GRAFT_REPO_ROOT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-26s %-28s %8.1f Gpix/s  step %.3f ms' % ('$1', d['config']['kernel'], d['value']/1e3, d['ms_per_step']))"; }
for c in N15T8 A137 D169 N15T4 N480; do
  for f in 8 16 24 32 48; do
    for m in 0 1 11; do
      python bench.py --config $c --frames $f --steps 30 --warmup 5 --no-cpu-baseline --kernel-mode $m 2>/dev/null | grep "^{" | tail -1 | line "$c frames=$f mode=$m"
    done
  done
done
