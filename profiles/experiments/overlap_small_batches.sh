# Border kernels on the side stream (fork / join through events) against the same stream, for single frames and small batches.
cd "$GRAFT_REPO_ROOT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-30s %-26s %8.1f Gpix/s  step %.4f ms' % ('$1', d['config']['kernel'], d['value']/1e3, d['ms_per_step']))"; }
for c in C2 C3 C4 N3 N15 D12 T6; do for f in 1 2 4 8 16; do for o in 0 1; do
  python bench.py --config $c --frames $f --steps 100 --warmup 10 --no-cpu-baseline --border-overlap $o 2>/dev/null | grep "^{" | tail -1 | line "$c frames=$f overlap=$o"
done; done; done
