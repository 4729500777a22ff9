#!/bin/bash
# ewa_periodic_quad2s_kernel (chains paired by phase over two periods: 31 taps per sample) against ewa_periodic_quad2_kernel
# (pairs = the two column phases of a period: 34 taps), alternating on one box.  usage: quad2s_ab.sh [configs...]
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  for c in ${@:-C2 C1 C2H C2YUV}; do
    for k in 0 1; do
      line=$(timeout 120 python bench.py --config $c --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --no-clock-sampler --knob quad2s=$k 2>/dev/null | tail -1)
      python - "$c" "$k" "$line" <<'PY'
import json, sys
c, k, line = sys.argv[1:4]
try:
    d = json.loads(line)
    r = d["roofline"]
    print(f"{c} quad2s={k} {d['value']/1e3:8.1f} Gpix/s  self_check {d.get('self_check')}  kernel {r.get('kernel')}  kernel ms {r.get('kernel_ms_per_launch', r.get('kernel_ms'))}", flush=True)
except Exception as e:
    print(c, k, "no line:", line[:300], e, flush=True)
PY
    done
  done
done
