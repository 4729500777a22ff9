#!/bin/bash
# ewa_periodic_rowpair1_kernel (one chain row per lane and item, 64 registers: 8 waves per SIMD) against ewa_periodic_rowpair_kernel
# (two chain rows, 71 .. 75 registers: 6 waves) on C3 and on tap 6 at 2x, alternating, same box (VERDICT r5 Next 4).
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
  for c in C3 T6; do
    for k in 1 1016 1032 1064; do
      line=$(timeout 120 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-clock-sampler --knob rows_pair=$k 2>/dev/null | tail -1)
      python - "$c" "$k" "$line" <<'PY'
import json, sys
c, k, line = sys.argv[1:4]
try:
    d = json.loads(line)
    r = d["roofline"]
    print(f"{c} rows_pair={k:5s} {d['value']/1e3:8.1f} Gpix/s  self_check {d.get('self_check')}  kernel {r.get('kernel')}  valu_frac {r.get('valu_frac')}", flush=True)
except Exception as e:
    print(c, k, "no line:", line[:200], e, flush=True)
PY
    done
  done
done
