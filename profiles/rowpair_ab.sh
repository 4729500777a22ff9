#!/bin/bash
# ewa_periodic_rowpair_kernel (round 5) against ewa_periodic_rows_kernel and across its tile shapes, same box, alternating:
#   profiles/rowpair_ab.sh [configs ...]      (through gpurun from the repo root; default: C3 T6)
# JINC_ROWS_PAIR: 0 = rows kernel, 1 = automatic tile shape, 64 / 32 / 16 = lanes along x (tiles of 256 x 16, 128 x 32, 64 x 64 periods)
cfgs=${@:-C3 T6}
mkdir -p gpurun_out
for round in 1 2; do
  for c in $cfgs; do
    for k in 0 1 64 32 16; do
      JINC_ROWS_PAIR=$k python bench.py --config $c --no-cpu-baseline --no-e2e --no-clock-sampler > gpurun_out/rowpair_ab_${c}_$k.json 2> gpurun_out/rowpair_ab.err || { echo "$c knob $k FAILED"; tail -3 gpurun_out/rowpair_ab.err; continue; }
      python - "$c" "$k" "$round" <<'PY'
import json, sys
c, k, rnd = sys.argv[1:4]
d = json.load(open(f"gpurun_out/rowpair_ab_{c}_{k}.json")); r = d["roofline"]
print(f"round {rnd} {c} ROWS_PAIR={k}: {d['value'] / 1e3:.1f} Gpix/s  {r['kernel']}  valu_frac {r['valu_frac']}  taps {r['taps_per_sample_executed']}  self_check {d['self_check']}")
PY
    done
  done
done
