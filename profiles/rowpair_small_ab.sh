#!/bin/bash
# ewa_periodic_rowpair_kernel on SHORT kernel rows (taps 3 .. 5 at 2x: 6 .. 11 taps per row) against the window / quad forms that
# are the choice there, same box, alternating:   profiles/rowpair_small_ab.sh [configs ...]   (through gpurun from the repo root)
# JINC_ROWPAIR_SMALL: 0 = never (the quad / window forms), 1 = wherever the plan carries the coefficient pairs
cfgs=${@:-C2 C4 C2F C2H C2T4 C2HT4 C2YUV C1}
mkdir -p gpurun_out
for round in 1 2; do
  for c in $cfgs; do
    for k in 0 1; do
      JINC_ROWPAIR_SMALL=$k python bench.py --config $c --no-cpu-baseline --no-e2e --no-clock-sampler > gpurun_out/rowpair_small_${c}_$k.json 2> gpurun_out/rowpair_small.err || { echo "$c knob $k FAILED"; tail -3 gpurun_out/rowpair_small.err; continue; }
      python - "$c" "$k" "$round" <<'PY'
import json, sys
c, k, rnd = sys.argv[1:4]
d = json.load(open(f"gpurun_out/rowpair_small_{c}_{k}.json")); r = d["roofline"]
print(f"round {rnd} {c} ROWPAIR_SMALL={k}: {d['value'] / 1e3:.1f} Gpix/s  {r['kernel']}  valu_frac {r['valu_frac']}  taps {r['taps_per_sample_executed']}  self_check {d['self_check']}")
PY
    done
  done
done
