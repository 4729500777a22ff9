#!/bin/bash
# ewa_strip_kernel (round 5: border rows / columns of periodic plans, one register window per lane for the strip's thickness) against
# the round-4 border kernels (row strips on ewa_direct_kernel, columns on the frame-lane / column-strip kernels), same box, alternating:
#   profiles/strip_ab.sh "<configs>" "<frames per call ...>"        JINC_STRIP_LDS: 0 = round-4 kernels, 1 = the rule (rows always, columns below the frame-lane threshold), 2 = rows and columns always
cfgs=${1:-C2 C1 C4 C2T4 C2H}
frames=${2:-0}
for round in 1 2; do
  for c in $cfgs; do
    for n in $frames; do
      for k in 0 1 2; do
        extra=""; [ "$n" != 0 ] && extra="--frames $n"
        JINC_STRIP_LDS=$k python bench.py --config $c $extra --no-cpu-baseline --no-e2e > gpurun_out/strip_${c}_${n}_$k.json 2> gpurun_out/strip.err || { echo "$c $n knob $k FAILED"; tail -3 gpurun_out/strip.err; continue; }
        python - "$c" "$k" "$round" "$n" <<'PY'
import json, sys
c, k, rnd, n = sys.argv[1:5]
d = json.load(open(f"gpurun_out/strip_{c}_{n}_{k}.json")); r = d["roofline"]
print(f"round {rnd} {c} frames {d['config']['frames_per_step_per_gpu']} STRIP_LDS={k}: {d['value'] / 1e3:.1f} Gpix/s  step {d['ms_per_step']} ms  step-interior {r['step_minus_interior_ms']}  border alone {r['border_ms_alone']}  self_check {d['self_check']}")
PY
      done
    done
  done
done
