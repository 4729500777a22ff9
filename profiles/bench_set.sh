#!/bin/bash
# One short bench line per configuration: bash profiles/bench_set.sh <tag> CONFIG...   (kernel-only figures: no e2e record, no CPU baseline)
tag=$1; shift
mkdir -p gpurun_out/$tag
for c in "$@"; do
  timeout 120 python bench.py --config $c --no-cpu-baseline --no-e2e --no-clock-sampler 2>/dev/null | tail -1 > gpurun_out/$tag/$c.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/$c.json").read())
r=d["roofline"]
print("$c", round(d["value"]), r.get("kernel"), "valu", r.get("valu_frac"), "ms", d["ms_per_step"])
PY
done
