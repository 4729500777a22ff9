#!/bin/bash
# Same-box A/B of library builds on bench configurations, builds alternating:
#   bash profiles/ab_libs.sh <tag> <rounds> "<configs>" name=path/to/lib.so ...
tag=$1; rounds=$2; configs=$3; shift 3
mkdir -p gpurun_out/$tag
for r in $(seq 1 $rounds); do
  for c in $configs; do
    for nl in "$@"; do
      n=${nl%%=*}; l=${nl#*=}
      JINC_LIB=$PWD/$l timeout 120 python bench.py --config $c --no-cpu-baseline --no-e2e --no-clock-sampler 2>/dev/null | tail -1 > gpurun_out/$tag/${c}_${n}_$r.json
      python - <<PY | tee -a gpurun_out/$tag/table.txt
import json
d=json.loads(open("gpurun_out/$tag/${c}_${n}_$r.json").read())
r=d["roofline"]
print("$c", "$n", $r, round(d["value"]), "valu", r.get("valu_frac"), "kernel_ms", r.get("kernel_ms"))
PY
    done
  done
done
