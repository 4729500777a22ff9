#!/bin/bash
# bash profiles/bench_json.sh <tag> <name> [bench args]: one bench line into gpurun_out/<tag>/<name>.json, key figures printed
tag=$1; name=$2; shift 2
mkdir -p gpurun_out/$tag
timeout 150 python bench.py --no-cpu-baseline --no-e2e --no-clock-sampler "$@" 2>/dev/null | grep "^{" | tail -1 > gpurun_out/$tag/$name.json
python - <<PY | tee -a gpurun_out/$tag/table.txt
import json
d=json.loads(open("gpurun_out/$tag/$name.json").read()); r=d["roofline"]
print("%-14s %-26s %8.1f Gpix/s valu %.3f step %.3f ms kernel %.3f ms x%d border %.3f ms" % ("$name", r["kernel"], d["value"]/1e3, r["valu_frac"], d["ms_per_step"], r.get("kernel_ms_per_launch") or 0, r.get("launches_per_step") or 0, r.get("border_kernel_ms_per_step") or 0))
PY
