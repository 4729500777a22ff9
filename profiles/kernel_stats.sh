#!/bin/bash
# Per-kernel time of one bench configuration (rocprofv3 --kernel-trace --stats), top rows printed:
#   profiles/kernel_stats.sh <tag> <config> [extra bench args]
tag=$1; cfg=$2; shift 2
ulimit -c 0
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R" || exit 1
out=gpurun_out/${tag}_stats_$cfg
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out.log 2>&1
tail -1 $out.log | cut -c1-160
f=$(ls $out/*/*kernel_stats.csv | head -1)
python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
