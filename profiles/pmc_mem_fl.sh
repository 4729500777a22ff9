#!/bin/bash
# Memory-side counters of the frame-lane kernels: gpurun -- bash profiles/pmc_mem_fl.sh <tag> <config> [bench args]
tag=$1; cfg=$2; shift 2
ulimit -c 0
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R" || exit 1
out=gpurun_out/${tag}_mem_$cfg
mkdir -p $out
run() { n=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$n -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $out/$n.log 2>&1 || echo "pass $n failed: $(tail -2 $out/$n.log)"; }
EXTRA="$*"
run m1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run m2 TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum
run m3 TCC_WRITEBACK_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
run m4 TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
python profiles/pmc_summary.py $out > $out/summary.json 2>$out/summary.err
python - <<PY
import json
d=json.load(open("$out/summary.json"))
for name,k in d.items():
    if any(c.startswith("TCC") for c in k):
        print("$tag $cfg $EXTRA", name, {a: round(b/1e6,3) for a,b in sorted(k.items())})
PY
