#!/bin/bash
# Memory-side counter passes for one bench configuration (run through gpurun from the repo root):
#   profiles/pmc_mem.sh <tag> <config> [extra bench args]
# L2 requests / hits / misses, HBM bytes (FETCH_SIZE, WRITE_SIZE), vector-memory instruction counts and TA busy.
tag=$1; cfg=$2; shift 2
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_mem_$cfg
mkdir -p $out
timeout 240 rocprofv3 --kernel-trace --pmc TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p1 -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p1.log 2>&1
# (one counter per pass: together the two exceed what the hardware collects at once on gfx950 -- rocprofiler error 38)
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p2 -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p2.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p2w -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p2w.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES TA_TA_BUSY_sum TA_BUSY_avr --output-format csv -d $out/p3 -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p3.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d $out/p4 -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p4.log 2>&1
python profiles/pmc_summary.py $out > $out/summary.json 2>$out/summary.err
cat $out/summary.json
tail -3 $out/p3.log $out/p4.log
