#!/bin/bash
# Vector-memory issue counters of one bench configuration (are the waves held at their stores?):
#   profiles/pmc_vmem.sh <tag> <config> [extra bench args]      (through gpurun from the repo root; one --pmc pass, alone with --kernel-trace)
tag=$1; cfg=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_vmem_$cfg
mkdir -p $out
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p1 -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p1.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/p2 -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p2.log 2>&1
python profiles/pmc_summary.py $out > $out/summary.json 2>$out/summary.err
