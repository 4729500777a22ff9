#!/bin/bash
# SQ counter passes for one bench configuration (run through gpurun from the repo root):
#   profiles/pmc_sq.sh <tag> <config> [extra bench args]
# Each --pmc pass runs alone with --kernel-trace (never together with other trace domains).
tag=$1; cfg=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${tag}_pmc_$cfg
mkdir -p $out
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/p1 -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p1.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/p2 -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p2.log 2>&1
python profiles/pmc_summary.py $out > $out/summary.json 2>$out/summary.err
cat $out/summary.json
