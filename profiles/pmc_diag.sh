#!/bin/bash
# Stall diagnosis passes for one bench configuration (run through gpurun from the repo root):
#   [JINC_... env] profiles/pmc_diag.sh <tag> <config> [extra bench args]
# occupancy (SQ_LEVEL_WAVES / SQ_BUSY_CYCLES), instruction fetch, scalar / vector memory latency
# (SQ_INST_LEVEL_x / SQ_INSTS_x), TA fifo back-pressure.  Each --pmc pass runs alone with --kernel-trace.
tag=$1; cfg=$2; shift 2
ulimit -c 0
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R" || exit 1
out=gpurun_out/${tag}_diag_$cfg
mkdir -p $out
run() { n=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$n -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $out/$n.log 2>&1; }
EXTRA="$*"
run p1 SQ_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run p2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM
run p3 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
run p4 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
python profiles/pmc_summary.py $out > $out/summary.json 2>$out/summary.err
python - <<PY
import json
d=json.load(open("$out/summary.json"))
k=d.get("ewa_direct_kernel") or max(d.values(), key=lambda v: v.get("SQ_WAVE_CYCLES",0))
print("$tag $cfg", {a: round(b/1e6,3) for a,b in sorted(k.items())})
PY
