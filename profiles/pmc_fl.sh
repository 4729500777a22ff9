#!/bin/bash
# Frame-lane stall diagnosis: pmc_diag passes + scalar-cache / store-path counters.  gpurun -- bash profiles/pmc_fl.sh <tag> <config>
tag=$1; cfg=$2; shift 2
ulimit -c 0
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R" || exit 1
out=gpurun_out/${tag}_fl_$cfg
mkdir -p $out
run() { n=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$n -- python bench.py --no-clock-sampler --no-e2e --config $cfg --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $out/$n.log 2>&1 || echo "pass $n failed: $(tail -2 $out/$n.log)"; }
EXTRA="$*"
run p1 SQ_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run p2 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
run p3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS
run p4 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ SQC_DCACHE_INPUT_VALID_READYB SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL
run p5 TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
run p6 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_REQ_sum
run p7 GRBM_GUI_ACTIVE GRBM_COUNT
python profiles/pmc_summary.py $out > $out/summary.json 2>$out/summary.err
python - <<PY
import json
d=json.load(open("$out/summary.json"))
for name,k in d.items():
    if k.get("SQ_WAVE_CYCLES",0) or k.get("SQ_INSTS_SMEM",0):
        print("$tag $cfg", name, {a: round(b/1e6,3) for a,b in sorted(k.items())})
PY
