#!/bin/bash
# Coefficient traffic before / after: the same unstructured workload (A137, 64 frames per launch) on the
# gather kernel (kernel mode 1: one lane per pixel, per-lane coefficient fetches) and on the frame-lane kernel (automatic
# choice: lanes = frames, scalar coefficient loads).  L2 read requests (TCC_READ), scalar / vector memory instructions.
#   profiles/pmc_coeff_traffic.sh <tag>
tag=${1:-r2t}
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for mode in 1 0; do
  out=gpurun_out/${tag}_coeff_mode$mode
  mkdir -p $out
  timeout 240 rocprofv3 --kernel-trace --pmc TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p1 -- python bench.py --no-clock-sampler --no-e2e --config A137 --frames 64 --steps 3 --warmup 1 --no-cpu-baseline --kernel-mode $mode > $out/p1.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/p2 -- python bench.py --no-clock-sampler --no-e2e --config A137 --frames 64 --steps 3 --warmup 1 --no-cpu-baseline --kernel-mode $mode > $out/p2.log 2>&1
  python profiles/pmc_summary.py $out > $out/summary.json 2>/dev/null
  cat $out/summary.json
done
