#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   profiles/collect_round.sh <tag>
# kernel-trace stats for C2/C3/C4, FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs) for C2/C3/C4, and one bench
# line per configuration.  Raw output lands in gpurun_out/<tag>_*; profiles/summarize.py condenses it.
tag=${1:-r1x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for c in C2 C3 C4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_$c -- python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_stats_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch_$c -- python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write_$c -- python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python profiles/summarize.py ${tag}_$(echo $c | tr A-Z a-z) gpurun_out/${tag}_stats_$c gpurun_out/${tag}_fetch_$c gpurun_out/${tag}_write_$c > /dev/null 2>&1
  cp profiles/${tag}_$(echo $c | tr A-Z a-z).json gpurun_out/ 2>/dev/null
done
for c in C1 C2 C3 C4 N15 N3 U43 N480 N15T4 D23 D12 D13 D169 T6 T16 N15T8 A137 A1875; do
  python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$c.json
  python profiles/bench_line.py < gpurun_out/${tag}_bench_$c.json
done
python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_default.json
python profiles/bench_line.py < gpurun_out/${tag}_bench_default.json
