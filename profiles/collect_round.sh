#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   profiles/collect_round.sh <tag> [round-dir]
# kernel-trace stats for C2/C3/C4 and two frame-pair configurations (A137, N15) with the DEFAULT bench command, FETCH_SIZE /
# WRITE_SIZE passes (separate --pmc runs, each alone with --kernel-trace) for the same five, and one bench line per configuration.  Raw output lands in
# gpurun_out/<tag>_*; profiles/summarize.py condenses it into profiles/<round-dir>/ (copied to gpurun_out/ as well).
tag=${1:-r3x}
export JINC_PROFILE_DIR=${2:-round3}
part=${3:-all}   # profiles | lines | all  (one gpurun call may run 1200 s at most: the two halves fit, the whole does not)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$JINC_PROFILE_DIR
if [ "$part" != lines ]; then
for c in ${JINC_PROFILE_CONFIGS:-C2 C3 C4 A137 N15}; do   # (JINC_PROFILE_CONFIGS / JINC_LINE_CONFIGS: other lists for partial re-collections)
  export JINC_FRAMES_PER_LAUNCH=$(python -c "import bench; print(bench.CONFIGS['$c'][6])")
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_$c -- python bench.py --no-clock-sampler --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/${tag}_stats_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch_$c -- python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-clock-sampler > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write_$c -- python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-clock-sampler > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_clock_$c -- python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-clock-sampler > /dev/null 2>&1
  python profiles/summarize.py ${tag}_$(echo $c | tr A-Z a-z) gpurun_out/${tag}_stats_$c gpurun_out/${tag}_fetch_$c gpurun_out/${tag}_write_$c gpurun_out/${tag}_clock_$c > /dev/null 2>&1
  grep "^{" gpurun_out/${tag}_stats_$c.log | tail -1 > profiles/$JINC_PROFILE_DIR/${tag}_stats_bench_$c.json
done
cp profiles/$JINC_PROFILE_DIR/${tag}_* gpurun_out/$JINC_PROFILE_DIR/ 2>/dev/null
fi
[ "$part" = profiles ] && exit 0
for c in ${JINC_LINE_CONFIGS:-C1 C2 C3 C4 C2T4 C2YUV C2H C2HT4 C2F N15 N3 U43 N480 N15T4 D23 D12 D12H D12F D13 D12T4 D12T8 D169 T6 T16 N15T8 A137 A137L32 A137L16 A137L4 A1875 D169L16 N3T4 N3T8 N480T4 N480T6 N25T6}; do
  timeout 120 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > gpurun_out/$JINC_PROFILE_DIR/${tag}_bench_$c.json
  python profiles/bench_line.py < gpurun_out/$JINC_PROFILE_DIR/${tag}_bench_$c.json
done
[ -n "$JINC_LINE_CONFIGS" ] && exit 0
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/$JINC_PROFILE_DIR/${tag}_bench_default.json
python profiles/bench_line.py < gpurun_out/$JINC_PROFILE_DIR/${tag}_bench_default.json
