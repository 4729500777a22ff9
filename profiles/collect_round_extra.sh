cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/round2
for c in N15 N15T4 D169 N15T8; do
  timeout 120 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/round2/r2v_bench_$c.json
  python profiles/bench_line.py < gpurun_out/round2/r2v_bench_$c.json
done
bash profiles/pmc_fl.sh r2v64 A137 --kernel-mode 11 > gpurun_out/r2v_pmc64.log 2>&1
bash profiles/pmc_fl.sh r2vpair A137 > gpurun_out/r2v_pmcpair.log 2>&1
cp gpurun_out/r2v64_fl_A137/summary.json gpurun_out/round2/pmc_sqc_framelane64_A137.json
cp gpurun_out/r2vpair_fl_A137/summary.json gpurun_out/round2/pmc_sqc_framepair_A137.json
tail -2 gpurun_out/r2v_pmc64.log | cut -c1-1500; tail -2 gpurun_out/r2v_pmcpair.log | cut -c1-1500
