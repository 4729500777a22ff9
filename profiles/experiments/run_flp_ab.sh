cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_framelane_pair.py tests/test_framelane.py -x -q > gpurun_out/flp_test.log 2>&1; tail -3 gpurun_out/flp_test.log
CONFIGS="A137 N15 C2" bash profiles/flp_bench.sh 256 > gpurun_out/flp_bench.log 2>&1; cat gpurun_out/flp_bench.log
echo "--- 512-thread 64-frame form"
JINC_FL_1K=0 CONFIGS="A137 N15 C2" bash profiles/flp_bench.sh 256 2>&1 | grep "64-frame"
