# The direct kernel (kernel mode 9: wherever it applies) against the automatic choice, whole-step rate.
ulimit -c 0
run() { timeout 120 python bench.py --config $2 --no-cpu-baseline $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$2', '$1', round(d['value'],1), d['roofline']['valu_frac'], d['roofline']['kernel'])"; }
for c in ${CONFIGS:-C3 C4 T6 C2}; do run auto $c; run direct $c "--kernel-mode 9"; done
