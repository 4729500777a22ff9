ulimit -c 0
run() { timeout 60 python bench.py --config $2 --frames 64 --no-cpu-baseline --kernel-mode 11 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$2', '$1', d['value'], d['roofline']['valu_frac'], d['roofline']['kernel'])"; }
for c in A137 A1875 N15 D169 C2 N480; do JINC_FL_PERSISTENT=0 run plain $c; run persistent $c; done
