# frame-lane kernel (kernel mode 11) against the automatic choice, 64 frames per step:  bash profiles/fl_vs_auto.sh [configs...]
ulimit -c 0
run() { timeout 90 python bench.py --config $1 --frames $3 --no-cpu-baseline --kernel-mode $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', 'mode=$2', 'frames=$3', d['value'], d['roofline']['valu_frac'], d['roofline']['kernel'])"; }
for c in ${@:-N15 N3 U43 N480 N15T4 D23 D12 D13 T6 T16 C1 C2 C3 C4}; do
  f=64; [ $c = C4 ] && f=16; [ $c = C3 ] && f=64
  run $c 0 $f; run $c 11 $f
done
