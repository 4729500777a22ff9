ulimit -c 0
run() { timeout 60 python bench.py --config $2 --frames 64 --no-cpu-baseline --kernel-mode 11 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$2', '$1', d['value'], d['roofline']['valu_frac'])"; }
for c in A137 A1875; do
JINC_FL_VARIANT=0 run base $c
for n in 1 2 4 8; do JINC_FL_VARIANT=$((16 + n*256)) run stagger$n $c; done
done
