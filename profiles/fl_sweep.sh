ulimit -c 0
run() { timeout 60 python bench.py --config $1 --no-cpu-baseline --kernel-mode 11 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', '$2', d['value'], d['roofline']['valu_frac'])"; }
for c in A137 D169 N15T4; do
  for v in 0 1; do for kb in 32 48 64; do JINC_FL_VARIANT=$v JINC_FL_LDS_KB=$kb run $c "variant=$v lds=$kb"; done; done
  JINC_FL_THREADS=256 run $c "threads=256"
done
