ulimit -c 0
run() { timeout 60 python bench.py --config A137 --frames 64 --no-cpu-baseline --kernel-mode 11 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['roofline']['valu_frac'])"; }
JINC_FL_VARIANT=0 run base
JINC_FL_VARIANT=2 run nostore
JINC_FL_VARIANT=4 run nostage
JINC_FL_VARIANT=8 run sameset
JINC_FL_VARIANT=14 run all_off
