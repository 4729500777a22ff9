# A/B of the direct kernel's chain shapes (JINC_DIRECT_SHAPE = DirectShape index of kernel_direct.hip; unset: launcher's choice)
ulimit -c 0
run() { timeout 90 python bench.py --config $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$2', '$1', round(d['value'],1), d['roofline']['valu_frac'], d['roofline']['kernel'])"; }
for c in ${CONFIGS:-D12 D12H D12F D23 D13 T16}; do for s in ${SHAPES:-0 1 2 3}; do JINC_DIRECT_SHAPE=$s run shape$s $c; done; run auto $c; done
