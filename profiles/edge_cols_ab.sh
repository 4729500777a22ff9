#!/bin/bash
# A/B of the border columns inside ewa_periodic_quad2_kernel's edge tiles (knob edge_cols) over batch sizes: which calls gain.
#   gpurun -- 'bash profiles/edge_cols_ab.sh > gpurun_out/edge_cols_sweep.log'
cd "$GRAFT_REPO_ROOT" || exit 1
for spec in "C2 16" "C2 64" "C2 256" "C2YUV 16" "C2YUV 64" "C2YUV 256" "C2H 32" "C2H 128" "C1 64" "C1 256"; do
  set -- $spec
  for k in 1 0 1 0; do
    echo -n "frames $2 edge_cols=$k  "
    timeout -k 10 120 python bench.py --config $1 --frames $2 --steps 20 --warmup 3 --knob edge_cols=$k --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python profiles/bench_line.py
  done
done
