#!/bin/bash
# Runs form of the direct kernel (kernel mode 14) against the automatic choice, same box: bash profiles/runs_vs_auto.sh <tag> "<configs>" "<frames>"
tag=$1; configs=$2; frames=$3
for c in $configs; do for f in $frames; do
  bash profiles/bench_json.sh $tag ${c}_auto_f$f --config $c --frames $f --steps 30 --warmup 5
  bash profiles/bench_json.sh $tag ${c}_runs_f$f --config $c --frames $f --steps 30 --warmup 5 --kernel-mode 14
done; done
