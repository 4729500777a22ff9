#!/bin/bash
# Same-box A/B of library builds with arbitrary bench arguments per case:
#   bash profiles/ab_libs_args.sh <tag> <rounds> "name=lib.so name=lib.so" "case1 args" "case2 args" ...   (case = "<label> <bench args>")
tag=$1; rounds=$2; libs=$3; shift 3
for r in $(seq 1 $rounds); do
  for cs in "$@"; do
    label=${cs%% *}; args=${cs#* }
    for nl in $libs; do
      n=${nl%%=*}; l=${nl#*=}
      JINC_LIB=$PWD/$l bash profiles/bench_json.sh $tag ${label}_${n}_$r $args
    done
  done
done
