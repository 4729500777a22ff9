#!/bin/bash
# Rates of the drifting-plan configurations under the automatic rules (runs form of the direct kernel): bash profiles/runs_final.sh <tag>
tag=$1
for c in 'N15T8_f256 --config N15T8 --frames 256' 'N15T8_f64 --config N15T8 --frames 64' 'N15T4_f256 --config N15T4 --frames 256' \
         'N15T4_f64 --config N15T4 --frames 64' 'N3T4_f128 --config N3T4 --frames 128' 'N3T8_f128 --config N3T8 --frames 128' \
         'N480T4_f128 --config N480T4 --frames 128' 'N480T6_f128 --config N480T6 --frames 128' 'N25T6_f128 --config N25T6 --frames 128' \
         'N480T4_f32 --config N480T4 --frames 32' 'N15T4_f1 --config N15T4 --frames 1 --steps 50 --warmup 5' \
         'N15T4_f4 --config N15T4 --frames 4 --steps 50 --warmup 5' 'N15T4_f16 --config N15T4 --frames 16 --steps 50 --warmup 5' \
         'N15T8_f1 --config N15T8 --frames 1 --steps 50 --warmup 5' 'N15T8_f4 --config N15T8 --frames 4 --steps 50 --warmup 5' \
         'N15T8_f16 --config N15T8 --frames 16 --steps 50 --warmup 5'; do
  l=${c%% *}; bash profiles/bench_json.sh $tag $l ${c#* }
done
