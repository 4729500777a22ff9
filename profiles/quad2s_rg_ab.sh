cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  for k in "quad2s=0" "quad2s=1" "quad2s=0 --knob quad_rg=4" "quad2s=1 --knob quad_rg=4"; do
    line=$(timeout 120 python bench.py --config C2 --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --no-clock-sampler --knob $k 2>/dev/null | tail -1)
    python -c "
import json,sys
d=json.loads(sys.argv[2]); print(sys.argv[1].ljust(30), round(d['value']/1e3,1), d['self_check'], d['roofline']['kernel'])" "$k" "$line"
  done
done
