import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith("{"): continue
    d=json.loads(l); r=d["roofline"]
    print("%-44s %9.1f Mpix/s %-28s valu %.3f kern %.4f ms border %s" % (d["config"]["workload"], d["value"], d["config"]["kernel"], r["valu_frac"], r["kernel_ms_per_launch"], r.get("step_minus_interior_ms", r.get("border_kernel_ms_per_step"))))
