#!/usr/bin/env python3
"""Rebuilds profiles/traffic.json (what bench.py copies into roofline.traffic) from the PMC summaries of a round.

usage: make_traffic.py <tag>        e.g.  make_traffic.py r1k   -> reads profiles/round1/<tag>_c{2,3,4}.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402  (CONFIGS: default frames per launch)

out = {"_comment": "HBM bytes per launch of the dominant (interior) kernel from rocprofv3 PMC passes (FETCH_SIZE and "
                   "WRITE_SIZE in separate runs, KiB*1024, raw: the gfx950 x2 FETCH correction is calibrated for 16-B/lane "
                   "streaming reads only; these kernels stage with 1/2/4-B-per-lane loads). Sources: "
                   f"profiles/round1/{tag}_c*.json (profiles/collect_round.sh). For C3/C4 a step has three launches (one per "
                   "plane); the figure is the mean over them, like roofline.algorithmic_bytes_per_launch. bench.py copies "
                   "the value for its config into roofline.traffic, scaled to the frames per launch of the run."}
for cfg in ("C2", "C3", "C4"):
    d = json.load(open(os.path.join(HERE, "round1", f"{tag}_{cfg.lower()}.json")))
    name, e = max(((k, v) for k, v in d["kernels"].items() if "hbm_bytes_per_launch_raw" in v), key=lambda kv: kv[1]["avg_ns"] * kv[1]["calls"])
    out[cfg] = {"hbm_bytes_per_launch": int(round(e["hbm_bytes_per_launch_raw"])), "frames_per_launch": bench.CONFIGS[cfg][6],
                "kernel": name, "fetch_bytes": int(round(e["FETCH_SIZE_bytes_mean"])), "write_bytes": int(round(e["WRITE_SIZE_bytes_mean"])),
                "avg_ns_under_rocprof": e["avg_ns"], "calls": e["calls"]}
json.dump(out, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
