#!/usr/bin/env python3
"""Rebuilds profiles/traffic.json (what bench.py copies into roofline.traffic) from the PMC summaries of a round.

usage: make_traffic.py <round-dir> <tag> [<tag> ...]   e.g.  make_traffic.py round3 r3w r3y  -> reads profiles/round3/r3w_c{2,3,4}.json ...;
       a later tag replaces the configurations it holds

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB x 1024, separate --pmc passes).  The x2 on the read side is the
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the requests are 128 B); round 1
published the raw figure, which came out BELOW the compulsory source bytes (70.8 MB fetched for 132.7 MB of distinct
source samples per 64 C2 frames) -- the raw counter under-reports here too, so the correction applies.  The raw sum is
kept next to it.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
rdir, tags = sys.argv[1], sys.argv[2:]
tag = "{" + ",".join(tags) + "}"

out = {"_comment": "HBM bytes per launch of the dominant (interior) kernel from rocprofv3 PMC passes: 2 x FETCH_SIZE + WRITE_SIZE "
                   "(separate --pmc runs, KiB*1024; x2 = gfx950 FETCH_SIZE correction of the micro-architecture guide). Sources: "
                   f"profiles/{rdir}/{tag}_c*.json (profiles/collect_round.sh). For C3/C4 a step has three launches (one per "
                   "plane); the figure is the mean over them, like roofline.algorithmic_bytes_per_launch. bench.py copies the "
                   "value for its config into roofline.traffic, scaled to the frames per launch of the run."}
for cfg in ("C2", "C3", "C4", "A137", "N15", "N15T8", "N15T4"):
    paths = [os.path.join(HERE, rdir, f"{t}_{cfg.lower()}.json") for t in tags]
    paths = [q for q in paths if os.path.exists(q)]
    if not paths:
        continue
    path = paths[-1]
    d = json.load(open(path))
    cands = [(k, v) for k, v in d["kernels"].items() if "hbm_bytes_per_launch_raw" in v]
    # the interior kernel, not a border kernel that runs beside it for as long (1.5x with tap 4: the gather kernel over the border)
    interior = [kv for kv in cands if not kv[0].startswith(("ewa_gather_kernel", "ewa_colstrip_kernel"))]
    name, e = max(interior or cands, key=lambda kv: kv[1]["avg_ns"] * kv[1]["calls"])
    out[cfg] = {"hbm_bytes_per_launch": int(round(2 * e["FETCH_SIZE_bytes_mean"] + e["WRITE_SIZE_bytes_mean"])),
                "hbm_bytes_per_launch_raw": int(round(e["hbm_bytes_per_launch_raw"])),
                "frames_per_launch": d.get("frames_per_launch"),
                "kernel": name, "fetch_bytes_raw": int(round(e["FETCH_SIZE_bytes_mean"])), "write_bytes": int(round(e["WRITE_SIZE_bytes_mean"])),
                "avg_ns_under_rocprof": e["avg_ns"], "calls": e["calls"]}
    if "effective_clock_ghz" in e:   # GRBM_GUI_ACTIVE / 8 / dispatch duration, mean over the launches of that pass
        out[cfg]["effective_clock_ghz"] = round(e["effective_clock_ghz"], 4)
        out[cfg]["clock_pass_kernel_ms"] = round(e["clock_pass_kernel_ns"] / 1e6, 4)
json.dump(out, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
