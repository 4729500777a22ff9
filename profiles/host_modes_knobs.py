#!/usr/bin/env python3
"""The default host path (pageable planes through the library's pinned buffers) against its two knobs: copy lanes (COPY_THREADS,
1 = the calling thread only = what threads = 1 gives) and row bands (STAGE_BANDS, 1 = whole planes), C2, 1 / 2 / 8 frames in flight."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
for cfg in (sys.argv[1:] or ["C2"]):
    for depth in (1, 2, 8):
        for lanes, bands in ((4, 4), (4, 1), (4, 2), (4, 8), (1, 4), (2, 4), (3, 4), (6, 4), (8, 4)):
            pkg.set_knob("copy_threads", lanes)
            pkg.set_knob("stage_bands", bands)
            rec = bench.e2e_record(pkg, cfg, depth=depth, seconds=1.2, pin_mode=0)
            print(json.dumps({"config": cfg, "depth": rec["frames_in_flight"], "copy_lanes": lanes, "bands": bands, "frames_per_s": rec["frames_per_s"],
                              "host_GB_per_s": rec["host_GB_per_s"]}), flush=True)
