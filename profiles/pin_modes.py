#!/usr/bin/env python3
"""Host-to-host rate of one instance under the ways of treating the caller's buffers (jinc_filter_set_pipeline's
register_host_buffers), C2 and two other plans, look-ahead 128 / 16 / 2.  Written in round 6 for THREE modes -- 0 pageable, 1 pinned
while the frame is in flight, 2 pinned once and cached by address -- and run on the build that had all three
(profiles/round6/pin_modes.log).  Mode 1 was withdrawn (registration churn: profiles/round6/README.md); on today's library every
non-zero value means "cached", so the script's mode 1 rows repeat mode 2.  Output: one JSON line per point."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
for cfg in (sys.argv[1:] or ["C2", "A137", "N15T8"]):
    for depth in (128, 16, 2):
        for mode in (2, 1, 0):
            rec = bench.e2e_record(pkg, cfg, depth=depth, seconds=1.5, pin_mode=mode)
            print(json.dumps({"config": cfg, "depth": depth, "pin_mode": mode, "frames_per_s": rec["frames_per_s"], "host_GB_per_s": rec["host_GB_per_s"],
                              "frames_per_launch": rec["frames_per_launch"]}), flush=True)
