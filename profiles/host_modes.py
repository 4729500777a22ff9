#!/usr/bin/env python3
"""Host-to-host rate of one instance under the three ways of treating the caller's planes (register_host_buffers of
jinc_filter_set_pipeline): 0 pageable planes copied by the CPU through the library's own pinned buffers (the default from round 6 on),
3 pageable planes handed to the HIP runtime as they are (the default of rounds 1 - 5), 2 registered once and cached by address.
Frames in flight 1 (submit + wait: what jinc_filter_get_frame does) ... 128.  One JSON line per point
(profiles/round6/host_modes.log)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
for cfg in (sys.argv[1:] or ["C2", "A137", "C1"]):
    for depth in (1, 2, 8, 16, 128):
        for mode in (0, 3, 2):
            rec = bench.e2e_record(pkg, cfg, depth=depth, seconds=1.5, pin_mode=mode)
            print(json.dumps({"config": cfg, "depth": rec["frames_in_flight"], "mode": mode, "frames_per_s": rec["frames_per_s"],
                              "host_GB_per_s": rec["host_GB_per_s"], "frames_per_launch": rec["frames_per_launch"]}), flush=True)
