#!/usr/bin/env python3
"""Small pageable frames: source planes copied at submit (STAGE_DEFER_KB = 0) against with their group at its launch (1536, the default;
4096 also takes C2's 2 MB frames), 8 / 16 / 128 frames in flight, two rounds alternating."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
for cfg in (sys.argv[1:] or ["A137", "C1", "C2"]):
    for depth in (8, 16, 128):
        for rnd in range(2):
            for kb in (0, 1536, 4096):
                pkg.set_knob("stage_defer_kb", kb)
                rec = bench.e2e_record(pkg, cfg, depth=depth, seconds=1.0, pin_mode=0)
                print(json.dumps({"config": cfg, "depth": rec["frames_in_flight"], "stage_defer_kb": kb, "round": rnd, "frames_per_s": rec["frames_per_s"]}), flush=True)
