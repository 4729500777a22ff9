#!/usr/bin/env python3
"""PCIe-inclusive rate of the library's multi-device sharder on C5 (BASELINE.json configs[4]): 512 distinct 1080p -> 4K
Y8 tap-3 frames, host planes in, host planes out, through jinc_batch_process on the visible devices.
NOT bench.py's `value` (device-resident frames); DESIGN.md quotes it next to it.

usage: python profiles/e2e_batch_c5.py [--frames 512] [--streams 2 3 4] [--register 0 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--streams", type=int, nargs="+", default=[4, 16, 64], help="frames in flight per device")
    ap.add_argument("--register", type=int, nargs="+", default=[0, 1])
    a = ap.parse_args()
    pkg = entry.load_package()
    fmt = pkg.FORMATS["Y8"]
    sw, sh, tw, th = 1920, 1080, 3840, 2160
    rng = np.random.default_rng(12345)
    frames = []
    for _ in range(a.frames):
        p = pkg.alloc_plane(sw, sh, np.uint8)
        p[:] = rng.integers(0, 256, p.shape, dtype=np.uint8)
        frames.append([p])
    outs = [[pkg.alloc_plane(tw, th, np.uint8)] for _ in range(a.frames)]
    for reg in a.register:
        for st in a.streams:
            b = pkg.Batch(fmt, sw, sh, tw, th, ndevices=0, streams=st, register_host_buffers=bool(reg), tap=3)
            b.process(frames[:8], outs[:8])  # warm-up (plan upload, slot allocation, first registrations)
            t0 = time.perf_counter()
            b.process(frames, outs)
            el = time.perf_counter() - t0
            print(json.dumps({"workload": f"C5: {a.frames} frames 1920x1080->3840x2160 Y8 tap=3, host to host", "devices": b.devices,
                              "streams_per_device": st, "registered_host_buffers": bool(reg), "seconds": round(el, 4),
                              "frames_per_s": round(a.frames / el, 1), "Mpix_per_s": round(a.frames * tw * th / el / 1e6, 1),
                              "host_GB_per_s": round(a.frames * (sw * sh + tw * th) / el / 1e9, 2)}), flush=True)
            b.close()


if __name__ == "__main__":
    main()
