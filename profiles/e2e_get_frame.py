#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in entry point jinc_filter_get_frame (host planes in, host planes out),
one filter instance per host thread (AviSynth MT_MULTI_INSTANCE).  This is NOT bench.py's `value`
(device-resident frames); DESIGN.md quotes it next to it.

usage: python profiles/e2e_get_frame.py [--config C2] [--threads 1 2 4 8] [--seconds 3]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--pipeline", type=int, nargs="*", default=[], help="also time ONE host thread with these pipeline depths")
    ap.add_argument("--group", type=int, default=0, help="frames coalesced per launch (0: automatic = depth / 2)")
    ap.add_argument("--registered-only", action="store_true")
    ap.add_argument("--pin-mode", type=int, default=-1, help="-1: the instance's default (0); 0 / 2 / 3: set_pipeline(1, mode) -- through the library's pinned buffers / cached registrations / handed to the runtime")
    a = ap.parse_args()
    pkg = entry.load_package()
    fmt_name, sw, sh, dw, dh, kw, _ = bench.CONFIGS[a.config]
    fmt = pkg.FORMATS[fmt_name]
    rng = np.random.default_rng(1)
    for nt in a.threads:
        counts = [0] * nt
        stop = threading.Event()

        def work(k):
            f = pkg.Filter(fmt, sw, sh, dw, dh, device=0, **kw)
            if a.pin_mode >= 0:
                f.set_pipeline(1, a.pin_mode)
            src = []
            for (w, h) in fmt.plane_dims(sw, sh):
                p = pkg.alloc_plane(w, h, fmt.dtype)
                p[:] = (rng.random(p.shape) * (255 if fmt.sample_bytes == 1 else 1)).astype(fmt.dtype)
                src.append(p)
            f.get_frame(src)  # warm-up (allocates device staging)
            while not stop.is_set():
                f.get_frame(src)
                counts[k] += 1
            f.close()

        ts = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        time.sleep(a.seconds)
        stop.set()
        [t.join() for t in ts]
        el = time.perf_counter() - t0
        fps = sum(counts) / el
        print(json.dumps({"config": a.config, "threads": nt, "frames_per_s": round(fps, 1),
                          "Mpix_per_s": round(fps * dw * dh / 1e6, 1),
                          "host_GB_per_s": round(fps * bench.algorithmic_bytes_per_frame(fmt, sw, sh, dw, dh) / 1e9, 2)}))
    for depth in a.pipeline:
        for register in ((True,) if a.registered_only else (False, True)):
            pipeline_run(pkg, fmt, sw, sh, dw, dh, kw, depth, register, a.seconds, a.config, min(a.group, depth))


def pipeline_run(pkg, fmt, sw, sh, dw, dh, kw, depth, register, seconds, config, group=0):
    f = pkg.Filter(fmt, sw, sh, dw, dh, device=0, **kw)
    f.set_pipeline(depth, register, group)
    rng = np.random.default_rng(2)
    nbuf = depth + 1
    srcs, dsts = [], []
    for _ in range(nbuf):
        s = []
        for (w, h) in fmt.plane_dims(sw, sh):
            p = pkg.alloc_plane(w, h, fmt.dtype)
            p[:] = (rng.random(p.shape) * (255 if fmt.sample_bytes == 1 else 1)).astype(fmt.dtype)
            s.append(p)
        srcs.append(s)
        dsts.append([pkg.alloc_plane(w, h, fmt.dtype) for (w, h) in fmt.plane_dims(dw, dh)])
    tickets = []
    for k in range(nbuf):  # warm-up: allocations, registration
        tickets.append(f.submit(srcs[k % nbuf], dsts[k % nbuf]))
        if len(tickets) >= depth:
            f.wait(tickets.pop(0))
    while tickets:
        f.wait(tickets.pop(0))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        tickets.append(f.submit(srcs[n % nbuf], dsts[n % nbuf]))
        if len(tickets) >= depth:
            f.wait(tickets.pop(0))
        n += 1
    while tickets:
        f.wait(tickets.pop(0))
    el = time.perf_counter() - t0
    fps = n / el
    print(json.dumps({"config": config, "threads": 1, "pipeline_depth": depth, "frames_per_launch": f.pipeline_group,
                      "kernel": f.last_kernel(0), "registered_host_buffers": register,
                      "frames_per_s": round(fps, 1), "Mpix_per_s": round(fps * dw * dh / 1e6, 1),
                      "host_GB_per_s": round(fps * bench.algorithmic_bytes_per_frame(fmt, sw, sh, dw, dh) / 1e9, 2)}))
    f.close()


if __name__ == "__main__":
    main()
