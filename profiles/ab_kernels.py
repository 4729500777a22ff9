#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process on ONE device (cdna_hip_programming.md
rule 24): for every variant, `rounds` rounds of `steps` steps each, round-robin over variants.
Prints median / min of the whole-step wall time (device-synchronised, `steps` steps per sample)
and of the periodic-kernel time reported by the library's own events.

usage: python profiles/ab_kernels.py [--config C2] [--frames 64] [--rounds 7] [--steps 5] variant [variant ...]
variant = <kernel_mode>[o|s]   kernel_mode 0 auto, 1 gather only, 3 row-streamed periodic;
                               o = border kernel overlapped on a side stream, s = serial, neither = automatic
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: its HIP runtime must be the one in the process)

import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    pkg = entry.load_package()
    torch.cuda.set_device(0)
    B = a.frames or bench.CONFIGS[a.config][6]
    flt, step, stream, fmt, ddims = bench.make_workload(pkg, torch, a.config, B, 0, 12345)
    _, sw, sh, dw, dh, kw, _ = bench.CONFIGS[a.config]
    res = {v: {"step_ms": [], "periodic_ms": [], "gather_ms": []} for v in a.variants}
    for rnd in range(a.rounds + 1):
        for v in a.variants:
            mode = int(v.rstrip("os"))
            flt.set_kernel_mode(mode)
            flt.set_border_overlap(False if v.endswith("s") else (True if v.endswith("o") else None))
            flt.set_profiling(True)
            flt.kernel_times()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()  # device-wide: the library may run on its own streams
            wall_ms = (time.perf_counter() - t0) * 1e3
            pm, pn, gm, gn = flt.kernel_times()
            flt.set_profiling(False)
            if rnd == 0:
                continue  # warm-up round
            res[v]["step_ms"].append(wall_ms / a.steps)
            res[v]["periodic_ms"].append(pm / a.steps)
            res[v]["gather_ms"].append(gm / a.steps)
    pix = dw * dh * B
    for v, r in res.items():
        med = statistics.median(r["step_ms"])
        print(json.dumps({"config": a.config, "frames": B, "variant": v,
                          "step_ms_median": round(med, 4), "step_ms_min": round(min(r["step_ms"]), 4),
                          "periodic_ms_median": round(statistics.median(r["periodic_ms"]), 4),
                          "gather_ms_median": round(statistics.median(r["gather_ms"]), 4),
                          "Gpix_per_s_median": round(pix / med / 1e6, 1)}))


if __name__ == "__main__":
    main()
