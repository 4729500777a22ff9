#!/usr/bin/env python3
"""How the own AVX2 / AVX-512-order CPU code (oracle/simd_avx2.c, simd_avx512.c: bench.py's cpu_baseline) compares with the
reference's own resize_plane_avx2 / _avx512 in the BUILD CONTAINER, one thread, C2 (VERDICT r5, weak 2 / Next 2).

The reference's figures are the ones recorded from executing the reference here -- SURVEY.md section 6 (survey-time probe) and
VERDICT.md of round 5 (judge-side run, "best of 5 alternating runs") -- because this repository may not build the reference
(no avisynth_c.h in the image; stand-in headers are ruled out).  The port is timed by this script: best of N alternating runs
per path on preallocated planes, one thread.  The container is a shared host (the opt=0 figure of the SAME code moves between
16 and 25 Mpix/s from minute to minute), so each figure is also given relative to the opt=0 port of its own run, next to the
reference's own opt2/opt0 and opt3/opt0 ratios: that quotient does not move with the host's load.

Writes profiles/cpu_port_build_container.json; bench.py copies it into cpu_baseline.port_vs_reference_build_container.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REFERENCE = {  # Mpix/s, one thread, C2 (1920x1080 -> 3840x2160 Y8 tap 3), this container
    "SURVEY.md section 6": {"opt0": 26.45, "opt2_avx2": 97.3, "opt3_avx512": 97.9, "note": "opt0 given as 23.9-29.0; midpoint used for the ratios"},
    "VERDICT r5 (judge-side run)": {"opt0": 16.1, "opt2_avx2": 112.2, "opt3_avx512": 102.1},
}


def main(rounds=12):
    fmt = O.FORMATS["Y8"]
    sw, sh, tw, th = 1920, 1080, 3840, 2160
    f = O.OracleFilter(fmt, sw, sh, tw, th, tap=3)
    src = O.lcg_frame(fmt, sw, sh)
    dst = [O.alloc_plane(w, h, fmt.dtype) for (w, h) in f.out_dims()]
    t = f.table_for_plane(0)
    paths = {"opt0_port": lambda: t.resize(src[0], dst[0], f.peak, 1),
             "avx2_order_port": lambda: t.resize_simd(2, src[0], dst[0], 0.0, 1, True),
             "avx512_order_port": (lambda: t.resize_simd(3, src[0], dst[0], 0.0, 1, False, True)) if O.lib().oracle_avx512_available() else None}
    best = {k: 1e9 for k, v in paths.items() if v}
    per_round = []
    for _ in range(rounds):  # alternating, so that a quiet or a busy minute of the host is seen by every path
        row = {}
        for k, fn in paths.items():
            if fn is None:
                continue
            t0 = time.perf_counter()
            fn()
            el = time.perf_counter() - t0
            best[k] = min(best[k], el)
            row[k] = round(tw * th / el / 1e6, 1)
        per_round.append(row)
    port = {k: round(tw * th / v / 1e6, 1) for k, v in best.items()}
    # per-round quotients against the same round's opt=0 figure (median over rounds)
    def med(xs):
        xs = sorted(xs)
        return xs[len(xs) // 2]
    quot = {k: round(med([r[k] / r["opt0_port"] for r in per_round]), 2) for k in port if k != "opt0_port"}
    out = {"config": "C2", "threads": 1, "unit": "Mpix/s", "host": "build container (8 vCPU, AVX-512 capable)",
           "port_best_of_%d_alternating_runs" % rounds: port, "port_over_its_own_opt0_median": quot,
           "reference_recorded": REFERENCE,
           "reference_over_its_own_opt0": {k: {"opt2_avx2": round(v["opt2_avx2"] / v["opt0"], 2), "opt3_avx512": round(v["opt3_avx512"] / v["opt0"], 2)}
                                           for k, v in REFERENCE.items()},
           "port_vs_reference": {k: {"avx2_order": round(port["avx2_order_port"] / v["opt2_avx2"], 2),
                                     "avx512_order": round(port.get("avx512_order_port", 0) / v["opt3_avx512"], 2) if "avx512_order_port" in port else None}
                                 for k, v in REFERENCE.items()},
           "per_round_Mpix_s": per_round}
    path = os.path.join(ROOT, "profiles", "cpu_port_build_container.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_round_Mpix_s"}, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
