#!/usr/bin/env python3
"""Prints the Markdown table of BASELINE.md section 6 from the bench lines of a round: make_tables.py round2 r2v"""
import glob, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
rdir, tag = sys.argv[1], sys.argv[2]
order = ["C2", "C1", "C3", "C4", "C2T4", "C2YUV", "C2H", "C2HT4", "C2F", "A137", "A137L32", "A137L16", "A137L4", "A1875", "D169L16", "N15", "N3", "U43", "N480", "N15T4", "N15T8", "D169", "D12", "D23", "D13", "D12H", "D12F",
         "D12T4", "D12T8", "T6", "T16", "N3T4", "N3T8", "N480T4", "N480T6", "N25T6"]
new = False
for c in order:
    p = os.path.join(HERE, rdir, f"{tag}_bench_{c}.json")
    if os.path.exists(p) and {"reference_equivalent_rate_vs_peak_with_zero_taps_elided", "valu_frac_algorithmic"} & set(json.loads(open(p).read())["roofline"]):
        new = True
if new:  # round 4 on: executed and algorithmic operations differ where zero-coefficient taps are left out
    print("| config (frames per step) | GPU Mpix/s | VALU ceiling, executed ops | 2 fs^2 per sample \"algorithmic\" | taps per sample executed / reference | HBM fraction (algorithmic bytes) | interior kernel |")
    print("|---|---|---|---|---|---|---|")
else:
    print("| config (frames per step) | GPU Mpix/s | of the VALU ceiling | HBM fraction (algorithmic bytes) | interior kernel |")
    print("|---|---|---|---|---|")
for c in order:
    p = os.path.join(HERE, rdir, f"{tag}_bench_{c}.json")
    if not os.path.exists(p):
        continue
    d = json.loads(open(p).read())
    r, cfg = d["roofline"], d["config"]
    if new:
        # (round 4 called it valu_frac_algorithmic; since round 5 the line says what it is: the reference-equivalent rate)
        ref_eq = r.get('reference_equivalent_rate_vs_peak_with_zero_taps_elided', r.get('valu_frac_algorithmic', r['valu_frac']))
        print(f"| {cfg['workload']} ({cfg['frames_per_step_per_gpu']}) | {d['value']:,.0f} | {100 * r['valu_frac']:.1f} % | {100 * ref_eq:.1f} % | "
              f"{r.get('taps_per_sample_executed', '')} / {r.get('taps_per_sample_reference', '')} | {100 * r['frac']:.1f} % | `{cfg['kernel']}` |")
    else:
        print(f"| {cfg['workload']} ({cfg['frames_per_step_per_gpu']}) | {d['value']:,.0f} | {100 * r['valu_frac']:.1f} % | {100 * r['frac']:.1f} % | `{cfg['kernel']}` |")
