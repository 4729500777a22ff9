#!/usr/bin/env python3
"""Re-measures the cross-over rules of csrc/dispatch.cpp (struct Rules) on the box it runs on: for every rule one bench run with the
rule's choice and one with the alternative at a batch size next to the threshold, printed as a table with the ratio.  The constants
were each taken from one box's A/B; boxes of the pool differ by up to 9 % in the clock they hold, and several rules sit within a few
per cent of level -- this is the tool that says which ones a given box would set differently.

    python profiles/recheck_rules.py [out.log]            (through gpurun, from the repository root)

Every run is a child process (`bench.py --steps 20 --no-cpu-baseline --no-e2e --no-clock-sampler`) with the A/B knob in its environment."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (rule, configuration, frames per call, {env / args of the rule's choice}, {env / args of the alternative}, what the alternative is)
CHECKS = [
    ("kFlSubMinFrames = 2: sub-group form from 2 frames", "A137", 2, {}, {"args": ["--kernel-mode", "1"]}, "gather kernel"),
    ("kFlSubMinFrames = 2: gather kernel for one frame", "A137", 1, {}, {"args": ["--kernel-mode", "16"]}, "sub-group form"),
    ("kFlSub4MaxFrames = 48 -> 4 sub-groups at 16 frames", "A137", 16, {}, {"env": {"JINC_FL_SUB": "2"}}, "2 sub-groups"),
    ("kFlSub4MaxFrames = 48 -> 4 sub-groups at 32 frames", "A137", 32, {}, {"env": {"JINC_FL_SUB": "2"}}, "2 sub-groups"),
    ("kFlSub4MaxFrames = 48: sub-groups at 48 frames", "A137", 48, {}, {"env": {"JINC_FL_SUB": "0"}}, "64-frame form"),
    ("kFlSub4MaxFrames = 48: 64-frame form above", "A137", 56, {}, {"env": {"JINC_FL_SUB": "4"}}, "4 sub-groups"),
    ("sub-group form at all (16 frames)", "D169", 16, {}, {"env": {"JINC_FL_SUB": "0"}}, "64-frame form"),
    ("frame-pair form for whole groups of 128", "A137", 128, {}, {"args": ["--kernel-mode", "11"]}, "64-frame form"),
    # (kFlColsMinFrames -- border columns of periodic plans on the frame-lane kernel from 16 frames on -- no longer decides anything at
    #  source step 1: since round 5 the interior kernel's edge tiles or ewa_colpair_kernel take those columns; rules further down)
    ("kRunsFrameLaneBorderMinFramesSub = 8", "N15T4", 8, {}, {"env": {"JINC_RUNS_FL_BORDER_FRAMES": "0"}}, "gather kernel on the border"),
    ("kRunsFrameLaneBorderMinFrames = 32 (tap 8)", "N15T8", 32, {}, {"env": {"JINC_RUNS_FL_BORDER_FRAMES": "0"}}, "gather kernel on the border"),
    ("kQuad2x8MinWorkgroups = 1024: two periods per lane on 8 x 8 from 2 frames", "C2T4", 4, {}, {"env": {"JINC_QUAD2X8": "0"}}, "one period per lane"),
    ("kQuad2x8MinWorkgroups = 1024: one period per lane for one frame", "C2T4", 1, {}, {"env": {"JINC_QUAD2X8": "1"}}, "two periods per lane"),
    ("trimmed support, integer planes", "C2", 64, {}, {"args": ["--kernel-mode", "15"]}, "full window"),
    ("trimmed support on float planes (8 x 8)", "C4", 16, {}, {"args": ["--kernel-mode", "15"]}, "full window"),
    ("float planes on the 6 x 6 support (the trimmed launch its own scan)", "C2F", 64, {}, {"args": ["--kernel-mode", "15"]}, "full window"),
    ("float planes: no scan pass in front of the trimmed launch", "C4", 16, {}, {"env": {"JINC_FLOAT_SCAN": "1"}}, "scan pass over the source"),
    ("kFloatTrimMinTaps = 1e9: one 1080p -> 4K float frame stays on the full window", "C2F", 1, {}, {"env": {"JINC_FLOAT_TRIM_MIN_TAPS": "0"}}, "trimmed"),
    ("rows kernel: per-row spans and row count", "C3", 32, {}, {"args": ["--kernel-mode", "15"]}, "full window"),
    ("direct kernel's interior on the trimmed support", "D12", 128, {}, {"args": ["--kernel-mode", "15"]}, "full window"),
    ("quad form of the periodic kernel (single frames)", "C2", 1, {}, {"args": ["--kernel-mode", "2"]}, "window kernel"),
    # round 5: the rows kernel in packed phase-pair form (taps 5 .. 8 at 2x), its tile shape, and the quad forms on taps 3 / 4
    ("row-pair kernel for 12 .. 17 taps per kernel row (tap 8)", "C3", 32, {}, {"env": {"JINC_ROWS_PAIR": "0"}}, "rows kernel"),
    ("row-pair kernel (tap 6)", "T6", 64, {}, {"env": {"JINC_ROWS_PAIR": "0"}}, "rows kernel"),
    ("row-pair kernel, one C3 frame per call", "C3", 1, {}, {"env": {"JINC_ROWS_PAIR": "0"}}, "rows kernel"),
    ("row-pair tile shape: the squarer tile on a tie", "C3", 32, {}, {"env": {"JINC_ROWS_PAIR": "32"}}, "128 x 32 tiles"),
    ("quad forms, not the row-pair kernel, on 6 taps per kernel row", "C2", 64, {}, {"env": {"JINC_ROWPAIR_SMALL": "1"}}, "row-pair kernel"),
    ("quad forms, not the row-pair kernel, on 8 taps per kernel row (float)", "C4", 16, {}, {"env": {"JINC_ROWPAIR_SMALL": "1"}}, "row-pair kernel"),
    ("border rows on ewa_strip_kernel", "C2", 1024, {}, {"env": {"JINC_STRIP_LDS": "0"}}, "ewa_direct_kernel row strips"),
    # round 5, second half: the border where it executes least
    ("border columns inside the interior kernel's edge tiles (tap 3)", "C2", 1024, {}, {"env": {"JINC_EDGE_COLS": "0"}}, "column kernel beside the interior"),
    ("border columns inside the interior kernel's edge tiles (tap 3, 16 frames)", "C2", 16, {}, {"env": {"JINC_EDGE_COLS": "0"}}, "column kernel beside the interior"),
    ("border columns inside the interior kernel's edge tiles (tap 4)", "C2T4", 256, {}, {"env": {"JINC_EDGE_COLS": "0"}}, "column kernel beside the interior"),
    ("border columns on ewa_colpair_kernel (filter size 17)", "C3", 32, {}, {"env": {"JINC_COLPAIR": "0"}}, "ewa_colstrip_kernel"),
    ("border columns on ewa_colpair_kernel (filter size 9, float)", "C4", 16, {}, {"env": {"JINC_COLPAIR": "0"}}, "frame-lane kernel's sub-group form"),
    ("border columns on ewa_colpair_kernel (filter size 9, two frames per call)", "C4", 2, {}, {"env": {"JINC_COLPAIR": "0"}}, "ewa_strip_kernel columns"),
    ("border columns on ewa_colpair_kernel (filter size 7, float)", "C2F", 64, {}, {"env": {"JINC_COLPAIR": "0"}}, "frame-lane kernel"),
    ("kStripBorderMinTapsEdgeCols = 2.4e9: strip border from 6 C2 frames per call", "C2", 8, {}, {"args": ["--border-strips", "0"]}, "gather kernel over the border frame"),
    ("kStripBorderMinTapsEdgeCols = 2.4e9: gather border below (4 C2 frames)", "C2", 4, {}, {"args": ["--border-strips", "4"]}, "strip border"),
    ("kStripBorderMinTapsEdgeCols: C1 at 64 frames", "C1", 64, {}, {"args": ["--border-strips", "0"]}, "gather kernel over the border frame"),
    ("kStripBorderMinTaps = 5e9 with chroma planes: gather border at 8 frames", "C2YUV", 8, {}, {"args": ["--border-strips", "4"]}, "strip border"),
    ("border rows of taps 5 .. 8 as strips of the row-pair kernel (tap 8)", "C3", 32, {}, {"env": {"JINC_ROWPAIR_ROWS": "0"}}, "ewa_direct_kernel row strips"),
    ("border rows of taps 5 .. 8 as strips of the row-pair kernel (tap 6)", "T6", 64, {}, {"env": {"JINC_ROWPAIR_ROWS": "0"}}, "ewa_direct_kernel row strips"),
]


def run(cfg, frames, variant):
    env = dict(os.environ)
    env.update(variant.get("env", {}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--frames", str(frames), "--steps", "20", "--warmup", "3",
           "--no-cpu-baseline", "--no-e2e", "--no-clock-sampler"] + variant.get("args", [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
    line = json.loads(out[-1])
    return line["value"], line["roofline"]["kernel"]


def main():
    rows = []
    first = int(os.environ.get("RECHECK_FROM", "0"))   # (a call of gpurun lasts 20 minutes at most: the list in two or three parts)
    for rule, cfg, frames, choice, other, other_name in CHECKS[first:int(os.environ.get("RECHECK_TO", str(len(CHECKS))))]:
        a = [run(cfg, frames, choice) for _ in range(2)]   # alternating: choice, alternative, choice, alternative
        b = [run(cfg, frames, other) for _ in range(2)]
        va, vb = max(v for v, _ in a), max(v for v, _ in b)
        verdict = "holds" if va >= vb else ("level" if va >= 0.98 * vb else "FLIPS on this box")
        rows.append(f"{rule:66s} {cfg:6s} {frames:4d} frames  rule {va / 1000:8.1f} Gpix/s ({a[0][1]})  {other_name}: {vb / 1000:8.1f} ({b[0][1]})  "
                    f"ratio {va / vb:5.3f}  {verdict}")
        print(rows[-1], flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            fh.write("\n".join(rows) + "\n")


if __name__ == "__main__":
    main()
