#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/...) into the small summaries kept under profiles/.

usage: summarize.py <round-tag> <stats_dir> [<fetch_dir> <write_dir>]      (env JINC_FRAMES_PER_LAUNCH: recorded in the summary;
                                                                          env JINC_PROFILE_DIR: output directory under profiles/)
  stats_dir : output of  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python bench.py ...
  fetch_dir : output of  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python bench.py ...
  write_dir : output of  rocprofv3 --pmc WRITE_SIZE ...   (separate pass: TCC has 4 slots, FETCH_SIZE takes 3)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (MI355X_MICROARCH.md, HBM section: bytes =
counter * 1024).  Both the raw sum and the sum with the guide's x2 correction of FETCH_SIZE are written;
profiles/make_traffic.py publishes the corrected one (the raw FETCH figure is below the compulsory source bytes).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(ewa_\w+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2)}>" if m else None


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("JINC_PROFILE_DIR", ""))
    os.makedirs(out_dir, exist_ok=True)
    out = {"tag": tag, "kernels": {}}
    if os.environ.get("JINC_FRAMES_PER_LAUNCH"):
        out["frames_per_launch"] = int(os.environ["JINC_FRAMES_PER_LAUNCH"])
    for f in glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                out["kernels"].setdefault(k, {}).update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]),
                                                        min_ns=int(r["MinNs"]), max_ns=int(r["MaxNs"]),
                                                        pct=float(r["Percentage"]))
    for d in sys.argv[3:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            meta = {}
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
                    meta[k] = dict(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]),
                                   grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]))
            for (k, c), v in agg.items():
                e = out["kernels"].setdefault(k, {})
                e[c + "_KiB_mean"] = sum(v) / len(v)
                e[c + "_bytes_mean"] = sum(v) / len(v) * 1024
                e[c + "_launches"] = len(v)
                e.update(meta[k])
    # GRBM_GUI_ACTIVE pass (third pmc directory, optional): effective clock of a dispatch = counter / 8 XCDs / its duration
    # (MI355X_MICROARCH.md, DVFS give-back); durations from the same pass's kernel trace, joined by dispatch id
    for d in sys.argv[3:]:
        dur = {}
        for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r.get("Dispatch_Id")] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            clk = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k or r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                    continue
                ns = dur.get(r.get("Dispatch_Id"))
                if ns is None and "Start_Timestamp" in r:
                    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                if ns:
                    clk[k].append((float(r["Counter_Value"]) / 8.0 / ns, ns))
            for k, v in clk.items():
                e = out["kernels"].setdefault(k, {})
                e["effective_clock_ghz"] = sum(c for c, _ in v) / len(v)
                e["clock_pass_kernel_ns"] = sum(n for _, n in v) / len(v)
                e["clock_pass_launches"] = len(v)
    for k, e in out["kernels"].items():
        if "FETCH_SIZE_bytes_mean" in e and "WRITE_SIZE_bytes_mean" in e:
            e["hbm_bytes_per_launch_raw"] = e["FETCH_SIZE_bytes_mean"] + e["WRITE_SIZE_bytes_mean"]
            e["hbm_bytes_per_launch_fetch_x2"] = 2 * e["FETCH_SIZE_bytes_mean"] + e["WRITE_SIZE_bytes_mean"]
    for f in glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True):
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as o:
            o.write(open(f).read())
    path = os.path.join(out_dir, f"{tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
