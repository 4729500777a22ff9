#!/usr/bin/env python3
"""Averages rocprofv3 counter_collection CSVs per (kernel, counter): usage pmc_summary.py <dir>."""
import collections, csv, glob, json, os, re, sys

def short(name):
    m = re.search(r"(ewa_\w+)", name)
    return m.group(1) if m else None

agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
out = collections.defaultdict(dict)
for (k, c), v in agg.items():
    out[k][c] = sum(v) / len(v)
    out[k]["launches"] = len(v)
print(json.dumps(out, indent=1))
