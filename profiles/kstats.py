"""Prints per-kernel call counts and average durations from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ewa_" in r["Name"]:
                n = r["Name"].replace("void jinc::(anonymous namespace)::", "")
                print("%-70s calls %4s avg %10.1f us  %5s%%" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
