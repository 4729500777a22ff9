#!/usr/bin/env python3
"""Reads a rocprofv3 rocpd database (--hip-trace --kernel-trace --memory-copy-trace) and prints (1) host time per HIP API,
(2) copies by size: count, average duration, concurrency, (3) a binned busy timeline of H2D / D2H copies and kernels.
usage: python profiles/trace_timeline.py <results.db> [from_ms to_ms]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
t0 = c.execute("select min(start) from kernels").fetchone()[0]
t1 = c.execute("select max(end) from kernels").fetchone()[0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else (t1 - t0) / 1e6 * 0.5
hi = float(sys.argv[3]) if len(sys.argv) > 3 else lo + 30
print("== host time per HIP call")
for r in c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6, max(end-start)/1e3 from regions group by name order by 4 desc limit 12"):
    print("%-32s n=%6d avg=%9.1f us total=%9.1f ms max=%9.1f us" % r)
print("== kernels")
for r in c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc limit 6"):
    print("%-90s n=%5d avg=%9.1f us total=%8.1f ms" % (r[0][:90], r[1], r[2], r[3]))
print("== copies by size")
for r in c.execute("select size, count(*), avg(end-start)/1e3, sum(end-start)/1e6, count(distinct stream_id) from memory_copies group by size order by 2 desc limit 6"):
    print("size %9d n=%6d avg=%7.1f us busy=%8.1f ms streams=%d  -> %.1f GB/s per copy" % (r[0], r[1], r[2], r[3], r[4], r[0] / r[2] / 1e3))
a, b = t0 + lo * 1e6, t0 + hi * 1e6
sizes = [r[0] for r in c.execute("select size from memory_copies group by size order by count(*) desc limit 2")]
small = min(sizes) if len(sizes) == 2 else -1
ev = [(s, e, "H2D" if sz == small else "D2H") for s, e, sz in c.execute("select start,end,size from memory_copies where start>=? and start<=?", (a, b))]
ev += [(s, e, "B" if "blit" in n else "K") for s, e, n in c.execute("select start,end,name from kernels where start>=? and start<=?", (a, b))]
bins = collections.defaultdict(collections.Counter)
w = 0.5e6
for s, e, k in ev:
    x = s
    while x < e:
        bi = int((x - a) // w)
        nb = a + (bi + 1) * w
        bins[bi][k] += min(e, nb) - x
        x = min(e, nb)
print("== busy per 0.5 ms (100 %% = one copy / kernel at a time; more = overlapping)")
for bi in sorted(bins):
    d = bins[bi]
    print("%7.1f ms  H2D %4.0f%%  D2H %4.0f%%  resample %4.0f%%  blit %4.0f%%" % (lo + bi * 0.5, d["H2D"] / w * 100, d["D2H"] / w * 100, d["K"] / w * 100, d["B"] / w * 100))
