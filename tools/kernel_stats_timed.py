#!/usr/bin/env python
"""Average duration of the TIMED launches of the column kernel in a rocprofv3 --kernel-trace run of bench.py:
the last K dispatches (bench.py does W warm-up launches first; rocprofv3 --stats averages over all W + K).
usage: kernel_stats_timed.py <dir with *_kernel_trace.csv> <K> [kernel substring [skip]]
       with `skip` the K launches after the first `skip` ones are taken instead of the last K (the default bench command
       times three workloads one after the other: the headline's launches come first)"""
import csv
import glob
import json
import sys

d, k = sys.argv[1], int(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "thompson_column_step"
skip = int(sys.argv[4]) if len(sys.argv) > 4 else None
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if name in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
by_kernel = {}
for st, en, kn in rows:
    by_kernel.setdefault(kn, []).append(en - st)
out = {}
for kn, durs in by_kernel.items():
    timed = durs[-k:] if skip is None else durs[skip:skip + k]
    out[kn] = {"launches_total": len(durs), "launches_timed": len(timed),
               "avg_ns_timed": sum(timed) / len(timed), "min_ns_timed": min(timed), "max_ns_timed": max(timed),
               "avg_ns_all": sum(durs) / len(durs)}
print(json.dumps(out, indent=1))
