#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output for one kernel: per-dispatch mean of every counter.
usage: summarise_pmc.py <dir with *_counter_collection.csv> <kernel substring> [out.json]"""
import csv
import glob
import json
import sys
from collections import defaultdict

d, kname = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
meta = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kname not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        meta = {k: row[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                    "Accum_VGPR_Count", "SGPR_Count") if k in row}
out = {"kernel": kname, "dispatches": max((len(v) for v in acc.values()), default=0), "meta": meta,
       "mean": {k: sum(v) / len(v) for k, v in sorted(acc.items())}}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
