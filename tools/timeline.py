#!/usr/bin/env python3
"""One epoch of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue and short name per kernel
(the last `count` kernels before the end of the trace).  Usage: python tools/timeline.py <kernel_trace.csv> [count]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-count:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
