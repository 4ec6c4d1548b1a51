#!/usr/bin/env python3
"""Mean per-launch counter values per kernel from a directory of rocprofv3 --pmc CSV passes."""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in vals.items() if "project" in k or "attn" in k}
print(json.dumps(out, indent=1))
