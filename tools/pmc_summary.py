#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass): per kernel, mean counter value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "sglk" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:32s} mean {sum(v)/len(v):18.1f}   n={len(v)}")
