#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output (one row per dispatch and counter) as per-kernel averages."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(bivx::IndexView")[0].split("(unsigned")[0][:60]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:40s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
