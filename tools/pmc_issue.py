#!/usr/bin/env python3
"""Averages the rocprofv3 --pmc passes tools/pmc_issue.sh collected over the launches of the dominant query kernel.
usage: pmc_issue.py <dir with the pmc passes>"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(list)
name = "k_query_fused"
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
for f in files:
    rows = list(csv.DictReader(open(f)))
    if any("k_query_pipe" in r["Kernel_Name"] for r in rows):
        name = "k_query_pipe"
for f in files:
    rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
    first = min((int(r["Dispatch_Id"]) for r in rows), default=None)   # the sizing count, not a step
    for r in rows:
        if int(r["Dispatch_Id"]) != first:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"kernel {name}; averages per launch over {max((len(v) for v in acc.values()), default=0)} launches")
for k in sorted(acc):
    v = acc[k]
    print(f"  {k:28s} {sum(v) / len(v):16.0f}")
