#!/usr/bin/env python3
"""Averages the rocprofv3 --pmc passes tools/pmc_issue.sh collected over the launches of the dominant query kernel.
usage: pmc_issue.py <dir with the pmc passes>"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(list)
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
# the query kernel with the most dispatches is the one the steps ran (the sizing count is a single launch)
seen = defaultdict(int)
for f in files[:1]:
    for r in csv.DictReader(open(f)):
        for k in ("k_query_pipe_dense", "k_query_pipe<", "k_query_fused"):
            if k in r["Kernel_Name"]:
                seen[k] += 1
                break
name = max(seen, key=seen.get) if seen else "k_query_fused"
for f in files:
    rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
    first = min((int(r["Dispatch_Id"]) for r in rows), default=None)   # the sizing count, when it ran on this kernel
    for r in rows:
        if name == "k_query_pipe_dense" or int(r["Dispatch_Id"]) != first:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"kernel {name}; averages per launch over {max((len(v) for v in acc.values()), default=0)} launches")
for k in sorted(acc):
    v = acc[k]
    print(f"  {k:28s} {sum(v) / len(v):16.0f}")
