#!/usr/bin/env python3
"""Prints a rocprofv3 --kernel-trace --stats kernel_stats.csv compactly (name, calls, avg/min/max us)."""
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(bivx::IndexView")[0].split("(unsigned")[0][:64]
            print(f"{name:64s} {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}  {float(r['Percentage']):6.2f}%")
