#!/usr/bin/env python3
"""Per-kernel memory-side traffic of bivx_build from rocprofv3 --pmc passes of tools/build_bench.py (MI355X_MICROARCH.md,
HBM section: read bytes from the TCC_EA0_RDREQ request-size split — FETCH_SIZE tallies 128-B requests at 64 B on gfx950 and
is shown beside it; write bytes = WRITE_SIZE). usage: pmc_build.py <dir with the pmc passes> <intervals> [builds sampled]"""
import csv
import glob
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not name.startswith("bivx::"):
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name][r["Counter_Name"]] += 1
n = int(sys.argv[2])
rows = []
for k in sorted(acc):
    a, c = acc[k], calls[k]
    per = lambda ctr: a.get(ctr, 0.0) / max(c.get(ctr, 1), 1)
    rd32, rd64, rd = per("TCC_EA0_RDREQ_32B_sum"), per("TCC_EA0_RDREQ_64B_sum"), per("TCC_EA0_RDREQ_sum")
    rd128 = per("TCC_EA0_RDREQ_128B_sum") if "TCC_EA0_RDREQ_128B_sum" in a else max(rd - rd32 - rd64, 0.0)
    read_b = rd32 * 32 + rd64 * 64 + rd128 * 128
    write_b = per("WRITE_SIZE") * 1024
    rows.append(dict(kernel=k, launches_sampled=max(c.values()), read_bytes_per_launch=read_b,
                     fetch_size_x2_bytes_per_launch=per("FETCH_SIZE") * 2048, write_bytes_per_launch=write_b,
                     bytes_per_interval=(read_b + write_b) / n))
print(json.dumps({"intervals": n, "method": "separate rocprofv3 --pmc passes over tools/build_bench.py; read = TCC_EA0_RDREQ by "
                  "request size, write = WRITE_SIZE; per launch of each build kernel (fabric-side: includes Infinity-Cache hits)",
                  "kernels": rows}, indent=1))
