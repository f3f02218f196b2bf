#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of `python bench.py` into profiles/pmc_traffic.json: HBM-side bytes per
launch of the dominant kernel (k_query_fused), per MI355X_MICROARCH.md §HBM:
  read bytes  = TCC_EA0_RDREQ split by request size (32/64/128 B); FETCH_SIZE is shown beside it — on gfx950 it
                tallies 128-B requests at 64 B, so it reads half of the request-derived figure
  write bytes = WRITE_SIZE (KiB) cross-checked with TCC_EA0_WRREQ (64-B requests)
usage: pmc_traffic.py <dir with the pmc passes> <out.json> [bench.json of the same command]"""
import csv
import glob
import json
import sys
from collections import defaultdict

acc = defaultdict(list)
KERNEL = "k_query_fused"
STEP_KERNELS = None  # a step made of several kernels (bivx_self_overlaps_dev): their per-launch averages are added up
if len(sys.argv) > 3:  # the bench line names the kernel that did the work
    try:
        rl = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])["roofline"]
        KERNEL = rl["dominant_kernel"]
        STEP_KERNELS = rl.get("step_kernels")
    except Exception:
        pass


def is_it(name):
    # "k_query_pipe" must not match k_query_pipe_dense
    return (KERNEL + "<") in name or (KERNEL + "(") in name or name.rstrip().endswith(KERNEL) or \
        (KERNEL == "k_query_pipe_dense" and KERNEL in name)


for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if is_it(r["Kernel_Name"])]
    # the first launch of the kernel in a bench run may be the zero-capacity count that sizes the hit buffer, not a
    # step (when that count runs on the same kernel)
    first = min((int(r["Dispatch_Id"]) for r in rows), default=None)
    skip_first = KERNEL != "k_query_pipe_dense"
    for r in rows:
        if not (skip_first and int(r["Dispatch_Id"]) == first):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
if STEP_KERNELS:
    per = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in STEP_KERNELS:
                if ("::" + k + "<") in r["Kernel_Name"] or ("::" + k + "(") in r["Kernel_Name"] or r["Kernel_Name"].rstrip().endswith(k):
                    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = defaultdict(float)
    for k in per:
        for c, v in per[k].items():
            avg[c] += sum(v) / len(v)
    avg = dict(avg)
rd = avg.get("TCC_EA0_RDREQ_sum", 0.0)
rd32, rd64 = avg.get("TCC_EA0_RDREQ_32B_sum", 0.0), avg.get("TCC_EA0_RDREQ_64B_sum", 0.0)
rd128 = avg.get("TCC_EA0_RDREQ_128B_sum", rd - rd32 - rd64)
read_bytes = rd32 * 32 + rd64 * 64 + rd128 * 128
wr = avg.get("TCC_EA0_WRREQ_sum", 0.0)
wr64 = avg.get("TCC_EA0_WRREQ_64B_sum", 0.0)
write_bytes_req = wr64 * 64 + (wr - wr64) * 32
out = {
    "kernel": KERNEL if not STEP_KERNELS else " + ".join(STEP_KERNELS), "launches_sampled": len(acc.get("TCC_EA0_RDREQ_sum", [])),
    "counters_avg_per_launch": avg,
    "read_bytes_per_launch": read_bytes,
    "fetch_size_bytes_per_launch_uncorrected": avg.get("FETCH_SIZE", 0.0) * 1024,
    "fetch_size_bytes_per_launch_x2": avg.get("FETCH_SIZE", 0.0) * 2048,
    "write_bytes_per_launch_from_requests": write_bytes_req,
    "write_size_bytes_per_launch": avg.get("WRITE_SIZE", 0.0) * 1024,
    "traffic_bytes_per_step": read_bytes + (avg.get("WRITE_SIZE", 0.0) * 1024 or write_bytes_req),
    "method": "read = sum over TCC_EA0_RDREQ request sizes; write = WRITE_SIZE; separate --pmc passes, "
              "averaged over the launches of the bench's timed loop",
}
if len(sys.argv) > 3:  # the bench line of the same command: per-query and per-algorithmic-byte ratios
    try:
        b = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
        q, alg = b["config"]["queries"], b["roofline"]["algorithmic_bytes_per_step"]
        out["workload"] = b["config"]["workload"]
        out["queries_per_launch"] = q
        out["algorithmic_bytes_per_step"] = alg
        out["traffic_over_algorithmic"] = out["traffic_bytes_per_step"] / alg
        out["rdreq_128B_per_query"] = rd128 / q
        out["l2_hit_rate"] = avg.get("TCC_HIT_sum", 0.0) / max(avg.get("TCC_HIT_sum", 0.0) + avg.get("TCC_MISS_sum", 0.0), 1.0)
        out["gpu_ms_per_step_unprofiled"] = b["roofline"]["gpu_ms_per_step"]
        out["commit"] = b.get("commit")
    except Exception as e:
        out["bench_line_error"] = str(e)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k != "counters_avg_per_launch"}, indent=1))
