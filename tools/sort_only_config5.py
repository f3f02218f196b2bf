#!/usr/bin/env python3
"""config 5's self-overlap CSR ordered by k_sort_hits a few times (bivx_sort_hits through the self-overlap call with
sort_by_id): what tools/pmc_any.sh profiles to see what the ordering pass spends its time on."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from binary_amd import IntervalIndex, synth
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
d = synth.gen_genome(n, 0, 1000)
idx = IntervalIndex(0)
idx.insert_node(to(d["low"]), to(d["high"]), to(d["chrom"]))
idx.build()
off = torch.empty(n + 1, dtype=torch.int64, device=dev)
idx.self_overlaps_device(off, torch.empty(1, dtype=torch.int32, device=dev))
H = int(off[-1].item())
hits = torch.empty(H, dtype=torch.int32, device=dev)
for _ in range(3):
    idx.self_overlaps_device(off, hits, sort_by_id=True)
torch.cuda.synchronize()
print(H)
