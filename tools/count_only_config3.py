#!/usr/bin/env python3
"""config 3's batch through the zero-capacity call (offsets only) a few times: what tools/pmc_any.sh profiles to split the
pipelined kernel's instructions into counting and the rest."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from binary_amd import IntervalIndex, synth
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
d = synth.gen_genome(10_000_000, 10_000_000, 1000)
idx = IntervalIndex(0)
idx.insert_node(to(d["low"]), to(d["high"]), to(d["chrom"]))
idx.build()
ql, qh, qc = to(d["qlow"]), to(d["qhigh"]), to(d["qchrom"])
off = torch.empty(ql.numel() + 1, dtype=torch.int64, device=dev)
for _ in range(8):
    idx.count_overlaps_device(ql, qh, qc, offsets=off)
torch.cuda.synchronize()
print(int(off[-1].item()))
