import sys, numpy as np, torch, json
sys.path.insert(0, ".")
from binary_amd import IntervalIndex, synth
N = 50_000_000
dev = torch.device("cuda:0")
data = synth.gen_genome(N, 0, 1000)
o = np.lexsort((data["low"], data["chrom"]))
c, lo, hi = data["chrom"][o], data["low"][o], data["high"][o]
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
d_c, d_lo, d_hi = to(c), to(lo), to(hi)
idx = IntervalIndex(0); idx.insert_node(d_lo, d_hi, d_c); idx.build()
off = idx.count_overlaps_device(d_lo, d_hi, d_c)
H = int(off[-1].item())
hits = torch.empty(H, dtype=torch.int32, device=dev)
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 2
a = timed(lambda: idx.query_device(d_lo, d_hi, off, hits, qchrom=d_c))
b = timed(lambda: idx.query_device(d_lo, d_hi, off, hits, qchrom=d_c, sort_by_id=True))
beg = torch.empty(N, dtype=torch.int64, device=dev); cnt = torch.empty(N, dtype=torch.int32, device=dev); tot = torch.zeros(1, dtype=torch.int64, device=dev)
u = timed(lambda: idx.query_device_unordered(d_lo, d_hi, beg, cnt, hits, tot, qchrom=d_c))
print(json.dumps({"config5_position_sorted": {"N": N, "H": H, "single_pass_ms": a, "sorted_ids_ms": b, "unordered_ms": u, "build_ms": idx.stats()["build_ms"]}}))
