import sys, numpy as np, torch
sys.path.insert(0, '.')
from binary_amd import IntervalIndex, synth
dev = torch.device("cuda:0")
L = int(synth.HG38_LENGTHS[0])
lo, hi = synth.gen_intervals(1_000_000, L, 1000, 0)
ql, qh = synth.gen_point_queries(1_000_000, L, 0)
to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
idx = IntervalIndex(0); idx.insert_node(lo, hi); idx.build()
Q = ql.size
off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
hits = torch.empty(3_000_000, dtype=torch.int32, device=dev)
ws = torch.empty(idx.query_workspace_bytes(Q), dtype=torch.uint8, device=dev)
dql, dqh = to(ql), to(qh)
def timed(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rep in range(3):
    a = timed(lambda: idx.query_device(dql, dqh, off, hits, ws))
    b = timed(lambda: idx.query_device(dql, dqh, off, hits))
    print(f"caller ws + memset: {a:.1f} us   index-owned self-cleaning: {b:.1f} us")
