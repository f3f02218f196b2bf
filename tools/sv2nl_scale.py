#!/usr/bin/env python3
"""Times the sv2nl tool on synthetic VCFs of the size the reference quotes for its only published number
(documentation/current_tools/sv2nl.md:51-55: SV VCF 72,496 records x NL VCF 10,510 records -> 18 s, hardware
unstated, "not a benchmark"). Different data (synthetic), so the comparison is indicative only.
usage: sv2nl_scale.py [N_SV N_NL]"""
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "vcf"))
import make_pair_fixture as mk  # noqa: E402  (reuses the header writers)

N_SV, N_NL = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (72496, 10510)
rng = random.Random(7)
chroms = [c for c, _ in mk.CONTIGS if "_" not in c and c != "chrM"]
sv, nl = [], []
for i in range(N_SV):
    c = rng.choice(chroms)
    L = dict(mk.CONTIGS)[c]
    t = rng.choice(["DUP", "INV", "DEL", "BND"])
    p = rng.randrange(1, L - 3_000_000)
    if t == "BND":
        c2 = rng.choice(chroms)
        sv.append((c, p, f"SVTYPE=BND;END={p+1};CHR2={c2};POS2={rng.randrange(1, 80_000_000)};PE=5;CT=3to3"))
    else:
        sv.append((c, p, f"SVTYPE={t};END={p + rng.choice([300, 5000, 80000, 1500000])};PE=5;CT=3to5"))
for i in range(N_NL):
    s = rng.choice(sv)
    c, p, inf = s
    t = inf.split("SVTYPE=")[1].split(";")[0]
    e = int(inf.split("END=")[1].split(";")[0])
    if t == "DUP":
        nl.append((c, p + 10, f"SVTYPE=TDUP;SR=3;CHR2={c};SVEND={max(p + 11, e - 10)};STRAND1=+;STRAND2=+"))
    elif t == "INV":
        nl.append((c, max(1, p - 100), f"SVTYPE=INV;SR=3;CHR2={c};SVEND={p + (e - p) // 2};STRAND1=+;STRAND2=-"))
    elif t == "BND":
        c2 = inf.split("CHR2=")[1].split(";")[0]
        p2 = int(inf.split("POS2=")[1].split(";")[0])
        nl.append((c, p + 5, f"SVTYPE=TRA;SR=3;CHR2={c2};SVEND={p2 + 7};STRAND1=+;STRAND2=+"))
    else:
        nl.append((c, p, f"SVTYPE=TDUP;SR=3;CHR2={c};SVEND={p + 100};STRAND1=+;STRAND2=+"))
order = {c: i for i, (c, _) in enumerate(mk.CONTIGS)}
sv.sort(key=lambda r: (order[r[0]], r[1]))
nl.sort(key=lambda r: (order[r[0]], r[1]))
d = tempfile.mkdtemp()


def write(path, info, recs, tag):
    with open(path, "w") as f:
        f.write("\n".join(mk.header(info)) + "\n")
        for i, (c, p, inf) in enumerate(recs):
            t = inf.split("SVTYPE=")[1].split(";")[0]
            f.write(f"{c}\t{p}\t{tag}{i}\tN\t<{t}>\t.\tPASS\t{inf}\tGT\t0/1\n")


write(os.path.join(d, "sv.vcf"), mk.INFO_SV, sv, "SV")
write(os.path.join(d, "nl.vcf"), mk.INFO_NL, nl, "NL")
subprocess.run(["make", "-C", os.path.join(ROOT, "binary_amd", "sv2nl"), "-s", "all"], check=True)
tool = os.path.join(ROOT, "binary_amd", "sv2nl", "sv2nl")
# --host-filter (unfiltered hits post-filtered on the host, a debugging aid) only at the small size: at 1 M records the
# unfiltered hit lists are thousands of ids per query
for extra in ([], ["--host-filter"]) if N_SV <= 100_000 else ([],):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        subprocess.run([tool, os.path.join(d, "sv.vcf"), os.path.join(d, "nl.vcf"), "-o", os.path.join(d, "out.tsv")] + extra,
                       check=True, capture_output=True)
        best = min(best, time.perf_counter() - t0)
    lines = sum(sum(1 for _ in open(os.path.join(d, "out.tsv" + e))) - 1 for e in (".dup", ".inv", ".tra"))
    print(f"sv2nl {' '.join(extra) or '(device filters)'}: {N_SV} SV x {N_NL} NL records, {lines} output lines, best of 3 wall {best:.3f} s "
          f"(sizes: sv {os.path.getsize(os.path.join(d, 'sv.vcf'))/1e6:.1f} MB, nl {os.path.getsize(os.path.join(d, 'nl.vcf'))/1e6:.1f} MB)")
