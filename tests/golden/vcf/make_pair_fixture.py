#!/usr/bin/env python3
"""Authors the paired SV / NL fixture VCFs for sv2nl and the expected TSVs.

The reference ships only a ScanNLS-style NL VCF (test/data/debug*.vcf*) and no delly-style SV VCF, and it holds
no expected sv2nl output, so the pair is authored here: pair_sv.vcf (END for
DUP/INV/DEL, CHR2+POS2 for BND) and pair_nl.vcf (SVEND; TDUP, INV with STRAND1/2, TRA with CHR2). Expected
outputs come from oracle/sv2nl_oracle.py — sv2nl-level parity is therefore "unpinned" against the reference
itself; the reader is pinned separately on the reference's own fixture.

Run from the repository root:  python tests/golden/vcf/make_pair_fixture.py
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)

CONTIGS = [("chr1", 248956422), ("chr2", 242193529), ("chr10", 133797422), ("chr17", 83257441), ("chrX", 156040895),
           ("chrM", 16569), ("chrUn_KI270302v1", 2274), ("chr1_KI270706v1_random", 175055)]
PRIMARY = ["chr1", "chr2", "chr10", "chr17", "chrX"]

INFO_SV = """##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">
##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the structural variant">
##INFO=<ID=CHR2,Number=1,Type=String,Description="Chromosome for POS2 coordinate in case of an inter-chromosomal translocation">
##INFO=<ID=POS2,Number=1,Type=Integer,Description="Genomic position for CHR2 in case of an inter-chromosomal translocation">
##INFO=<ID=PE,Number=1,Type=Integer,Description="Paired-end support of the structural variant">
##INFO=<ID=CT,Number=1,Type=String,Description="Paired-end signature induced connection type">"""
INFO_NL = """##INFO=<ID=CANONICAL,Number=0,Type=Flag,Description="Canonical splice site">
##INFO=<ID=SVTYPE,Number=1,Type=String,Description="The type of event, INS, DEL, TDUP, IDUP, INV, TRA.">
##INFO=<ID=CHR2,Number=1,Type=String,Description="Chromosome for END coordinate in case of a translocation">
##INFO=<ID=SVEND,Number=1,Type=Integer,Description="2nd position of the structural variant">
##INFO=<ID=SR,Number=1,Type=Integer,Description="The number of support reads for the breakpoints">
##INFO=<ID=STRAND1,Number=1,Type=String,Description="Strand for breakpoint1">
##INFO=<ID=STRAND2,Number=1,Type=String,Description="Strand for breakpoint2">"""


def header(info):
    lines = ["##fileformat=VCFv4.2"]
    lines += [f"##contig=<ID={c},length={l}>" for c, l in CONTIGS]
    lines += info.split("\n")
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample")
    return lines


def main():
    rng = random.Random(20221004)
    sv, nl = [], []
    dups, invs, bnds = [], [], []
    # SV side -------------------------------------------------------------------------------------------
    for c in PRIMARY + ["chrM", "chr1_KI270706v1_random"]:
        L = dict(CONTIGS)[c]
        for _ in range(40 if c in PRIMARY else 4):
            t = rng.choice(["DUP", "DUP", "INV", "INV", "DEL"])
            p = rng.randrange(1, max(2, L - 2_000_000))
            e = p + rng.choice([300, 5_000, 80_000, 1_500_000]) + rng.randrange(0, 1000)
            e = min(e, L)
            sv.append((c, p, f"SVTYPE={t};END={e};PE={rng.randrange(2, 30)};CT=3to5"))
            (dups if t == "DUP" else invs if t == "INV" else []).append((c, p, e))
    # a DUP written with POS > END (validate_record swaps it, helper.hpp:52-63), and exact duplicates
    sv.append(("chr2", 5_000_900, "SVTYPE=DUP;END=5000100;PE=9;CT=5to3"))
    dups.append(("chr2", 5_000_100, 5_000_900))
    sv.append(sv[0])
    for _ in range(60):  # translocations: BND with CHR2 / POS2, both chromosome orders, some POS > POS2
        c1, c2 = rng.sample(PRIMARY, 2)
        p1, p2 = rng.randrange(1, 80_000_000), rng.randrange(1, 80_000_000)
        sv.append((c1, p1, f"SVTYPE=BND;END={p1 + 1};CHR2={c2};POS2={p2};PE=5;CT=3to3"))
        bnds.append((c1, p1, c2, p2))
    sv.sort(key=lambda r: ([c for c, _ in CONTIGS].index(r[0]), r[1]))
    # NL side -------------------------------------------------------------------------------------------
    def near(x, d):
        return max(1, x + rng.randrange(-d, d + 1))
    for c, p, e in dups:  # TDUPs inside / around DUPs, some too far, some reversed (POS > SVEND), some repeated
        for _ in range(2):
            a, b = near(p + (e - p) // 4, 50), near(e - (e - p) // 4, 50)
            if rng.random() < 0.3:
                a, b = near(p, 2000), near(e, 2000)
            if rng.random() < 0.15:
                a, b = b, a
            nl.append((c, a, f"CANONICAL;SVTYPE=TDUP;SR=3;CHR2={c};SVEND={b};STRAND1=+;STRAND2=+"))
    nl.append(nl[0])  # identical key twice: written once
    for c, p, e in invs:  # INV partially overlapping either side, with all strand combinations
        for s1, s2 in (("+", "-"), ("-", "+"), ("+", "+")):
            side = rng.random() < 0.5
            if side:
                a, b = near(p - (e - p) // 3, 30), near(p + (e - p) // 3, 30)
            else:
                a, b = near(e - (e - p) // 3, 30), near(e + (e - p) // 3, 30)
            a = max(1, a)
            nl.append((c, a, f"SVTYPE=INV;SR=2;CHR2={c};SVEND={b};STRAND1={s1};STRAND2={s2}"))
    nl.append(("chr10", 1000, "SVTYPE=INV;SR=2;CHR2=chr10;SVEND=2000"))  # INV without strand tags: defaults '+','+'
    for c1, p1, c2, p2 in bnds:  # TRA near BNDs, in either orientation, some beyond --dis
        d = rng.choice([10, 500, 900_000, 1_200_000])
        if rng.random() < 0.5:
            nl.append((c1, near(p1, d), f"CANONICAL;SVTYPE=TRA;SR=2;CHR2={c2};SVEND={near(p2, d)};STRAND1=+;STRAND2=+"))
        else:
            nl.append((c2, near(p2, d), f"CANONICAL;SVTYPE=TRA;SR=2;CHR2={c1};SVEND={near(p1, d)};STRAND1=+;STRAND2=-"))
    for c in PRIMARY:  # unrelated records, an INS, records on skipped contigs
        for _ in range(10):
            p = rng.randrange(1, dict(CONTIGS)[c] - 10_000)
            nl.append((c, p, f"SVTYPE=TDUP;SR=2;CHR2={c};SVEND={p + rng.randrange(1, 5000)};STRAND1=+;STRAND2=+"))
        nl.append((c, 12345, f"SVTYPE=INS;SR=2;CHR2={c};SVEND=12345"))
    nl.append(("chr1_KI270706v1_random", 100, "SVTYPE=TDUP;SR=2;CHR2=chr1_KI270706v1_random;SVEND=90000;STRAND1=+;STRAND2=+"))
    nl.append(("chrM", 10, "SVTYPE=TDUP;SR=2;CHR2=chrM;SVEND=16000;STRAND1=+;STRAND2=+"))
    nl.sort(key=lambda r: ([c for c, _ in CONTIGS].index(r[0]), r[1]))

    def write(path, info, recs, alt):
        with open(path, "w") as f:
            f.write("\n".join(header(info)) + "\n")
            for i, (c, p, inf) in enumerate(recs):
                t = inf.split("SVTYPE=")[1].split(";")[0]
                f.write(f"{c}\t{p}\t{alt}{i:05d}\tN\t<{t}>\t.\tPASS\t{inf}\tGT\t0/1\n")

    write(os.path.join(HERE, "pair_sv.vcf"), INFO_SV, sv, "SV")
    write(os.path.join(HERE, "pair_nl.vcf"), INFO_NL, nl, "NL")

    from oracle import sv2nl_oracle
    for tag, kw in (("", dict()), ("_short_dis50000", dict(dis=50000, use_strand=False))):
        res = sv2nl_oracle.run(os.path.join(HERE, "pair_nl.vcf"), os.path.join(HERE, "pair_sv.vcf"), **kw)
        assert res["sv_error"] is None and res["nl_error"] is None
        for k in ("dup", "inv", "tra"):
            with open(os.path.join(HERE, f"pair_expected{tag}.{k}.tsv"), "w") as f:
                f.write(sv2nl_oracle.HEADER + "\n")
                f.write("".join(l + "\n" for l in sorted(res[k])))
            print(f"pair_expected{tag}.{k}.tsv: {len(res[k])} lines")


if __name__ == "__main__":
    main()
