"""oracle/sv2nl_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of what the reference's `sv2nl` tool computes, driven exactly as its mappers prescribe, on top
of the tree oracle (oracle/ivtree.c). Citations are under /root/reference/standalone/sv2nl/.

  read VCFs            include/vcf_info.hpp:42-46, source/vcf_info.cpp:9-43 (SVTYPE, CHR2, STRAND1/2, POS2/SVEND/END)
  per-chromosome map   include/mapper.hpp:147-162 (build_tree: filter chrom && svtype, validate_record),
                       :194-236 (map_impl), :238-246 (map_delegate: NL header contigs without '_')
  translocations       source/mapper.cpp:81-170 (one un-validated tree of all BND records)
  predicates           include/helper.hpp:16-91, source/mapper.cpp:50-79,144-156
  output               source/writer.cpp:21-28, include/writer.hpp:27-53, include/mapper.hpp:29 (header)

Parity-pin status: the READER is pinned by the reference's own fixture and test values (test_vcf.cpp:94-100:
first record chr10 / TRA / pos 93567287; 6 records; 2 TRA) — tests/test_sv2nl_cpu.py. The MAPPING output is
"parity unpinned": the reference holds no delly-style SV fixture and no expected sv2nl output, so
expected TSVs are derived from this restatement. The reference's per-chromosome tasks run concurrently and append
to the output under a mutex, so its line order is not deterministic: outputs are compared as sorted line sets.
"""
from __future__ import annotations

import gzip

import numpy as np

from . import ivtree_oracle as ivt

HEADER = "chrom\tpos\tend\tsvtype\tchrom\tpos\tend\tsvtype"


class VcfReaderError(Exception):
    pass


class Rec:
    __slots__ = ("chrom", "pos", "svtype", "svend", "chr2", "strand1", "strand2")

    def __init__(self):
        self.chrom, self.pos, self.svtype, self.svend, self.chr2 = "", 0, "", 0, ""
        self.strand1 = self.strand2 = True

    def copy(self):
        r = Rec()
        for k in Rec.__slots__:
            setattr(r, k, getattr(self, k))
        return r


def read_vcf(path: str, source: str):
    """-> (contigs in header order, records up to the first unreadable record, error text or None)."""
    op = gzip.open if path.endswith(".gz") else open
    contigs, types, recs, err = [], {}, [], None
    with op(path, "rt") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if line.startswith("##"):
                if line.startswith("##contig=<") or line.startswith("##INFO=<"):
                    body = line[line.index("<") + 1:line.rindex(">")]
                    attrs = {}
                    for part in _split_attrs(body):
                        if "=" in part:
                            k, v = part.split("=", 1)
                            attrs[k] = v.strip('"')
                    if line.startswith("##contig") and attrs.get("ID") and attrs["ID"] not in contigs:
                        contigs.append(attrs["ID"])
                    elif line.startswith("##INFO") and attrs.get("ID"):
                        types.setdefault(attrs["ID"], attrs.get("Type", ""))
                continue
            if line.startswith("#"):
                continue
            try:
                recs.append(_parse_record(line, source, types, recs[-1] if recs else None))
            except VcfReaderError as e:
                err = str(e)
                break
    return contigs, recs, err


def _split_attrs(body):
    out, cur, quoted = [], "", False
    for ch in body:
        if ch == '"':
            quoted = not quoted
        if ch == "," and not quoted:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


def _info(kv, types, tag, want):
    if types.get(tag) != want or tag not in kv or kv[tag] in ("", "."):
        raise VcfReaderError("Failed to get info " + tag)
    return kv[tag]


def _parse_record(line, source, types, prev=None):
    f = line.split("\t")
    if len(f) < 8:
        raise VcfReaderError("Failed to read line in vcf ")
    r = Rec()
    if prev is not None:
        # vcf_info.cpp:9-43 updates ONE info object in place, record after record: what a record does not set (the strands
        # of an INV line without STRAND tags, CHR2 of a record that is neither TRA nor BND) is what the record before it left there
        r.strand1, r.strand2, r.chr2 = prev.strand1, prev.strand2, prev.chr2
    r.chrom = f[0]
    r.pos = (int(f[1]) - 1) & 0xFFFFFFFF
    kv = {}
    for item in f[7].split(";"):
        if item and item != ".":
            k, _, v = item.partition("=")
            kv.setdefault(k, v)
    r.svtype = _info(kv, types, "SVTYPE", "String")
    if r.svtype in ("TRA", "BND"):
        r.chr2 = _info(kv, types, "CHR2", "String")
    if r.svtype == "INV":
        try:  # vcf_info.cpp:18-31: one try block around both lookups
            r.strand1 = _info(kv, types, "STRAND1", "String") == "+"
            r.strand2 = _info(kv, types, "STRAND2", "String") == "+"
        except VcfReaderError:
            pass
    tag = "POS2" if r.svtype == "BND" else ("SVEND" if source == "nls" else "END")
    v = _info(kv, types, tag, "Integer").split(",")[0]
    try:
        r.svend = int(v) & 0xFFFFFFFF
    except ValueError:
        raise VcfReaderError("Failed to get info " + tag)
    return r


# ---- helper.hpp ------------------------------------------------------------------------------------------

def validate_record(r: Rec) -> Rec:  # helper.hpp:52-63
    v = r.copy()
    if v.pos > v.svend:
        v.pos, v.svend = v.svend, v.pos
        if r.svtype in ("BND", "TRA"):
            v.chrom, v.chr2 = v.chr2, v.chrom
    return v


def is_contained(target: Rec, source: Rec) -> bool:  # helper.hpp:16-24
    return target.pos <= source.pos and target.svend >= source.svend


def distance_less(a: Rec, b: Rec, d: int) -> bool:  # helper.hpp:31-40
    return abs(a.pos - b.pos) <= d and abs(a.svend - b.svend) <= d


def two_chroms_with_pos(r: Rec):  # helper.hpp:76-82
    return (r.chr2, r.svend, r.chrom, r.pos) if r.chrom > r.chr2 else (r.chrom, r.pos, r.chr2, r.svend)


def format_map_key(r: Rec) -> str:  # helper.hpp:84-91
    if r.svtype in ("TRA", "BND"):
        c1, p1, c2, p2 = two_chroms_with_pos(r)
        return f"{c1}-{c2}-{p1}-{p2}"
    return f"{r.chrom}-{r.pos}-{r.svend}"


def format_keys(r: Rec) -> str:  # writer.cpp:21-28
    if r.svtype in ("TRA", "BND"):
        return f"{r.chrom},{r.chr2}\t{r.pos + 1}\t{r.svend}\t{r.svtype}"
    return f"{r.chrom}\t{r.pos + 1}\t{r.svend}\t{r.svtype}"


# ---- mapper.cpp check_condition ----------------------------------------------------------------------------

def check_dup(nl: Rec, sv: Rec, d: int, use_strand: bool) -> bool:  # mapper.cpp:50-55
    return is_contained(sv, nl) and distance_less(nl, sv, d)


def check_inv(nl: Rec, sv: Rec, d: int, use_strand: bool) -> bool:  # mapper.cpp:57-79
    if is_contained(sv, nl) or is_contained(nl, sv) or not distance_less(nl, sv, d):
        return False
    if not use_strand:
        return True
    if nl.pos <= sv.pos:
        return nl.strand1 and not nl.strand2
    return (not nl.strand1) and nl.strand2


def check_tra(nl: Rec, sv: Rec, d: int, use_strand: bool) -> bool:  # mapper.cpp:144-156
    n1, np1, n2, np2 = two_chroms_with_pos(nl)
    s1, sp1, s2, sp2 = two_chroms_with_pos(sv)
    return n1 == s1 and n2 == s2 and abs(np1 - sp1) <= d and abs(np2 - sp2) <= d


def _map(nl_recs, nl_chroms, tree_of, nl_type, check, d, use_strand):
    """Mapper::map_impl over every primary chromosome; returns output lines (no header)."""
    lines, cache = [], set()
    for chrom in nl_chroms:
        if "_" in chrom:  # mapper.hpp:241-243
            continue
        tree, items = tree_of(chrom)
        for nl in nl_recs:
            if nl.chrom != chrom or nl.svtype != nl_type:
                continue
            key = format_map_key(nl)
            if key in cache:  # mapper.hpp:213
                continue
            q = validate_record(nl)
            hits = tree.find_overlaps(q.pos, q.svend) if tree is not None else []
            kept = [items[i] for i in hits if check(q, items[i], d, use_strand)]
            if kept:  # mapper.hpp:228-232: only keys with hits are cached and written
                cache.add(key)
                k = format_keys(nl)
                lines.extend(k + "\t" + format_keys(sv) for sv in kept)
    return lines


def run(nl_path: str, sv_path: str, dis: int = 1000000, use_strand: bool = True):
    """-> dict(dup=[lines], inv=[lines], tra=[lines], sv_error=..., nl_error=...)."""
    nl_chroms, nl_recs, nl_err = read_vcf(nl_path, "nls")
    _, sv_recs, sv_err = read_vcf(sv_path, "delly")
    out = dict(dup=[], inv=[], tra=[], sv_error=sv_err, nl_error=nl_err)
    if sv_err is not None:
        # every chromosome task re-reads the SV file and dies at the bad record (exception swallowed by the pool,
        # thread_pool.hpp:46-50): no DUP/INV output. TraMapper::build_sv_tree throws on the main thread.
        return out

    def per_chrom_tree(sv_type):
        def tree_of(chrom):
            items = [validate_record(r) for r in sv_recs if r.chrom == chrom and r.svtype == sv_type]  # mapper.hpp:153-156
            if not items:
                return None, items
            return ivt.OracleTree(np.array([r.pos for r in items], np.uint32),
                                  np.array([r.svend for r in items], np.uint32)), items
        return tree_of

    out["dup"] = _map(nl_recs, nl_chroms, per_chrom_tree("DUP"), "TDUP", check_dup, dis, use_strand)
    out["inv"] = _map(nl_recs, nl_chroms, per_chrom_tree("INV"), "INV", check_inv, dis, use_strand)
    bnd = [r for r in sv_recs if r.svtype == "BND"]  # mapper.cpp:158-170: not validated, all chromosomes
    tra_tree = ivt.OracleTree(np.array([r.pos for r in bnd], np.uint32), np.array([r.svend for r in bnd], np.uint32)) if bnd else None
    out["tra"] = _map(nl_recs, nl_chroms, lambda chrom: (tra_tree, bnd), "TRA", check_tra, dis, use_strand)
    return out
