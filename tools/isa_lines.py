#!/usr/bin/env python3
"""Static instruction counts of one kernel by source line, from `hipcc -S -gline-tables-only` output.

usage: tools/isa_lines.py <file.s> <kernel-name-substring> [--by-block]
Prints, per (file, line) of the .loc directives inside the kernel, the number of vector (v_*), scalar (s_*), LDS (ds_*)
and memory (global_/buffer_/flat_/scratch_) instructions; with --by-block the same per basic block in program order
(loops show up as blocks with a backward branch). Static counts: a block inside a loop executes many times.
"""
import collections
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    by_block = "--by-block" in sys.argv
    files = {}
    cur = None
    inside = False
    counts = collections.OrderedDict()
    blocks = []
    blk = None
    with open(path) as f:
        for raw in f:
            line = raw.strip()
            m = re.match(r"\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", line)
            if m:
                files[int(m.group(1))] = m.group(2)
                continue
            if not inside:
                if re.match(r"^_Z\w*:", line) and name in line.split(":")[0]:
                    inside = True
                    blk = [line.split(":")[0][:40], collections.Counter(), None]
                    blocks.append(blk)
                continue
            if line.startswith(".Lfunc_end") or line.startswith("s_endpgm") and False:
                break
            m = re.match(r"\.loc\s+(\d+)\s+(\d+)", line)
            if m:
                cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
                continue
            if re.match(r"^\.LBB\d+_\d+:", line):
                blk = [line.split(":")[0], collections.Counter(), None]
                blocks.append(blk)
                continue
            if not line or line.startswith(";") or line.startswith("."):
                continue
            op = line.split()[0]
            if op.startswith("v_"):
                kind = "v"
            elif op.startswith("s_"):
                kind = "s"
                if op.startswith("s_cbranch") or op == "s_branch":
                    blk[2] = line.split()[-1]
            elif op.startswith("ds_"):
                kind = "ds"
            elif op.split("_")[0] in ("global", "buffer", "flat", "scratch"):
                kind = "mem"
            else:
                kind = "other"
            counts.setdefault(cur, collections.Counter())[kind] += 1
            blk[1][kind] += 1
    if by_block:
        for b in blocks:
            c = b[1]
            print(f"{b[0]:<14} v {c['v']:4d}  s {c['s']:4d}  ds {c['ds']:3d}  mem {c['mem']:3d}   -> {b[2] or ''}")
    else:
        tot = collections.Counter()
        for k in sorted(counts, key=lambda k: (str(k[0]), k[1])) if None not in counts else counts:
            c = counts[k]
            tot.update(c)
            print(f"{str(k[0]):<18}:{k[1]:<5d} v {c['v']:4d}  s {c['s']:4d}  ds {c['ds']:3d}  mem {c['mem']:3d}")
        print("total", dict(tot))


if __name__ == "__main__":
    main()
