#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of libbivx.so, from hipcc -Rpass-analysis=kernel-resource-usage.
One line per kernel: file, demangled name, VGPRs, SGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS bytes
per workgroup. tests/test_capi_cpu.py asserts on this that no query kernel spills (ScratchSize == 0).
usage: tools/resource_usage.py [file.hip ...]   (default: every translation unit with kernels)"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "binary_amd", "csrc")
DEFAULT = ("query_fused.hip", "query_pipe.hip", "query.hip", "build.hip", "scan.hip")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return list(names)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(bivx::IndexView.*|\((unsigned|HIP_vector|bivx::BinStats|void).*", "", name)


def usage(files=DEFAULT, extra=()):
    rows = []
    for f in files:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", *extra, "-c", f, "-o",
               "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
        cur = {}
        for line in err.splitlines():
            m = re.search(r"remark: (?:\S+: )?\s*(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|"
                          r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            k, val = m.group(1), m.group(2)
            if k == "Function Name":
                cur = {"file": f, "mangled": val}
            elif k == "TotalSGPRs":
                cur["sgpr"] = int(val)
            elif k == "VGPRs":
                cur["vgpr"] = int(val)
            elif k.startswith("Scratch"):
                cur["scratch"] = int(val)
            elif k.startswith("Occupancy"):
                cur["occupancy"] = int(val)
            elif k.startswith("LDS"):
                cur["lds"] = int(val)
                rows.append(cur)
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = short(n)
    return rows


if __name__ == "__main__":
    for r in sorted(usage(sys.argv[1:] or DEFAULT), key=lambda r: (r["file"], r["name"])):
        print(f"{r['file']:16s} {r['name']:64s} vgpr={r['vgpr']:3d} sgpr={r['sgpr']:3d} scratch={r['scratch']:3d} "
              f"occupancy={r['occupancy']} lds={r['lds']}")
