"""Builds libbivx.so (hand-written HIP, gfx950) in-tree with hipcc via binary_amd/csrc/Makefile."""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
# BIVX_LIB: a variant build to load instead (tools/build_variant.sh; A/B measurements only)
LIB_PATH = os.environ.get("BIVX_LIB") or os.path.join(_PKG, "libbivx.so")
_SOURCES = ("scan.hip", "build.hip", "query.hip", "query_fused.hip", "query_pipe.hip", "prefix_device.h", "query_device.h", "wave_device.h", "capi.hip", "sharded.cpp", "common.h",
            "Makefile")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in _SOURCES] + [os.path.join(_PKG, "..", "include", "bivx.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force: bool = False, jobs: int = 4) -> str:
    """Compile every HIP translation unit for gfx950 and link binary_amd/libbivx.so."""
    if os.environ.get("BIVX_LIB"):
        return LIB_PATH
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    if force or _stale():
        subprocess.check_call(["make", "-C", CSRC, "-s", f"-j{jobs}", "all"])
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libbivx.so was not produced by the build")
    return LIB_PATH
