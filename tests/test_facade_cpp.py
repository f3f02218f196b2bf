"""The C++20 drop-in headers (include/binary/algorithm/*.hpp) against the reference's own test cases,
re-expressed in tests/cpp/test_facade.cpp. CPU: compiles with g++ -std=c++20 against libbivx.so and runs the
host-side structure cases; GPU: runs everything (overlap queries go through the C ABI to the HIP kernels)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def facade_binary(tmp_path_factory):
    from binary_amd import _build
    _build.build_lib()
    out = str(tmp_path_factory.mktemp("facade") / "test_facade")
    libdir = os.path.join(ROOT, "binary_amd")
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_facade.cpp"), "-o", out, "-L", libdir, "-lbivx", "-pthread",
           f"-Wl,-rpath,{libdir}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


def test_facade_compiles_and_host_cases_pass(facade_binary):
    r = subprocess.run([facade_binary, "--no-gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout


@pytest.mark.gpu
def test_facade_reference_cases_on_gpu(facade_binary):
    r = subprocess.run([facade_binary], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failed" in r.stdout
