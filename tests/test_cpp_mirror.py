"""Builds and runs tests/cpp/mirror_test.cpp: the C++ host mirror of the reference interface (include/labrador.hpp)
against the oracle, on the GPU.  The header must also compile as plain C++ on a machine without a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp):
    from oracle import coracle as co
    co.build()
    exe = os.path.join(tmp, "mirror_test")
    libdir, orcdir = os.path.join(ROOT, "labrador_snark_amd"), os.path.join(ROOT, "oracle")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "mirror_test.cpp"), "-o", exe,
           "-L" + libdir, "-llabrador_hip", "-L" + orcdir, "-llabrador_oracle", "-Wl,-rpath," + libdir,
           "-Wl,-rpath," + orcdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_mirror_compiles(tmp_path):
    build(str(tmp_path))


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_mirror_runs_against_oracle(tmp_path):
    exe = build(str(tmp_path))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout
