"""The C++ host mirror of the reference API (include/suffix_table.hpp) over the C
ABI: tests/cpp/test_suffix_table.cpp restates the reference's tests/tests.rs.
CPU: compiled and linked against the product library (link check) and RUN against
the emulator build of the same ABI.  GPU: run against libsuffix_hip.so."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "cpp", "test_suffix_table.cpp")


def _build(tmp_path, libdir, libname):
    exe = str(tmp_path / f"test_st_{libname}")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", libdir, f"-l{libname}", f"-Wl,-rpath,{libdir}", "-pthread", "-o", exe])
    return exe


def test_cpp_mirror_runs_on_emulator(tmp_path):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-j8", "-C", emu])
    exe = _build(tmp_path, emu, "suffix_emu")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


def test_cpp_mirror_links_against_product_library(tmp_path):
    lib = os.path.join(ROOT, "suffix_amd")
    if not os.path.exists(os.path.join(lib, "libsuffix_hip.so")):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(lib, "csrc")])
    _build(tmp_path, lib, "suffix_hip")


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(tmp_path):
    exe = _build(tmp_path, os.path.join(ROOT, "suffix_amd"), "suffix_hip")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, SFX_CPP_THREADS="1"))
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr
