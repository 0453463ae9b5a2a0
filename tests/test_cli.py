"""The `suffix-array <file>` driver (tools/suffix_array.cpp; reference src/main.rs:8-15)
over the C++ host mirror.  CPU: linked against the emulator build of the ABI; GPU: against
libsuffix_hip.so.  Expected values: tests/golden."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "tools", "suffix_array.cpp")


def _build(tmp_path, libdir, libname):
    exe = str(tmp_path / f"suffix-array-{libname}")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", libdir, f"-l{libname}", f"-Wl,-rpath,{libdir}", "-o", exe])
    return exe


def _exercise(exe, tmp_path, fasta, golden):
    name = "AP009048_10000"
    g = golden["fixtures"][name]
    text = bytes(fasta[name])
    path = tmp_path / "in.fasta"
    path.write_bytes(text)
    pre = str(tmp_path / "dump")
    out = subprocess.run([exe, str(path), "--lcp", "--dump", pre, "--query", "ACTTACGTGTCTGC", "--query", "H",
                          "--query", "C"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0] == f"Suffixes: {len(text)}"                         # src/main.rs:14
    assert f"LCP: max {g['max_lcp']} " in lines[1]
    assert lines[2] == 'positions("ACTTACGTGTCTGC"): 1 [1825]'
    assert lines[3] == 'positions("H"): 0'
    assert lines[4].startswith(f'positions("C"): {text.count(b"C")} ')
    sa = np.fromfile(pre + ".sa", dtype="<u4")
    lcp = np.fromfile(pre + ".lcp", dtype="<u4")
    assert hashlib.sha256(sa.tobytes()).hexdigest() == g["sha256_sa"]
    assert hashlib.sha256(lcp.tobytes()).hexdigest() == g["sha256_lcp"]
    # from_parts reload (:111-119): no construction, same answers
    out2 = subprocess.run([exe, str(path), "--load", pre, "--query", "ACTTACGTGTCTGC"], capture_output=True,
                          text=True, timeout=600)
    assert out2.returncode == 0 and 'positions("ACTTACGTGTCTGC"): 1 [1825]' in out2.stdout
    # errors: missing file -> 1; length mismatch on from_parts -> 2 (the reference panics, :117)
    assert subprocess.run([exe, str(tmp_path / "nope")], capture_output=True).returncode == 1
    (tmp_path / "short.txt").write_bytes(b"abc")
    bad = subprocess.run([exe, str(tmp_path / "short.txt"), "--load", pre], capture_output=True, text=True)
    assert bad.returncode == 2 and "len" in bad.stderr


def test_cli_on_emulator(tmp_path, fasta, golden):
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-s", "-j8", "-C", emu])
    _exercise(_build(tmp_path, emu, "suffix_emu"), tmp_path, fasta, golden)


@pytest.mark.gpu
def test_cli_on_gpu(tmp_path, fasta, golden):
    _exercise(_build(tmp_path, os.path.join(ROOT, "suffix_amd"), "suffix_hip"), tmp_path, fasta, golden)
