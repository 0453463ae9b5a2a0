"""CPU tests of the drop-in boundary: the product library (hipcc, gfx950) loads
without a GPU and exports every symbol include/suffix_hip.h declares; the ctypes
binding covers exactly that set; compute calls fail loudly without a device."""
import os
import re
import subprocess

import numpy as np
import pytest

import suffix_amd
from suffix_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_engine():
    if not os.path.exists(_lib.DEFAULT_LIB):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "suffix_amd", "csrc")])
    return suffix_amd.Engine()


def _header_functions():
    src = open(os.path.join(ROOT, "include", "suffix_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sfx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    declared = _header_functions()
    bound = sorted(name for name, _, _ in _lib.ABI)
    assert declared == bound


def test_library_exports_every_declared_symbol(hip_engine):
    for name in _header_functions():
        assert hasattr(hip_engine.lib, name), name
    assert os.path.basename(hip_engine.path) == "libsuffix_hip.so"


def test_library_contains_gfx950_code_objects():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", _lib.DEFAULT_LIB],
                         capture_output=True, text=True)
    assert "gfx950" in (out.stdout + out.stderr)


def test_pure_host_entry_points(hip_engine):
    lib = hip_engine.lib
    assert lib.sfx_strerror(0) == b"ok"
    assert lib.sfx_sa_workspace_bytes(1000) >= 40 * 1000
    assert lib.sfx_lcp_workspace_bytes(1000) >= 4 * 1000
    assert lib.sfx_build_sa_u32(None, 0, None) == 0                 # empty text is fine (:395-396)
    assert lib.sfx_build_sa_u32(None, 1 << 32, None) == 2           # > u32::MAX (:380)


def test_fails_loudly_without_gpu(hip_engine):
    if hip_engine.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(suffix_amd.SuffixHipError):
        suffix_amd.SuffixTable("banana")
    t = np.frombuffer(b"banana", dtype=np.uint8)
    sa = np.zeros(6, dtype=np.uint32)
    assert hip_engine.lib.sfx_build_sa_u32(t.ctypes.data, 6, sa.ctypes.data) == 3   # NO_DEVICE


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(suffix_amd.SuffixHipError):
        suffix_amd.Engine(str(tmp_path / "nope.so"))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "suffix_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower(), f"{f} mentions the oracle"
