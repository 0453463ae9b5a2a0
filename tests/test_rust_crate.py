"""The cargo crate under rust/suffix-hip cannot be compiled here (no rustc / cargo in the image), so the CPU
suite checks what can go wrong silently: (1) every function in its `extern "C"` block exists in
include/suffix_hip.h with the same arity and argument / return types; (2) the crate has the files cargo needs;
(3) the patches apply to the reference checkout (`patch --dry-run`; skipped where /root/reference is absent,
i.e. on the GPU box)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "suffix-hip")

C2RUST = {
    "const uint8_t*": "*const u8", "uint8_t*": "*mut u8", "const uint32_t*": "*const u32", "uint32_t*": "*mut u32",
    "const uint64_t*": "*const u64", "uint64_t*": "*mut u64", "uint64_t": "u64", "uint32_t": "u32", "int": "c_int",
    "void*": "*mut c_void", "const char*": "*const c_char", "sfx_index**": "*mut *mut SfxIndex",
    "const sfx_index*": "*const SfxIndex", "sfx_index*": "*mut SfxIndex", "double*": "*mut f64", "void": None,
}


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def c_declarations():
    src = _strip_c_comments(open(os.path.join(ROOT, "include", "suffix_hip.h")).read())
    decls = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(sfx_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret or "enum" in ret:
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                ty = re.sub(r"\b[A-Za-z_]\w*$", "", a).strip() if not a.endswith("*") else a     # drop the name
                ty = ty.replace(" *", "*").replace("* ", "*")
                params.append(ty)
        decls[name] = (" ".join(ret.split()).replace(" *", "*"), params)
    return decls


def rust_externs():
    src = open(os.path.join(CRATE, "src", "lib.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    block = re.sub(r"//[^\n]*", "", block)
    block = re.sub(r"#\[[^\]]*\]", "", block)
    out = {}
    for m in re.finditer(r"fn\s+(sfx_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "").strip() or None
        params = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        out[name] = (ret, params)
    return out


def test_extern_block_matches_the_c_header():
    c, r = c_declarations(), rust_externs()
    assert len(r) >= 10, sorted(r)
    for name, (rret, rparams) in r.items():
        assert name in c, f"{name} is not declared in include/suffix_hip.h"
        cret, cparams = c[name]
        assert C2RUST[cret] == rret, (name, cret, rret)
        assert [C2RUST[p] for p in cparams] == rparams, (name, cparams, rparams)
    # the three seams of the drop-in boundary are bound
    for must in ("sfx_build_sa_u32", "sfx_build_lcp_u32", "sfx_positions_batch", "sfx_contains_batch",
                 "sfx_index_create", "sfx_index_destroy", "sfx_build_sa_lcp_u32"):
        assert must in r, must


def test_crate_is_cargo_ready():
    for f in ("Cargo.toml", "build.rs", os.path.join("src", "lib.rs"), "table.rs.patch", "Cargo.toml.patch"):
        assert os.path.exists(os.path.join(CRATE, f)), f
    manifest = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert 'name = "suffix-hip"' in manifest and 'build = "build.rs"' in manifest and 'links = "suffix_hip"' in manifest
    build = open(os.path.join(CRATE, "build.rs")).read()
    assert "rustc-link-lib=dylib=suffix_hip" in build and "SUFFIX_HIP_LIB_DIR" in build
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    for fn in ("pub fn sais_table(text: &str) -> Vec<u32>", "pub fn lcp_lens(text: &str, table: &[u32]) -> Vec<u32>",
               "pub fn positions_batch", "impl Drop for DeviceIndex"):
        assert fn in lib, fn
    # the patch only touches the two private seams + the additive API, behind the `hip` feature
    patch = open(os.path.join(CRATE, "table.rs.patch")).read()
    added = [l[1:] for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++")]
    removed = [l for l in patch.splitlines() if l.startswith("-") and not l.startswith("---")]
    assert not removed, "the patch must not delete reference code (the CPU path stays)"
    assert sum('cfg(feature = "hip")' in l for l in added) == 6
    assert any("::suffix_hip::sais_table(text)" in l for l in added)
    assert any("::suffix_hip::lcp_lens(self.text(), self.table())" in l for l in added)
    # short texts stay on the reference's own CPU path (both seams compare against the crate's threshold) ...
    assert sum("::suffix_hip::min_device_len()" in l for l in added) == 2
    assert "pub const MIN_DEVICE_LEN: usize" in lib and "pub fn min_device_len() -> usize" in lib
    # ... except in a test run: the upstream tests (tests/tests.rs: QuickCheck strings < 100 B, literals) are all far below
    # the production threshold, so the run must be able to set it to 0 -- environment (read once) or cargo feature
    assert 'std::env::var("SUFFIX_HIP_MIN_LEN")' in lib and 'cfg!(feature = "always")' in lib
    assert "MIN_DEVICE_LEN" not in patch, "the seams must compare against min_device_len(), not the constant"
    assert "always = []" in manifest
    cargo_patch = open(os.path.join(CRATE, "Cargo.toml.patch")).read()
    assert 'hip-always = ["hip", "suffix-hip/always"]' in cargo_patch
    readme = open(os.path.join(CRATE, "README.md")).read()
    assert "SUFFIX_HIP_MIN_LEN=0" in readme and "--features hip-always" in readme
    # ... and the resident index outlives a batch: the caller holds the handle, positions_batch builds none per call
    assert any("pub fn device_index(&self) -> ::suffix_hip::DeviceIndex" in l for l in added)
    assert any("pub fn positions_batch_on<'a>(" in l for l in added)
    body = patch[patch.index("pub fn positions_batch_on"):patch.index("pub fn positions_batch<'a>")]
    assert "DeviceIndex::new" not in body


@pytest.mark.skipif(not os.path.isdir("/root/reference/src") or shutil.which("patch") is None,
                    reason="reference checkout / patch(1) not available")
def test_patches_apply_to_the_reference():
    for p in ("table.rs.patch", "Cargo.toml.patch"):
        out = subprocess.run(["patch", "--dry-run", "-p1", "-d", "/root/reference", "-i", os.path.join(CRATE, p)],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "FAILED" not in out.stdout and "fuzz" not in out.stdout, out.stdout


# ---- the upstream suite through the FFI, wherever a Rust toolchain exists ----------------
# /root/reference/tests/tests.rs:14-96 (SA = naive_table on 10 literals + 2 QuickCheck properties), :100-243 (positions / contains
# known answers + 3 properties) and Cargo.toml:25-37 ([[test]] tests, dev-dependency quickcheck 0.9).  Needs: cargo, an upstream
# checkout (SUFFIX_REFERENCE_DIR, default /root/reference), a built libsuffix_hip.so and a GPU.  The image this repository is
# built in has no cargo and its GPU boxes have neither cargo nor the checkout: the test is collected under `-m gpu` and skips there.
def _reference_dir():
    return os.environ.get("SUFFIX_REFERENCE_DIR", "/root/reference")


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("cargo") is None, reason="no cargo on this box (the build image and the pool's GPU boxes have none)")
@pytest.mark.skipif(not os.path.isdir(os.path.join(_reference_dir(), "src")), reason="no upstream checkout (SUFFIX_REFERENCE_DIR)")
def test_upstream_cargo_tests_through_the_ffi(tmp_path):
    lib_dir = os.path.join(ROOT, "suffix_amd")
    assert os.path.exists(os.path.join(lib_dir, "libsuffix_hip.so")), "build the engine first (__graft_entry__.build())"
    work = tmp_path / "suffix"
    shutil.copytree(_reference_dir(), work, ignore=shutil.ignore_patterns("target", ".git"))
    for p in ("table.rs.patch", "Cargo.toml.patch"):
        subprocess.run(["patch", "-p1", "-d", str(work), "-i", os.path.join(CRATE, p)], check=True, capture_output=True)
    # Cargo.toml.patch assumes sibling checkouts; point the dependency at this tree
    manifest = (work / "Cargo.toml").read_text().replace('path = "../suffix_amd/rust/suffix-hip"', f'path = "{CRATE}"')
    (work / "Cargo.toml").write_text(manifest)
    env = dict(os.environ, SUFFIX_HIP_LIB_DIR=lib_dir, CARGO_TARGET_DIR=str(tmp_path / "target"))
    # offline boxes: `cargo vendor` output or a primed ~/.cargo/registry is used as is (rust/suffix-hip/README.md)
    offline = ["--offline"] if os.environ.get("CARGO_NET_OFFLINE") == "true" else []
    out = subprocess.run(["cargo", "test", "--features", "hip-always", *offline, "--", "--test-threads", "4"], cwd=work, env=env,
                         capture_output=True, text=True, timeout=3000)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    # all 31 upstream tests ran (tests/tests.rs) and every one crossed extern "C" (the `always` feature: threshold 0)
    m = re.search(r"test result: ok\. (\d+) passed; 0 failed", out.stdout)
    assert m and int(m.group(1)) >= 31, out.stdout[-2000:]
    # the benches compile against the patched crate too (tests/bench.rs:9-133); not run here
    out = subprocess.run(["cargo", "bench", "--features", "hip", *offline, "--no-run"], cwd=work, env=env, capture_output=True,
                         text=True, timeout=3000)
    assert out.returncode == 0, out.stderr[-4000:]
