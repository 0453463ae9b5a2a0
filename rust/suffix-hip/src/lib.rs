//! suffix-hip -- the Rust side of the drop-in boundary of the MI355X suffix-array engine.
//!
//! A cargo crate: `rust/suffix-hip/{Cargo.toml, build.rs, src/lib.rs}` plus two patches for a checkout
//! of BurntSushi/suffix v1.3.0 (`table.rs.patch`, `Cargo.toml.patch`).  With them applied,
//!
//!     SUFFIX_HIP_LIB_DIR=<dir of libsuffix_hip.so> cargo test --features hip
//!
//! runs the upstream `tests/tests.rs` (31 tests, 5 QuickCheck properties) with `SuffixTable::new`,
//! `lcp_lens` and the additive `positions_batch` going through `libsuffix_hip.so`.
//! NOT COMPILED IN THIS REPOSITORY'S CI: the build image has no rustc/cargo; what is
//! checked here, on the CPU, is that the patches apply to the reference sources and that every
//! signature in the `extern "C"` block below equals its declaration in `include/suffix_hip.h`
//! (tests/test_rust_crate.py).  The ctypes binding in `suffix_amd/_lib.py` and the C++ mirror
//! exercise the same symbols with the same argument meaning on the GPU box.
//!
//! What changes in the crate (nothing in `:140-312` or `lib.rs` changes):
//!   * `sais_table`  (src/table.rs:378-386)  body after `vec![0u32; n]`
//!   * `lcp_lens`    (src/table.rs:130-138)  body
//!   * new additive  `SuffixTable::device_index()` (the resident index, kept by the caller across batches),
//!     `positions_batch_on(&index, queries)` and the one-shot `positions_batch(queries)`
//!   * texts below `min_device_len()` bytes keep the crate's own CPU path.  The default is `MIN_DEVICE_LEN` = 64 KiB (a trip
//!     to the device costs ~100 us whatever the length): right for production, but every string of the upstream tests --
//!     QuickCheck's (< 100 bytes) and the literals of `tests/tests.rs` -- is far below it.  A TEST RUN THEREFORE SETS
//!     `SUFFIX_HIP_MIN_LEN=0` (read once per process) or builds with `--features hip-always`: then all 31 upstream tests
//!     cross the FFI, the empty text and the one-byte text included (tests/tests.rs:42-65).
//! `sais()`, `Bins`, `SuffixTypes` stay in the crate as the reference CPU path (feature `hip` off).

use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct SfxIndex {
    _private: [u8; 0],
}

extern "C" {
    fn sfx_strerror(status: c_int) -> *const c_char;
    fn sfx_last_hip_error() -> *const c_char;
    fn sfx_device_count() -> c_int;
    // SuffixTable::new -> sais_table
    fn sfx_build_sa_u32(text: *const u8, n: u64, sa_out: *mut u32) -> c_int;
    // lcp_lens
    fn sfx_build_lcp_u32(text: *const u8, n: u64, sa: *const u32, lcp_out: *mut u32) -> c_int;
    // SuffixTable::new + lcp_lens in one engine call (what suffix_tree's to_suffix_tree needs)
    fn sfx_build_sa_lcp_u32(text: *const u8, n: u64, sa_out: *mut u32, lcp_out: *mut u32) -> c_int;
    // device-resident index for batched queries
    fn sfx_index_create(text: *const u8, n: u64, sa: *const u32, out: *mut *mut SfxIndex) -> c_int;
    fn sfx_index_destroy(ix: *mut SfxIndex);
    fn sfx_index_len(ix: *const SfxIndex) -> u64;
    fn sfx_positions_batch(ix: *const SfxIndex, qbytes: *const u8, qoff: *const u64, nq: u64,
                           start_out: *mut u32, end_out: *mut u32) -> c_int;
    fn sfx_contains_batch(ix: *const SfxIndex, qbytes: *const u8, qoff: *const u64, nq: u64,
                          found_out: *mut u8, any_out: *mut u32) -> c_int;
    // device-pointer variants (`*_dev`) take a hipStream_t as *mut c_void; omitted here.
    #[allow(dead_code)]
    fn sfx_build_sa_u32_dev(d_text: *const u8, n: u64, d_sa: *mut u32, ws: *mut c_void,
                            ws_bytes: u64, stream: *mut c_void) -> c_int;
}

/// Texts shorter than this stay on the crate's own CPU path (`sais`, `lcp_lens_quadratic`): a build on the device costs
/// ~100 us of launches and two PCIe copies whatever the length, the reference needs ~1 us for QuickCheck's strings
/// (README.md:116) and ~1 ms for 64 KiB.  The patched `sais_table` / `lcp_lens` compare against it.
pub const MIN_DEVICE_LEN: usize = 1 << 16;

/// The threshold the patched `sais_table` / `lcp_lens` compare against: `MIN_DEVICE_LEN`, unless the crate was built with the
/// `always` feature (the parent crate's `hip-always`: 0) or the environment names another one (`SUFFIX_HIP_MIN_LEN=<bytes>`,
/// read once per process).  `SUFFIX_HIP_MIN_LEN=0 cargo test --features hip` is how the upstream test-suite -- whose
/// strings are all far below 64 KiB -- is made to run on the device (tests/tests.rs:73-96).
pub fn min_device_len() -> usize {
    use std::sync::atomic::{AtomicUsize, Ordering};
    static CACHED: AtomicUsize = AtomicUsize::new(usize::MAX);
    let v = CACHED.load(Ordering::Relaxed);
    if v != usize::MAX {
        return v;
    }
    let default = if cfg!(feature = "always") { 0 } else { MIN_DEVICE_LEN };
    let parsed = std::env::var("SUFFIX_HIP_MIN_LEN")
        .ok()
        .and_then(|s| s.trim().parse::<usize>().ok())
        .map(|x| if x == usize::MAX { usize::MAX - 1 } else { x })
        .unwrap_or(default);
    CACHED.store(parsed, Ordering::Relaxed);
    parsed
}

fn check(status: c_int, what: &str) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(sfx_strerror(status)) }.to_string_lossy();
        let hip = unsafe { std::ffi::CStr::from_ptr(sfx_last_hip_error()) }.to_string_lossy();
        // the reference's error convention is panic (assert! at :380, assert_eq! at :117)
        panic!("{}: {} {}", what, msg, hip);
    }
}

/// Replacement for `fn sais_table(text: &str) -> Vec<u32>` (src/table.rs:378-386).
pub fn sais_table(text: &str) -> Vec<u32> {
    let text = text.as_bytes();
    assert!(text.len() <= u32::MAX as usize);              // :380, unchanged
    let mut sa = vec![0u32; text.len()];                   // :381, unchanged: caller allocates
    if unsafe { sfx_device_count() } <= 0 {
        panic!("suffix: no HIP device (build without the `hip` feature for the CPU path)");
    }
    check(unsafe { sfx_build_sa_u32(text.as_ptr(), text.len() as u64, sa.as_mut_ptr()) },
          "sfx_build_sa_u32");
    sa
}

/// Replacement for the body of `SuffixTable::lcp_lens` (src/table.rs:130-138).
pub fn lcp_lens(text: &str, table: &[u32]) -> Vec<u32> {
    let mut lcp = vec![0u32; table.len()];
    check(unsafe {
        sfx_build_lcp_u32(text.as_ptr(), text.len() as u64, table.as_ptr(), lcp.as_mut_ptr())
    }, "sfx_build_lcp_u32");
    lcp
}

/// `SuffixTable::new` and `lcp_lens` in one engine call: (table, lcp).
pub fn sais_table_with_lcp(text: &str) -> (Vec<u32>, Vec<u32>) {
    let text = text.as_bytes();
    assert!(text.len() <= u32::MAX as usize);
    let mut sa = vec![0u32; text.len()];
    let mut lcp = vec![0u32; text.len()];
    check(unsafe {
        sfx_build_sa_lcp_u32(text.as_ptr(), text.len() as u64, sa.as_mut_ptr(), lcp.as_mut_ptr())
    }, "sfx_build_sa_lcp_u32");
    (sa, lcp)
}

/// Additive API: many `positions()` at once.  Returns (start, end) pairs;
/// `positions(q_k) == &table[start_k as usize .. end_k as usize]` exactly as :244-258.
pub struct DeviceIndex(*mut SfxIndex);
unsafe impl Send for DeviceIndex {}
unsafe impl Sync for DeviceIndex {}          // queries only read the index

impl DeviceIndex {
    pub fn new(text: &str, table: &[u32]) -> DeviceIndex {
        let mut h: *mut SfxIndex = std::ptr::null_mut();
        check(unsafe { sfx_index_create(text.as_ptr(), text.len() as u64, table.as_ptr(), &mut h) },
              "sfx_index_create");
        DeviceIndex(h)
    }
    pub fn positions_batch(&self, queries: &[&str]) -> Vec<(u32, u32)> {
        let mut off = Vec::with_capacity(queries.len() + 1);
        let mut blob = Vec::new();
        off.push(0u64);
        for q in queries {
            blob.extend_from_slice(q.as_bytes());
            off.push(blob.len() as u64);
        }
        let (mut s, mut e) = (vec![0u32; queries.len()], vec![0u32; queries.len()]);
        check(unsafe {
            sfx_positions_batch(self.0, blob.as_ptr(), off.as_ptr(), queries.len() as u64,
                                s.as_mut_ptr(), e.as_mut_ptr())
        }, "sfx_positions_batch");
        s.into_iter().zip(e).collect()
    }
    pub fn contains_batch(&self, queries: &[&str]) -> Vec<bool> {
        let mut off = vec![0u64];
        let mut blob = Vec::new();
        for q in queries {
            blob.extend_from_slice(q.as_bytes());
            off.push(blob.len() as u64);
        }
        let mut f = vec![0u8; queries.len()];
        check(unsafe {
            sfx_contains_batch(self.0, blob.as_ptr(), off.as_ptr(), queries.len() as u64,
                               f.as_mut_ptr(), std::ptr::null_mut())
        }, "sfx_contains_batch");
        f.into_iter().map(|b| b != 0).collect()
    }
}
impl DeviceIndex {
    pub fn len(&self) -> usize {
        unsafe { sfx_index_len(self.0) as usize }
    }
}
impl Drop for DeviceIndex {
    fn drop(&mut self) {
        unsafe { sfx_index_destroy(self.0) }
    }
}
