// build.rs -- link against libsuffix_hip.so (built by `make -C suffix_amd/csrc` or
// `python -c "import __graft_entry__ as g; g.build()"` in the suffix_amd repository).
//
//   SUFFIX_HIP_LIB_DIR   directory that holds libsuffix_hip.so   (default: ../../suffix_amd, i.e. this
//                        crate sitting in <suffix_amd repo>/rust/suffix-hip)
//   ROCM_PATH            ROCm installation whose libamdhip64.so the library needs (default /opt/rocm)
use std::env;
use std::path::PathBuf;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let lib_dir = env::var("SUFFIX_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| manifest.join("..").join("..").join("suffix_amd"));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rerun-if-env-changed=SUFFIX_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=suffix_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    // so that `cargo test` finds the shared objects without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/lib", rocm);
}
