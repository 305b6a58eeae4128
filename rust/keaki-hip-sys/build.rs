// Links libkeaki_hip.so. KEAKI_HIP_LIB_DIR = the directory that holds it (keaki_amd/ in the MI355X tree, after
// `make -C keaki_amd/csrc`); the run-time search path is set to the same directory.
fn main() {
    println!("cargo:rerun-if-env-changed=KEAKI_HIP_LIB_DIR");
    let dir = std::env::var("KEAKI_HIP_LIB_DIR")
        .expect("set KEAKI_HIP_LIB_DIR to the directory that contains libkeaki_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=keaki_hip");
    if std::env::var("CARGO_FEATURE_RCCL").is_ok() {
        println!("cargo:rustc-link-lib=dylib=keaki_hip_rccl"); // the optional RCCL collectives (include/keaki_hip_rccl.h)
    }
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
