// Links librttnw_hip.so (make -C rttnw_amd/csrc).  RTTNW_HIP_LIB_DIR overrides the default in-tree location.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("RTTNW_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../rttnw_amd/csrc")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rttnw_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RTTNW_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/rttnw_hip.h");
}
