// Points the linker at libarrow_hip.so: ARROW_HIP_LIB_DIR, or the in-tree build output
// (make -C arrow-rs_amd/csrc writes arrow-rs_amd/lib/libarrow_hip.so).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("ARROW_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../arrow-rs_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=arrow_hip");
    println!("cargo:rerun-if-env-changed=ARROW_HIP_LIB_DIR");
}
