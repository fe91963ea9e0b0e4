//! Tell cargo where libidsp_hip.so lives.  IDSP_HIP_LIB_DIR = the directory holding the shared object
//! (in this repository: idsp_amd/lib, built by `make lib`); default: ../../idsp_amd/lib relative to this crate.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var_os("IDSP_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").unwrap()).join("../../idsp_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=idsp_hip");
    // consumers run tests / examples without installing the library
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=IDSP_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
