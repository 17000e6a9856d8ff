// Links libkokorox_hip.so, which is built from kokorox_amd/csrc/*.hip by `python -m kokorox_amd.build`
// (hipcc --offload-arch=gfx950).  KOKOROX_HIP_LIB_DIR names the directory holding it
// (default: ../kokorox_amd/lib relative to this crate).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("KOKOROX_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("kokorox_amd").join("lib")
    });
    println!("cargo:rerun-if-env-changed=KOKOROX_HIP_LIB_DIR");
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=kokorox_hip");
    // let binaries find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
