// build.rs for the reference's Rust host when it links libpt_hip.so instead of building the OptiX wrapper
// (the reference's build.rs:1-72 drives cmake/nvcc and links `optix_wrapper`, build.rs:44).
//
// NOT compiled in the environment this backend was built in (no cargo there); src/main.rs and
// src/guiding.rs of the reference stay as they are -- their extern "C" blocks already name the symbols this
// library exports.
//
//   PT_HIP_DIR   directory holding libpt_hip.so (default: rust-pathtracer_amd next to Cargo.toml)
//   PT_HIP_MAKE  set to 1 to run `make -C $PT_HIP_DIR libpt_hip.so` first (needs hipcc)
use std::{env, path::PathBuf, process::Command};

fn main() {
    let dir = env::var("PT_HIP_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("rust-pathtracer_amd")
    });
    if env::var("PT_HIP_MAKE").map(|v| v == "1").unwrap_or(false) {
        let status = Command::new("make")
            .arg("-C")
            .arg(&dir)
            .arg("libpt_hip.so")
            .status()
            .expect("could not run make");
        assert!(status.success(), "building libpt_hip.so failed");
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=pt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=PT_HIP_DIR");
    println!("cargo:rerun-if-changed={}", dir.join("libpt_hip.so").display());
}
