// Builds libvgpu.so with the repository's Makefile (hipcc --offload-arch=gfx950) and tells cargo where to find it.
// Not exercised in this repository's CI: the build environment has no Rust toolchain.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../..").canonicalize().unwrap();
    let status = Command::new("make").arg("-C").arg(&root).arg("lib").status().expect("make");
    assert!(status.success(), "building libvgpu.so failed");
    println!("cargo:rustc-link-search=native={}", root.join("valida_amd").display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", root.join("valida_amd").display());
    println!("cargo:rerun-if-changed={}", root.join("include/vgpu.h").display());
}
