//! Build script for the HIP back end: replaces the glslc step of the reference's build.rs (build.rs:1-31) on Linux.
//! Compiles the render core into `librt_hip.so` with hipcc for gfx950 and links the crate against it.
//!
//! Written without a Rust toolchain at hand (none in the build image): the C side it binds is checked mechanically
//! against `include/rt_api.h` by tests/test_rust_binding.py; this file itself has not been compiled.
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    // RT_HIP_ROOT points at the checkout that holds include/ and ray-tracer_amd/csrc/ (default: two levels up).
    let root = env::var("RT_HIP_ROOT").map(PathBuf::from).unwrap_or_else(|_| PathBuf::from("../.."));
    let csrc = root.join("ray-tracer_amd").join("csrc");
    let out = PathBuf::from(env::var("OUT_DIR").expect("OUT_DIR"));
    let lib = out.join("librt_hip.so");
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());

    let sources = ["rt_api.hip", "rt_shade_fast.hip", "scene_pack.cpp", "bvh_build.cpp", "bvh_sbvh.cpp", "bvh_sweep.cpp"];
    for s in sources.iter() {
        println!("cargo:rerun-if-changed={}", csrc.join(s).display());
    }
    for h in ["rt_kernels.h", "rt_kernels_common.h", "rt_shade.h", "rt_traverse.h", "rt_device_math.h", "rt_internal.h", "bvh_common.h"].iter() {
        println!("cargo:rerun-if-changed={}", csrc.join(h).display());
    }
    println!("cargo:rerun-if-changed={}", root.join("include").join("rt_api.h").display());
    println!("cargo:rerun-if-changed={}", root.join("include").join("rt_types.h").display());

    // -ffp-contract=off: intersection and shading arithmetic is specified operation by operation (no fused multiply-add).
    let status = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"])
        .arg(format!("-I{}", root.join("include").display()))
        .arg(format!("-I{}", csrc.display()))
        .args(sources.iter().map(|s| csrc.join(s)))
        .arg("-o")
        .arg(&lib)
        .arg("-lpthread")
        .status()
        .expect("could not run hipcc (set HIPCC to its path)");
    if !status.success() {
        panic!("hipcc failed to build librt_hip.so");
    }
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_hip");
}
