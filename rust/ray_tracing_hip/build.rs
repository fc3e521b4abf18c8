// Builds ray_tracing_amd/csrc (hipcc --offload-arch=gfx950 --genco for the code object, g++ for the C-ABI
// library that embeds it) and links librt_hip.so.  RT_HIP_CSRC overrides the location of csrc/.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let csrc = env::var("RT_HIP_CSRC").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../ray_tracing_amd/csrc")
    });
    let ok = Command::new("make").arg("-C").arg(&csrc).arg("librt_hip.so").status().expect("running make").success();
    assert!(ok, "building librt_hip.so in {} failed", csrc.display());
    println!("cargo:rustc-link-search=native={}", csrc.display());
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", csrc.display());
    for f in ["rt_kernels.hip", "rt_hip.cpp", "rt_host_helpers.cpp", "rt_device.h"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
}
