// NEVER COMPILED (no Rust toolchain in the build image).
//
// Links libcrowdstep_hip.so, the HIP engine built by `python -c "import __graft_entry__ as g;
// g.build()"` into rmf_crowdsim_amd/lib/.  CROWDSTEP_LIB_DIR overrides the search path.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("CROWDSTEP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
        manifest.join("../../rmf_crowdsim_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=crowdstep_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=CROWDSTEP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/crowdstep.h");
}
