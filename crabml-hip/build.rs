// build.rs -- links libcrabml_hip.so (the C ABI declared in include/crabml_hip.h, mirrored by hand in src/ffi.rs;
// tests/test_rust_crate.py in the backend repository keeps the two in step: names, arities, integer widths).
//
// The library is built from the backend repository with `python -c 'import __graft_entry__ as g; g.build()'`
// (hipcc --offload-arch=gfx950) and lands in crabml_amd/libcrabml_hip.so.  Point CRABML_HIP_LIB_DIR at that
// directory (or install the library somewhere the linker already looks).  There is no fallback: without the
// library the crate does not link, and without a HIP device HipTensorDevice::new returns an error.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=CRABML_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
    if let Ok(dir) = env::var("CRABML_HIP_LIB_DIR") {
        let dir = PathBuf::from(dir);
        println!("cargo:rustc-link-search=native={}", dir.display());
        // the tests and the CLI find the library at run time without LD_LIBRARY_PATH
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    }
    println!("cargo:rustc-link-lib=dylib=crabml_hip");
}
