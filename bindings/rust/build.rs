// Links libgymrs_amd.so (built by `python gym-rs_amd/build.py`).  GYMRS_AMD_LIB_DIR overrides the default
// location, which is the package directory of this repository.
fn main() {
    let dir = std::env::var("GYMRS_AMD_LIB_DIR").unwrap_or_else(|_| {
        let manifest = std::env::var("CARGO_MANIFEST_DIR").expect("cargo sets CARGO_MANIFEST_DIR");
        format!("{manifest}/../../gym-rs_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=gymrs_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=GYMRS_AMD_LIB_DIR");
}
