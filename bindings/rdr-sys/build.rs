// Links librdr_hip.so (built by `make -C renderer_amd/csrc`).  RDR_HIP_LIB_DIR names its directory;
// the default is the in-tree location relative to this crate.
fn main() {
    let dir = std::env::var("RDR_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../renderer_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rdr_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RDR_HIP_LIB_DIR");
}
