// Links libloupiote_hip.so.  LOUPIOTE_HIP_LIB_DIR names the directory that holds it (the repository's loupiote_amd/ after
// `python -c "import __graft_entry__ as g; g.build()"`); the default is that directory relative to this crate.
fn main() {
    let dir = std::env::var("LOUPIOTE_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../../loupiote_amd", here)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=loupiote_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=LOUPIOTE_HIP_LIB_DIR");
}
