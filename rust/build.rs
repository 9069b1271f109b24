// build.rs fragment for the `hip` feature of dusk-hades (SOURCE ONLY -- not compiled in this image).
// Only the SEARCH PATH comes from here; the library itself is named once, by `#[link(name = "hades252")]` on the
// extern block of src/strategies/hip.rs.  HADES252_LIB_DIR = the directory of libhades252.so
// (`python -m hades252_amd.build` leaves it in hades252_amd/csrc/).
fn main() {
    if std::env::var_os("CARGO_FEATURE_HIP").is_some() {
        let dir = std::env::var("HADES252_LIB_DIR").unwrap_or_else(|_| "/usr/local/lib".into());
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rerun-if-env-changed=HADES252_LIB_DIR");
    }
}
