// build.rs fragment for the `hip` feature of dusk-hades (SOURCE ONLY -- not compiled in this
// image).  Points rustc at libhades252.so built by `python -m hades252_amd.build`.
fn main() {
    if std::env::var_os("CARGO_FEATURE_HIP").is_some() {
        let dir = std::env::var("HADES252_LIB_DIR").unwrap_or_else(|_| "/usr/local/lib".into());
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-lib=dylib=hades252");
        println!("cargo:rerun-if-env-changed=HADES252_LIB_DIR");
    }
}
