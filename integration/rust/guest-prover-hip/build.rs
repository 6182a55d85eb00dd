// Locates libzkhip.so: ZKHIP_LIB_DIR (the directory that holds it, e.g. <zktls-hip>/zktls_amd) or the
// default search path.  The HIP runtime is a dependency of libzkhip itself, not of this crate.
fn main() {
    println!("cargo:rerun-if-env-changed=ZKHIP_LIB_DIR");
    if let Ok(dir) = std::env::var("ZKHIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=zkhip");
}
