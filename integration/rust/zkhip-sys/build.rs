// Links libzkhip.so (built by `python -c "import __graft_entry__ as g; g.build()"` into halo2-zkcert_amd/).
fn main() {
    let dir = std::env::var("ZKHIP_LIB_DIR").expect("set ZKHIP_LIB_DIR to the directory holding libzkhip.so (…/halo2-zkcert_amd)");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ZKHIP_LIB_DIR");
}
