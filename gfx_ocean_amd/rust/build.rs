// Links libocean_hip.so (built by `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    let dir = std::env::var("OCEAN_HIP_LIB_DIR").unwrap_or_else(|_| "../".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=ocean_hip");
}
