// Link against libnnnoiseless_mi355x.so (built by `python -c 'import __graft_entry__ as g; g.build()'`).
fn main() {
    let dir = std::env::var("NNN_MI355X_LIB_DIR").unwrap_or_else(|_| "../../nnnoiseless_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=nnnoiseless_mi355x");
}
