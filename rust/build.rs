// haplotyper/build.rs
// NOT compiled in this repository: the build image has no Rust toolchain (cargo, rustc: command not found) and the
// reference's git dependencies are un-vendored.  Source a jtk maintainer adds to ban-m/jtk; INTEGRATION.md explains it and
// tests/test_rust_shim_source.py keeps it in step with include/jtk_lc.h.  The same call sequence is exercised end to end
// by the C++ host mirror (jtk_amd/csrc/host/local_clustering.hpp) and the Python harness (jtk_amd/api.py).
fn main() {
    // libjtk_lc.so is built by `python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950)
    println!("cargo:rustc-link-search=native={}", std::env::var("JTK_LC_LIB_DIR").unwrap());
    println!("cargo:rustc-link-lib=dylib=jtk_lc");
}
