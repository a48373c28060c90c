// Links libcrescent_gpu.so.  CRESCENT_GPU_LIB_DIR = directory holding the library
// (`<amd repo>/crescent-credentials_amd/` after `python -c 'import __graft_entry__ as g; g.build()'`).
fn main() {
    println!("cargo:rerun-if-env-changed=CRESCENT_GPU_LIB_DIR");
    if let Ok(dir) = std::env::var("CRESCENT_GPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=crescent_gpu");
}
