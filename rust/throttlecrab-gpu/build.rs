// Links libtcgpu.so (built by `make -C throttlecrab_amd/csrc`, hipcc --offload-arch=gfx950).
// TCGPU_LIB_DIR points at the directory holding it; default: the in-tree build.
fn main() {
    let dir = std::env::var("TCGPU_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../throttlecrab_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=tcgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=TCGPU_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/tcgpu.h");
}
