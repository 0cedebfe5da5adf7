// Links the prebuilt C-ABI library (make -C kornia-rs_amd).  KORNIA_HIP_LIB_DIR points at the directory holding
// libkornia_hip.so; the only runtime dependency of that library is libamdhip64.so (ROCm).
fn main() {
    if let Ok(dir) = std::env::var("KORNIA_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=kornia_hip");
    println!("cargo:rerun-if-env-changed=KORNIA_HIP_LIB_DIR");
}
