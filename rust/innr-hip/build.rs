// Link against libinnr_hip.so built by `make -C innr_amd/csrc` (hipcc --offload-arch=gfx950).
fn main() {
    let dir = std::env::var("INNR_HIP_LIB_DIR").unwrap_or_else(|_| "../../innr_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=innr_hip");
    println!("cargo:rerun-if-env-changed=INNR_HIP_LIB_DIR");
}
