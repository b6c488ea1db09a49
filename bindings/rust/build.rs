// UNVERIFIED SOURCE (no rustc in the build image).  Links libbn254_hip.so; set BN254_HIP_LIB_DIR to the directory holding it.
fn main() {
    if let Ok(dir) = std::env::var("BN254_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=bn254_hip");
}
