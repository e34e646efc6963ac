// build.rs of micro-raytracer with the `hip` feature: link libmrt_hip.so (built by `make -C micro_raytracer_amd/csrc`).
fn main() {
    if std::env::var("CARGO_FEATURE_HIP").is_ok() {
        let dir = std::env::var("MRT_HIP_LIB_DIR").expect("MRT_HIP_LIB_DIR = directory holding libmrt_hip.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=mrt_hip");
        println!("cargo:rerun-if-env-changed=MRT_HIP_LIB_DIR");
    }
}
