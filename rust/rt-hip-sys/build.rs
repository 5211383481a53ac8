fn main() {
    // librtmi.so is built by `make -C raytracing_amd/csrc`; point RTMI_LIB_DIR at the directory holding it.
    let dir = std::env::var("RTMI_LIB_DIR").unwrap_or_else(|_| "../../raytracing_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rtmi");
}
