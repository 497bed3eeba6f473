// Links libvfsx.so (built by vfs_amd/csrc/Makefile).  Set VFSX_LIB_DIR to vfs_amd/lib.
fn main() {
    if let Ok(dir) = std::env::var("VFSX_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=vfsx");
}
