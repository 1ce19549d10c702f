// UNBUILT SOURCE -- see ../README.md
// Links the two shared libraries built by `python -c "import __graft_entry__ as g; g.build()"`
// (hipcc --offload-arch=gfx950 for libzkp_mi355x.so, g++ for libzkp_toolbox.so).
use std::env;
use std::path::PathBuf;

fn main() {
    // ZKP_MI355X_LIB_DIR = the directory holding the two .so files (<repo>/zkp_amd by default)
    let dir = env::var("ZKP_MI355X_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../zkp_amd"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=zkp_toolbox");
    println!("cargo:rustc-link-lib=dylib=zkp_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=ZKP_MI355X_LIB_DIR");
}
