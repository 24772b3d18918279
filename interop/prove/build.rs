fn main() {
    let dir = std::env::var("H2MI_LIB_DIR").expect("set H2MI_LIB_DIR to the directory that holds libh2mi.so (halo2-scaffold_amd/)");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=h2mi");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
