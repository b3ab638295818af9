// Drop-in replacement for the reference's src/saca.rs (23 lines): same public items, same
// asserts, the engine call goes to libsuffix_array_amd.so instead of cdivsufsort.
// Not compiled in this repository's CI (no Rust toolchain on the build image); kept minimal
// so that review suffices.  Build: add `println!("cargo:rustc-link-lib=dylib=suffix_array_amd");`
// to a build.rs (and `cargo:rustc-link-search=native=<dir of the .so>`), drop the
// `cdivsufsort` dependency from Cargo.toml.

/// Maximum length of the input string.
pub const MAX_LENGTH: usize = std::i32::MAX as usize;

extern "C" {
    // include/suffix_array_amd.h: same signature as libdivsufsort's `divsufsort`
    fn sa_amd_divsufsort(t: *const u8, sa: *mut i32, n: i32) -> i32;
}

/// Wrapper of the underlying suffix array construction algorithm.
pub fn saca(s: &[u8], sa: &mut [u32]) {
    assert!(s.len() <= MAX_LENGTH);
    assert_eq!(s.len() + 1, sa.len());

    sa[0] = s.len() as u32;
    let ret = unsafe { sa_amd_divsufsort(s.as_ptr(), sa[1..].as_mut_ptr() as *mut i32, s.len() as i32) };
    assert_eq!(ret, 0, "suffix_array_amd engine failed with status {}", ret);
}
