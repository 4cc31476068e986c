//! zkhip.rs — routes `msm::best_multiexp` and `fft::best_fft` of halo2curves 0.4.0 to libzkhip.so (MI355X) for bn256.
//! Add to `src/` of a checkout of axiom-crypto/halo2curves @ e185711 (see ../README.md for the two call-site edits).
//!
//! Layout facts this relies on (halo2curves 0.4.0): `bn256::Fr` / `Fq` are `#[repr(transparent)]` over `[u64; 4]` in Montgomery form
//! (R = 2^256); `bn256::G1Affine { x: Fq, y: Fq }` and `bn256::G1 { x, y, z }` are plain structs of those fields in that order, the
//! affine identity is (0, 0).  They are the ABI forms of include/zkhip.h, so slices cross the boundary without conversion.
use std::any::TypeId;
use std::collections::HashMap;
use std::sync::{Mutex, OnceLock};

use ff::Field;
use pasta_curves::arithmetic::CurveAffine;
use zkhip_sys as sys;

use crate::bn256::{Fr, G1Affine, G1};
use crate::fft::FftGroup;

/// Sizes below these stay on the CPU path: a host-pointer call pays an upload and a download.
const MIN_MSM: usize = 1 << 12;
const MIN_FFT_LOG_N: u32 = 12;

struct Ctx(*mut sys::zkhip_ctx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}

/// One context per process (device `ZKHIP_DEVICE`, default 0).  No GPU / library error is FATAL — a build patched with this module is
/// meant to run on the GPU and must not silently become a CPU prover — unless `ZKHIP_ALLOW_CPU=1` asks for upstream's own CPU body.
fn ctx() -> Option<*mut sys::zkhip_ctx> {
    static CTX: OnceLock<Option<Ctx>> = OnceLock::new();
    CTX.get_or_init(|| {
        let dev = std::env::var("ZKHIP_DEVICE").ok().and_then(|s| s.parse::<i32>().ok()).unwrap_or(0);
        let mut p: *mut sys::zkhip_ctx = std::ptr::null_mut();
        let rc = unsafe { sys::zkhip_init(&mut p, dev) };
        if rc == sys::ZKHIP_OK {
            Some(Ctx(p))
        } else if std::env::var("ZKHIP_ALLOW_CPU").as_deref() == Ok("1") {
            eprintln!("zkhip: disabled ({}); ZKHIP_ALLOW_CPU=1: upstream's CPU path runs", sys::last_error());
            None
        } else {
            panic!("zkhip: zkhip_init failed: {} (set ZKHIP_ALLOW_CPU=1 to run upstream's CPU path instead)", sys::last_error());
        }
    })
    .as_ref()
    .map(|c| c.0)
}

/// Device-resident window tables per SRS.  `ParamsKZG` keeps `g` and `g_lagrange` alive for the whole process and `commit` passes
/// `&bases[..len]`, so the key is the slice's address, its length and a fingerprint of its first and last point (an address reused
/// by another allocation with other contents misses and is reloaded; `evict_srs` drops an entry explicitly).
#[derive(Hash, PartialEq, Eq, Clone, Copy)]
struct SrsKey {
    ptr: usize,
    len: usize,
    first: [u64; 8],
    last: [u64; 8],
}
struct SrsHandle(*mut sys::zkhip_srs);
unsafe impl Send for SrsHandle {}

fn srs_cache() -> &'static Mutex<HashMap<SrsKey, SrsHandle>> {
    static CACHE: OnceLock<Mutex<HashMap<SrsKey, SrsHandle>>> = OnceLock::new();
    CACHE.get_or_init(|| Mutex::new(HashMap::new()))
}

fn point_words(p: &G1Affine) -> [u64; 8] {
    let mut w = [0u64; 8];
    unsafe { std::ptr::copy_nonoverlapping(p as *const G1Affine as *const u64, w.as_mut_ptr(), 8) };
    w
}

fn srs_for(ctx: *mut sys::zkhip_ctx, bases: &[G1Affine]) -> Option<*mut sys::zkhip_srs> {
    let key = SrsKey { ptr: bases.as_ptr() as usize, len: bases.len(), first: point_words(&bases[0]), last: point_words(&bases[bases.len() - 1]) };
    let mut cache = srs_cache().lock().unwrap();
    if let Some(h) = cache.get(&key) {
        return Some(h.0);
    }
    let mut h: *mut sys::zkhip_srs = std::ptr::null_mut();
    let rc = unsafe { sys::zkhip_srs_load(ctx, bases.as_ptr() as *const u64, bases.len(), &mut h) };
    assert_eq!(rc, sys::ZKHIP_OK, "zkhip_srs_load: {}", sys::last_error());
    cache.insert(key, SrsHandle(h));
    Some(h)
}

/// Drops the device tables of every cached SRS that starts at `bases` (call before freeing a `ParamsKZG`).
pub fn evict_srs(bases: &[G1Affine]) {
    if let Some(ctx) = ctx() {
        let mut cache = srs_cache().lock().unwrap();
        let ptr = bases.as_ptr() as usize;
        let keys: Vec<SrsKey> = cache.keys().filter(|k| k.ptr == ptr).copied().collect();
        for k in keys {
            if let Some(h) = cache.remove(&k) {
                unsafe { sys::zkhip_srs_free(ctx, h.0) };
            }
        }
    }
}

/// `best_multiexp` for bn256 G1 on the GPU; `None` = not applicable (another curve, a small input): run the CPU body.
pub fn try_best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> Option<C::Curve> {
    if TypeId::of::<C>() != TypeId::of::<G1Affine>() || coeffs.len() < MIN_MSM || coeffs.len() != bases.len() {
        return None;
    }
    let ctx = ctx()?;
    // Safety: C == G1Affine (checked above), so the slices are &[Fr] / &[G1Affine] and C::Curve == G1.
    let (scalars, points) = unsafe {
        (std::slice::from_raw_parts(coeffs.as_ptr() as *const Fr, coeffs.len()), std::slice::from_raw_parts(bases.as_ptr() as *const G1Affine, bases.len()))
    };
    let srs = srs_for(ctx, points)?;
    let mut out = [0u64; 12];
    // From 2^20 scalars the library pipelines this call (round 6, csrc/msm.hip): it registers `scalars` with the HIP runtime for the duration of the call
    // (hipHostRegister / hipHostUnregister: the slice must stay alive and unmoved until the call returns, which a `&[Fr]` borrow guarantees), uploads it in
    // chunks and runs each chunk's bucket accumulation as its bytes land — 6.7 ms instead of 8.3 ms at 2^22 on an MI355X; nothing changes on this side.
    let rc = unsafe { sys::zkhip_msm_g1(ctx, srs, scalars.as_ptr() as *const u64, scalars.len(), out.as_mut_ptr()) };
    assert_eq!(rc, sys::ZKHIP_OK, "zkhip_msm_g1: {}", sys::last_error());
    debug_assert_eq!(std::mem::size_of::<G1>(), 96);
    let sum: G1 = unsafe { std::mem::transmute::<[u64; 12], G1>(out) }; // {x, y, z}, z = 1 or the identity (0, 1, 0)
    Some(unsafe { std::mem::transmute_copy::<G1, C::Curve>(&sum) })
}

/// `best_fft` over bn256 Fr scalars on the GPU; `false` = not applicable: run the CPU body.
pub fn try_best_fft<Scalar: Field, G: FftGroup<Scalar>>(a: &mut [G], omega: &Scalar, log_n: u32) -> bool {
    if TypeId::of::<Scalar>() != TypeId::of::<Fr>() || TypeId::of::<G>() != TypeId::of::<Fr>() || log_n < MIN_FFT_LOG_N || a.len() != 1usize << log_n {
        return false;
    }
    let Some(ctx) = ctx() else { return false };
    let rc = unsafe { sys::zkhip_fft(ctx, a.as_mut_ptr() as *mut u64, omega as *const Scalar as *const u64, log_n) };
    assert_eq!(rc, sys::ZKHIP_OK, "zkhip_fft: {}", sys::last_error());
    true
}

/// The process-wide context for the halo2_proofs side of the integration (same device, same SRS cache).
pub fn context() -> Option<*mut sys::zkhip_ctx> {
    ctx()
}
/// The device tables of an SRS slice, loading them on first use.
pub fn srs_handle(bases: &[G1Affine]) -> Option<*mut sys::zkhip_srs> {
    srs_for(ctx()?, bases)
}
