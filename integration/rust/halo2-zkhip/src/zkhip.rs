//! zkhip.rs — `plonk::create_proof` from the point where the witness exists, on libzkhip.so (MI355X).
//! Add to `halo2_proofs/src/` of axiom-crypto/halo2 @ 4b42325 (see ../README.md for the call-site edit in plonk/prover.rs).
//!
//! Field / module names below are upstream's as recalled ([UPSTREAM-RECALL]: `ProvingKey { vk, l0, l_last, l_active_row, fixed_values,
//! fixed_polys, fixed_cosets, permutation, ev }`, `Evaluator { custom_gates, lookups }`, `GraphEvaluator { constants, rotations,
//! calculations, num_intermediates }`, `Calculation`, `ValueSource`): the module lives inside the crate, so private fields are reachable.
use std::collections::HashMap;
use std::ffi::c_void;
use std::sync::{Arc, Mutex, OnceLock};

use ff::{Field, PrimeField, WithSmallOrderMulGroup};
use halo2curves::bn256::{Bn256, Fr, G1Affine};
use halo2curves::zkhip as curves_zkhip;
use rand_core::RngCore;
use zkhip_sys as sys;

use crate::plonk::evaluation::{Calculation, GraphEvaluator, ValueSource};
use crate::plonk::{Any, Error, ProvingKey};
use crate::poly::commitment::{CommitmentScheme, Params, Prover as _};
use crate::poly::kzg::commitment::{KZGCommitmentScheme, ParamsKZG};
use crate::poly::{ExtendedLagrangeCoeff, LagrangeCoeff, Polynomial};
use crate::transcript::{EncodedChallenge, TranscriptWrite};

fn check(rc: i32, what: &str) -> Result<(), Error> {
    if rc == sys::ZKHIP_OK {
        Ok(())
    } else if rc == sys::ZKHIP_ECONSTRAINT {
        Err(Error::ConstraintSystemFailure)
    } else {
        panic!("{what}: {}", sys::last_error())
    }
}

// ---------------------------------------------------------------------------------------------------------------- evaluation graphs
fn vs(v: &ValueSource) -> [i32; 3] {
    match v {
        ValueSource::Constant(i) => [sys::ZK_VS_CONSTANT, *i as i32, 0],
        ValueSource::Intermediate(i) => [sys::ZK_VS_INTERMEDIATE, *i as i32, 0],
        ValueSource::Fixed(c, r) => [sys::ZK_VS_FIXED, *c as i32, *r as i32],
        ValueSource::Advice(c, r) => [sys::ZK_VS_ADVICE, *c as i32, *r as i32],
        ValueSource::Instance(c, r) => [sys::ZK_VS_INSTANCE, *c as i32, *r as i32],
        ValueSource::Challenge(i) => [sys::ZK_VS_CHALLENGE, *i as i32, 0],
        ValueSource::Beta() => [sys::ZK_VS_BETA, 0, 0],
        ValueSource::Gamma() => [sys::ZK_VS_GAMMA, 0, 0],
        ValueSource::Theta() => [sys::ZK_VS_THETA, 0, 0],
        ValueSource::Y() => [sys::ZK_VS_Y, 0, 0],
        ValueSource::PreviousValue() => [sys::ZK_VS_PREVIOUS, 0, 0],
    }
}

/// A `GraphEvaluator` in the int32 code stream of include/zkhip.h: per calculation {op, target, nsrc, nsrc x {kind, a, b}}.
struct FlatGraph {
    constants: Vec<Fr>,
    rotations: Vec<i32>,
    code: Vec<i32>,
    n_calculations: u32,
    n_intermediates: u32,
}
impl FlatGraph {
    fn new(g: &GraphEvaluator<G1Affine>) -> Self {
        let mut code = Vec::new();
        for info in g.calculations.iter() {
            let (op, srcs): (i32, Vec<&ValueSource>) = match &info.calculation {
                Calculation::Add(a, b) => (sys::ZK_OP_ADD, vec![a, b]),
                Calculation::Sub(a, b) => (sys::ZK_OP_SUB, vec![a, b]),
                Calculation::Mul(a, b) => (sys::ZK_OP_MUL, vec![a, b]),
                Calculation::Square(a) => (sys::ZK_OP_SQUARE, vec![a]),
                Calculation::Double(a) => (sys::ZK_OP_DOUBLE, vec![a]),
                Calculation::Negate(a) => (sys::ZK_OP_NEGATE, vec![a]),
                Calculation::Store(a) => (sys::ZK_OP_STORE, vec![a]),
                Calculation::Horner(start, parts, factor) => (sys::ZK_OP_HORNER, std::iter::once(start).chain(std::iter::once(factor)).chain(parts.iter()).collect()),
            };
            code.push(op);
            code.push(info.target as i32);
            code.push(srcs.len() as i32);
            for s in srcs {
                code.extend_from_slice(&vs(s));
            }
        }
        FlatGraph { constants: g.constants.clone(), rotations: g.rotations.clone(), code, n_calculations: g.calculations.len() as u32, n_intermediates: g.num_intermediates as u32 }
    }
    fn ffi(&self) -> sys::zk_graph {
        sys::zk_graph {
            constants: self.constants.as_ptr() as *const u64,
            rotations: self.rotations.as_ptr(),
            code: self.code.as_ptr(),
            n_constants: self.constants.len() as u32,
            n_rotations: self.rotations.len() as u32,
            n_code_words: self.code.len() as u32,
            n_calculations: self.n_calculations,
            n_intermediates: self.n_intermediates,
        }
    }
}

/// theta-compression of a lookup's input / table expressions as a graph: Horner(0, parts, theta) over the expressions
/// (what `lookup::prover::compress_expressions` computes row by row).
fn compress_graph(exprs: &[crate::plonk::Expression<Fr>]) -> FlatGraph {
    let mut g = GraphEvaluator::<G1Affine>::default();
    let parts: Vec<ValueSource> = exprs.iter().map(|e| g.add_expression(e)).collect();
    g.add_calculation(Calculation::Horner(ValueSource::Constant(0), parts, ValueSource::Theta()));
    FlatGraph::new(&g)
}

// ---------------------------------------------------------------------------------------------------------------- device proving key
struct DevCols {
    ctx: *mut sys::zkhip_ctx,
    ptrs: Vec<*const c_void>,
}
impl Drop for DevCols {
    fn drop(&mut self) {
        for p in self.ptrs.drain(..) {
            unsafe { sys::zkhip_free(self.ctx, p as *mut c_void) };
        }
    }
}
impl DevCols {
    fn none(ctx: *mut sys::zkhip_ctx) -> Self {
        DevCols { ctx, ptrs: Vec::new() }
    }
    fn upload<B>(ctx: *mut sys::zkhip_ctx, polys: &[Polynomial<Fr, B>]) -> Self {
        let mut ptrs = Vec::with_capacity(polys.len());
        for p in polys {
            let bytes = p.values.len() * 32;
            let mut d: *mut c_void = std::ptr::null_mut();
            unsafe {
                assert_eq!(sys::zkhip_malloc(ctx, bytes, &mut d), sys::ZKHIP_OK, "zkhip_malloc: {}", sys::last_error());
                assert_eq!(sys::zkhip_memcpy_h2d(ctx, d, p.values.as_ptr() as *const c_void, bytes), sys::ZKHIP_OK, "zkhip_memcpy_h2d: {}", sys::last_error());
            }
            ptrs.push(d as *const c_void);
        }
        DevCols { ctx, ptrs }
    }
}

/// Everything `zk_proving_key` points to.  One per DISTINCT proving key, cached by the key's CONTENT (the verifying key's
/// `transcript_repr`, a hash over the constraint system and the fixed / permutation commitments, plus k) — never by the address of the
/// `ProvingKey`, which the allocator reuses: `let pk = keygen_pk(..)` in a loop would silently meet a stale entry.  Dropping the entry
/// (`evict_device_key`, or process exit) releases what the library caches under `key_id` (`zkhip_key_release`: the key's columns in the
/// coset layout, sorted lookup tables), the uploaded fixed / sigma columns and the domain.
struct DeviceKey {
    ctx: *mut sys::zkhip_ctx,
    ffi: sys::zk_proving_key,
    ext_uploaded: bool,
    _fixed: [DevCols; 3],
    _sigma: [DevCols; 3],
    _l: DevCols,
    _gates: FlatGraph,
    _lookup_graphs: Vec<FlatGraph>,
    _lookup_ffi: Vec<sys::zk_graph>,
    _compress_in: Vec<FlatGraph>,
    _compress_tab: Vec<FlatGraph>,
    _compress_in_ffi: Vec<sys::zk_graph>,
    _compress_tab_ffi: Vec<sys::zk_graph>,
    _single_in: Vec<i32>,
    _single_tab: Vec<i32>,
    _perm_type: Vec<u32>,
    _perm_index: Vec<u32>,
    _adv_q: (Vec<u32>, Vec<i32>),
    _fix_q: (Vec<u32>, Vec<i32>),
}
unsafe impl Send for DeviceKey {}
impl Drop for DeviceKey {
    fn drop(&mut self) {
        unsafe {
            sys::zkhip_key_release(self.ctx, self.ffi.key_id);
            sys::zkhip_domain_free(self.ctx, self.ffi.domain as *mut sys::zkhip_domain);
        }
        // the DevCols members free their columns in their own Drop; the SRS handles belong to halo2curves::zkhip's cache (evict_srs)
    }
}

type KeyFingerprint = ([u8; 32], u32);
fn fingerprint(pk: &ProvingKey<G1Affine>) -> KeyFingerprint {
    // VerifyingKey { transcript_repr, .. }: private field, reachable inside the crate [UPSTREAM-RECALL]
    (pk.vk.transcript_repr.to_repr(), pk.vk.get_domain().k())
}
fn key_cache() -> &'static Mutex<HashMap<KeyFingerprint, Arc<Mutex<DeviceKey>>>> {
    static KEYS: OnceLock<Mutex<HashMap<KeyFingerprint, Arc<Mutex<DeviceKey>>>>> = OnceLock::new();
    KEYS.get_or_init(|| Mutex::new(HashMap::new()))
}
/// Drops the device-side copy of `pk` (call when a proving key is retired: a long-lived service that rotates circuits).  The entry is
/// freed when the last proof using it has returned.
pub fn evict_device_key(pk: &ProvingKey<G1Affine>) -> bool {
    key_cache().lock().unwrap().remove(&fingerprint(pk)).is_some()
}
/// Drops every cached key (before `zkhip_destroy` of the process-wide context).
pub fn evict_all_device_keys() {
    key_cache().lock().unwrap().clear();
}

fn single_column(exprs: &[crate::plonk::Expression<Fr>], want_advice: bool) -> i32 {
    use crate::plonk::Expression;
    if exprs.len() != 1 {
        return -1;
    }
    match &exprs[0] {
        Expression::Advice(q) if want_advice && q.rotation.0 == 0 => q.column_index as i32,
        Expression::Fixed(q) if !want_advice && q.rotation.0 == 0 => q.column_index as i32,
        _ => -1,
    }
}

fn device_key(ctx: *mut sys::zkhip_ctx, params: &ParamsKZG<Bn256>, pk: &ProvingKey<G1Affine>) -> Arc<Mutex<DeviceKey>> {
    static NEXT_ID: std::sync::atomic::AtomicU64 = std::sync::atomic::AtomicU64::new(1);
    let fp = fingerprint(pk);
    let mut keys = key_cache().lock().unwrap();
    if let Some(k) = keys.get(&fp) {
        return k.clone();
    }
    let cs = pk.vk.cs();
    let domain = pk.vk.get_domain();
    let n = params.n() as usize;
    let mut dom: *mut sys::zkhip_domain = std::ptr::null_mut();
    let g_coset = Fr::ZETA; // EvaluationDomain::new's coset generator; passed explicitly so that the fork's choice is honoured
    unsafe {
        assert_eq!(sys::zkhip_domain_new(ctx, cs.degree() as u32, domain.k(), &g_coset as *const Fr as *const u64, &mut dom), sys::ZKHIP_OK, "{}", sys::last_error());
    }
    // cs.degree() - 1 cosets of the size-n domain are fewer rows than the extended domain (halo2-lib's degree 4: 3 < 4): the library
    // evaluates the quotient there and derives the key's columns in that layout from the coefficient forms (include/zkhip.h, "the same
    // quotient on quotient_poly_degree cosets"), so the pk's extended cosets — most of its bytes — are not uploaded at all
    // Whether that path applies is the LIBRARY's decision (option "coset_quotient" / ZKHIP_COSET_QUOTIENT, degree, key_id): the key is
    // first assembled without the extended forms, the library is asked (zkhip_coset_quotient_applies), and the extended forms are
    // uploaded only if it answers no (`ensure_extended` below) — the shim never guesses.
    let fixed = [DevCols::upload(ctx, &pk.fixed_values), DevCols::upload(ctx, &pk.fixed_polys), DevCols::none(ctx)];
    let sigma = [DevCols::upload(ctx, &pk.permutation.permutations), DevCols::upload(ctx, &pk.permutation.polys), DevCols::none(ctx)];
    let l = DevCols::none(ctx);
    let gates = FlatGraph::new(&pk.ev.custom_gates);
    let lookup_graphs: Vec<FlatGraph> = pk.ev.lookups.iter().map(FlatGraph::new).collect();
    let compress_in: Vec<FlatGraph> = cs.lookups().iter().map(|a| compress_graph(a.input_expressions())).collect();
    let compress_tab: Vec<FlatGraph> = cs.lookups().iter().map(|a| compress_graph(a.table_expressions())).collect();
    let single_in: Vec<i32> = cs.lookups().iter().map(|a| single_column(a.input_expressions(), true)).collect();
    let single_tab: Vec<i32> = cs.lookups().iter().map(|a| single_column(a.table_expressions(), false)).collect();
    let lookup_ffi: Vec<sys::zk_graph> = lookup_graphs.iter().map(|g| g.ffi()).collect();
    let compress_in_ffi: Vec<sys::zk_graph> = compress_in.iter().map(|g| g.ffi()).collect();
    let compress_tab_ffi: Vec<sys::zk_graph> = compress_tab.iter().map(|g| g.ffi()).collect();
    let perm_cols = cs.permutation().get_columns();
    let perm_type: Vec<u32> = perm_cols.iter().map(|c| match c.column_type() { Any::Advice(_) => 0, Any::Fixed => 1, Any::Instance => 2 }).collect();
    let perm_index: Vec<u32> = perm_cols.iter().map(|c| c.index() as u32).collect();
    let adv_q = (cs.advice_queries().iter().map(|(c, _)| c.index() as u32).collect::<Vec<_>>(), cs.advice_queries().iter().map(|(_, r)| r.0).collect::<Vec<_>>());
    let fix_q = (cs.fixed_queries().iter().map(|(c, _)| c.index() as u32).collect::<Vec<_>>(), cs.fixed_queries().iter().map(|(_, r)| r.0).collect::<Vec<_>>());
    // ParamsKZG { g, g_lagrange, .. }: pub(crate) fields of poly/kzg/commitment.rs
    let g = curves_zkhip::srs_handle(&params.g[..n]).expect("zkhip: SRS g");
    let g_lagrange = curves_zkhip::srs_handle(&params.g_lagrange[..n]).expect("zkhip: SRS g_lagrange");
    let delta = Fr::DELTA;
    let mut delta_w = [0u64; 4];
    unsafe { std::ptr::copy_nonoverlapping(&delta as *const Fr as *const u64, delta_w.as_mut_ptr(), 4) };
    let ffi = sys::zk_proving_key {
        k: domain.k(),
        cs_degree: cs.degree() as u32,
        blinding_factors: cs.blinding_factors() as u32,
        n_fixed: pk.fixed_values.len() as u32,
        n_advice: cs.num_advice_columns() as u32,
        n_instance: cs.num_instance_columns() as u32,
        n_lookups: cs.lookups().len() as u32,
        n_perm_columns: perm_cols.len() as u32,
        g,
        g_lagrange,
        domain: dom,
        fixed_lagrange: fixed[0].ptrs.as_ptr(),
        fixed_coeff: fixed[1].ptrs.as_ptr(),
        fixed_cosets: std::ptr::null(),
        sigma_lagrange: sigma[0].ptrs.as_ptr(),
        sigma_coeff: sigma[1].ptrs.as_ptr(),
        sigma_cosets: std::ptr::null(),
        l0: std::ptr::null(),
        l_last: std::ptr::null(),
        l_active_row: std::ptr::null(),
        custom_gates: gates.ffi(),
        lookup_graphs: lookup_ffi.as_ptr(),
        lookup_input_compress: compress_in_ffi.as_ptr(),
        lookup_table_compress: compress_tab_ffi.as_ptr(),
        lookup_input_advice_column: single_in.as_ptr(),
        lookup_table_fixed_column: single_tab.as_ptr(),
        key_id: NEXT_ID.fetch_add(1, std::sync::atomic::Ordering::Relaxed),
        perm_column_type: perm_type.as_ptr(),
        perm_column_index: perm_index.as_ptr(),
        n_advice_queries: adv_q.0.len() as u32,
        n_fixed_queries: fix_q.0.len() as u32,
        advice_query_column: adv_q.0.as_ptr(),
        advice_query_rotation: adv_q.1.as_ptr(),
        fixed_query_column: fix_q.0.as_ptr(),
        fixed_query_rotation: fix_q.1.as_ptr(),
        delta: delta_w,
        vk_transcript_repr: std::ptr::null(), // absorbed by upstream's code before the hand-over
        // phases: this shim hands over single-phase circuits only (the reference's three: /root/reference/src/helpers.rs:97-172,
        // src/sha256_bit_circuit.rs:51-56); a multi-phase circuit fills these from cs.advice_column_phase() / cs.challenge_phase() and
        // passes a zk_proof_inputs.advice_phase callback that re-runs its synthesize with the challenges
        advice_column_phase: std::ptr::null(),
        n_challenges: 0,
        challenge_phase: std::ptr::null(),
    };
    let key = Arc::new(Mutex::new(DeviceKey {
        ctx,
        ffi,
        ext_uploaded: false,
        _fixed: fixed,
        _sigma: sigma,
        _l: l,
        _gates: gates,
        _lookup_graphs: lookup_graphs,
        _lookup_ffi: lookup_ffi,
        _compress_in: compress_in,
        _compress_tab: compress_tab,
        _compress_in_ffi: compress_in_ffi,
        _compress_tab_ffi: compress_tab_ffi,
        _single_in: single_in,
        _single_tab: single_tab,
        _perm_type: perm_type,
        _perm_index: perm_index,
        _adv_q: adv_q,
        _fix_q: fix_q,
    }));
    keys.insert(fp, key.clone());
    key
}

/// The extended-domain forms of the key (fixed / sigma cosets, l_0, l_last, l_active_row), uploaded the first time the library says
/// it will work on the extended domain for this key (coset quotient switched off for byte parity on unsatisfied witnesses, or a
/// degree for which the cosets save nothing).
fn ensure_extended(key: &mut DeviceKey, pk: &ProvingKey<G1Affine>) {
    if key.ext_uploaded || unsafe { sys::zkhip_coset_quotient_applies(key.ctx, &key.ffi) } != 0 {
        return;
    }
    let ctx = key.ctx;
    key._fixed[2] = DevCols::upload(ctx, &pk.fixed_cosets);
    key._sigma[2] = DevCols::upload(ctx, &pk.permutation.cosets);
    key._l = DevCols::upload(ctx, &[pk.l0.clone(), pk.l_last.clone(), pk.l_active_row.clone()]);
    key.ffi.fixed_cosets = key._fixed[2].ptrs.as_ptr();
    key.ffi.sigma_cosets = key._sigma[2].ptrs.as_ptr();
    key.ffi.l0 = key._l.ptrs[0];
    key.ffi.l_last = key._l.ptrs[1];
    key.ffi.l_active_row = key._l.ptrs[2];
    key.ext_uploaded = true;
}

// ---------------------------------------------------------------------------------------------------------------- transcript callbacks
unsafe extern "C" fn cb_write_point<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>(user: *mut c_void, _bytes32: *const u8, xy: *const u64) {
    let t = &mut *(user as *mut T);
    let p: G1Affine = std::ptr::read(xy as *const G1Affine); // {x, y} in Montgomery form: the layout of G1Affine
    t.write_point(p).expect("transcript.write_point");
}
unsafe extern "C" fn cb_write_scalar<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>(user: *mut c_void, scalar: *const u64) {
    let t = &mut *(user as *mut T);
    t.write_scalar(std::ptr::read(scalar as *const Fr)).expect("transcript.write_scalar");
}
unsafe extern "C" fn cb_squeeze<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>>(user: *mut c_void, out: *mut u64) {
    let t = &mut *(user as *mut T);
    let c: Fr = t.squeeze_challenge().get_scalar();
    std::ptr::copy_nonoverlapping(&c as *const Fr as *const u64, out, 4);
}

// ---------------------------------------------------------------------------------------------------------------- entry points
/// bn256 + KZG with one circuit instance and a single phase: what every command of the reference proves.  The multi-open scheme cannot
/// be told from a type id (`ProverSHPLONK<'params, _>` is not `'static`): the call site in plonk/prover.rs passes `P::SHPLONK`-ness
/// itself (`is_shplonk`: add `const IS_SHPLONK: bool` to the `Prover` trait, true for `ProverSHPLONK`, or gate the call with a cargo
/// feature when only SHPLONK is ever used, as in the reference).
pub fn applicable<Scheme: CommitmentScheme + 'static>(pk: &ProvingKey<Scheme::Curve>, n_instances: usize, is_shplonk: bool) -> bool {
    use std::any::TypeId;
    is_shplonk
        && TypeId::of::<Scheme>() == TypeId::of::<KZGCommitmentScheme<Bn256>>()
        && n_instances == 1
        && pk.vk.cs().phases().count() == 1
        && curves_zkhip::context().is_some()
}

/// `create_proof` from the point where `advice` (Lagrange form, blinding rows already drawn from `rng`) exists.  Draws the remaining
/// randomness from `rng` in upstream's order — per lookup the permuted input's and the permuted table's blinding rows
/// (`permute_expression_pair`), per permutation set its z blinding rows (`permutation::commit`), per lookup its z blinding rows
/// (`commit_product`), then the vanishing argument's random polynomial (`vanishing::commit`) — and hands everything to the library.
pub fn create_proof_after_synthesis<E: EncodedChallenge<G1Affine>, R: RngCore, T: TranscriptWrite<G1Affine, E>>(
    params: &ParamsKZG<Bn256>,
    pk: &ProvingKey<G1Affine>,
    instances: &[&[Fr]],
    advice: &[Polynomial<Fr, LagrangeCoeff>],
    mut rng: R,
    transcript: &mut T,
) -> Result<(), Error> {
    let ctx = curves_zkhip::context().expect("zkhip context");
    let key_entry = device_key(ctx, params, pk);
    let mut key = key_entry.lock().unwrap();
    ensure_extended(&mut key, pk);
    let cs = pk.vk.cs();
    let n = params.n() as usize;
    let bf = cs.blinding_factors();
    let n_lookups = cs.lookups().len();
    let chunk = cs.degree() - 2;
    let n_sets = (cs.permutation().get_columns().len() + chunk - 1) / chunk.max(1);
    let draw = |rng: &mut R, count: usize| -> Vec<Fr> { (0..count).map(|_| Fr::random(&mut *rng)).collect() };
    let lookup_permuted = draw(&mut rng, n_lookups * 2 * (bf + 1));
    let perm_z = draw(&mut rng, n_sets * bf);
    let lookup_z = draw(&mut rng, n_lookups * bf);
    let random_poly = draw(&mut rng, n);
    let blinding = sys::zk_blinding {
        lookup_permuted: lookup_permuted.as_ptr() as *const c_void,
        perm_z: perm_z.as_ptr() as *const c_void,
        lookup_z: lookup_z.as_ptr() as *const c_void,
        random_poly: random_poly.as_ptr() as *const c_void,
        on_host: 1,
    };
    let advice_ptrs: Vec<*const c_void> = advice.iter().map(|p| p.values.as_ptr() as *const c_void).collect();
    let inst_ptrs: Vec<*const u64> = instances.iter().map(|c| c.as_ptr() as *const u64).collect();
    let inst_len: Vec<u32> = instances.iter().map(|c| c.len() as u32).collect();
    let inputs = sys::zk_proof_inputs {
        advice: advice_ptrs.as_ptr(),
        advice_on_host: 1,
        d_instance: std::ptr::null(),
        instance_values: inst_ptrs.as_ptr(),
        instance_len: inst_len.as_ptr(),
        blinding: &blinding,
        blinding_seed: 0,
        advice_phase: None,
        advice_phase_user: std::ptr::null_mut(),
    };
    let callbacks = sys::zk_transcript {
        user: transcript as *mut T as *mut c_void,
        write_point: Some(cb_write_point::<E, T>),
        squeeze_challenge: Some(cb_squeeze::<E, T>),
        write_scalar: Some(cb_write_scalar::<E, T>),
        common_scalar: None, // vk.hash_into and the instance values were absorbed by upstream's code before this point
    };
    let mut out = sys::zk_proof_out {
        d_h: std::ptr::null(),
        evals: std::ptr::null_mut(),
        eval_poly: std::ptr::null_mut(),
        eval_rotation: std::ptr::null_mut(),
        eval_write_order: std::ptr::null_mut(),
        evals_cap: 0,
        n_evals: 0,
    };
    check(unsafe { sys::zkhip_create_proof_ex(ctx, &key.ffi, &inputs, &callbacks, &mut out) }, "zkhip_create_proof_ex")
}
