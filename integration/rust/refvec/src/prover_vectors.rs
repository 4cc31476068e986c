//! Whole-`create_proof` vectors from the UNPATCHED pinned crates (halo2_proofs 4b42325, halo2curves e185711, snark-verifier-sdk 7011e8c):
//! what the three rows with the most recall in them need in order to be pinned — the proving SCHEDULE (commitment / evaluation write
//! order, lookup permutation, grand products, quotient, SHPLONK: SURVEY.md §8 a8), through it the quotient `evaluate_h` computes (a7: the
//! method is `pub(in crate::plonk)`, so it cannot be called from outside the crate; its output is pinned through the quotient-piece
//! commitments inside the proof and through upstream's own `verify_proof`, which this program runs on every proof it prints), and the
//! on-disk formats (f3: `ParamsKZG::write`, `ProvingKey::write(RawBytesUnchecked)`, bincode `Snark`).
//!
//! The circuit is the repository's `CircuitShape.small(6)` (halo2-lib shaped: two vertical-gate advice columns q·(a + b·c − d), one
//! lookup-advice column against a fixed table column, one constants column, one instance column) and `two_phase(6)` (one more advice
//! column in the second phase, a3 = challenge_0 · a0, one user challenge).  Selectors are plain FIXED columns, created first, so that the
//! fixed-column order is the shape's (q_0, q_1, constants, table) and not "user columns, then compressed selectors".  Columns, gates,
//! lookup and equality are declared in the order that makes upstream's query lists equal `CircuitShape.queries()`.
//! Everything the circuit contains — fixed columns, copy constraints, witness, instance — is PRINTED, so the consumer
//! (tests/test_reference_vectors.py) rebuilds the instance from the file and regenerates nothing.
//!
//! Randomness: `create_proof` takes its blinding from `rng`.  `CountingRng` is the SplitMix64 sequence of oracle/pyref.py
//! (state += 0x9E37…, output = mix(state)) and counts the u64 it hands out; an `Fr::random` takes 8 of them (512 bits, little-endian,
//! reduced mod r).  The consumer replays the same stream and maps draws to roles by upstream's draw order [UPSTREAM-RECALL]; the printed
//! count decides between the two candidate orders it knows (with / without the `Blind` draws KZG ignores).
//! API spellings marked [UPSTREAM-RECALL] are from memory of the axiom fork (no Rust toolchain where this was written): a compile error
//! on such a line is a spelling to fix, not a design question.
use std::io;

use ff::{Field, PrimeField};
use halo2_proofs::circuit::{Layouter, SimpleFloorPlanner, Value};
use halo2_proofs::plonk::{
    create_proof, keygen_pk, keygen_vk, verify_proof, Advice, Challenge, Circuit, Column, ConstraintSystem, Error, Fixed, FirstPhase, Instance,
    ProvingKey, SecondPhase,
};
use halo2_proofs::poly::commitment::{Params, ParamsProver};
use halo2_proofs::poly::kzg::commitment::{KZGCommitmentScheme, ParamsKZG};
use halo2_proofs::poly::kzg::multiopen::{ProverSHPLONK, VerifierSHPLONK};
use halo2_proofs::poly::kzg::strategy::SingleStrategy;
use halo2_proofs::poly::Rotation;
use halo2_proofs::transcript::{
    Blake2bRead, Blake2bWrite, Challenge255, EncodedChallenge, Transcript, TranscriptReadBuffer, TranscriptWrite, TranscriptWriterBuffer,
};
use halo2_proofs::SerdeFormat;
use halo2curves::bn256::{Bn256, Fr, G1Affine};
use rand_core::{Error as RngError, RngCore};
use serde_json::{json, Value as Json};

pub const K: u32 = 6;
pub const SRS_TRAPDOOR: u64 = 0x1D5C0FFEE;
pub const RNG_SEED: u64 = 0x5EED_0006;

fn hex_fr(x: &Fr) -> String {
    let mut b = x.to_repr().as_ref().to_vec();
    b.reverse();
    hex::encode(b)
}

// ------------------------------------------------------------------------------------------------------------------------------ rng
fn splitmix_next(state: &mut u64) -> u64 {
    *state = state.wrapping_add(0x9E3779B97F4A7C15);
    let mut z = *state;
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}
/// SplitMix64 as an `RngCore`, counting the u64 handed out.  `fill_bytes` and `next_u64` draw from ONE little-endian u64 stream, so the
/// count is the same whichever of the two `Fr::random` uses.
pub struct CountingRng {
    state: u64,
    pub u64_drawn: u64,
}
impl CountingRng {
    pub fn new(seed: u64) -> Self {
        CountingRng { state: seed, u64_drawn: 0 }
    }
}
impl RngCore for CountingRng {
    fn next_u32(&mut self) -> u32 {
        self.next_u64() as u32
    }
    fn next_u64(&mut self) -> u64 {
        self.u64_drawn += 1;
        splitmix_next(&mut self.state)
    }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        for chunk in dest.chunks_mut(8) {
            let w = self.next_u64().to_le_bytes();
            chunk.copy_from_slice(&w[..chunk.len()]);
        }
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), RngError> {
        self.fill_bytes(dest);
        Ok(())
    }
}
/// `ParamsKZG::setup(k, rng)` draws its trapdoor with one `Fr::random(rng)`: this rng makes that draw equal `s` (low limb s, the other
/// seven words zero: the 512-bit value reduces to s), so the SRS is [s^i] G for the s the test suite uses everywhere.
struct TrapdoorRng {
    first: Option<u64>,
}
impl RngCore for TrapdoorRng {
    fn next_u32(&mut self) -> u32 {
        self.next_u64() as u32
    }
    fn next_u64(&mut self) -> u64 {
        self.first.take().unwrap_or(0)
    }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        for chunk in dest.chunks_mut(8) {
            let w = self.next_u64().to_le_bytes();
            chunk.copy_from_slice(&w[..chunk.len()]);
        }
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), RngError> {
        self.fill_bytes(dest);
        Ok(())
    }
}

// ------------------------------------------------------------------------------------------------------------------------------ transcript recorder
/// Delegates to the wrapped transcript and keeps every squeezed challenge (as a scalar), in squeeze order.
pub struct Rec<T> {
    pub inner: T,
    pub challenges: Vec<Fr>,
}
impl<E: EncodedChallenge<G1Affine>, T: Transcript<G1Affine, E>> Transcript<G1Affine, E> for Rec<T> {
    fn squeeze_challenge(&mut self) -> E {
        let c = self.inner.squeeze_challenge();
        self.challenges.push(c.get_scalar());
        c
    }
    fn common_point(&mut self, point: G1Affine) -> io::Result<()> {
        self.inner.common_point(point)
    }
    fn common_scalar(&mut self, scalar: Fr) -> io::Result<()> {
        self.inner.common_scalar(scalar)
    }
}
impl<E: EncodedChallenge<G1Affine>, T: TranscriptWrite<G1Affine, E>> TranscriptWrite<G1Affine, E> for Rec<T> {
    fn write_point(&mut self, point: G1Affine) -> io::Result<()> {
        self.inner.write_point(point)
    }
    fn write_scalar(&mut self, scalar: Fr) -> io::Result<()> {
        self.inner.write_scalar(scalar)
    }
}

// ------------------------------------------------------------------------------------------------------------------------------ the circuit
#[derive(Clone)]
pub struct ShapeConfig {
    q: [Column<Fixed>; 2],        // fixed 0, 1: the vertical gates' selectors (plain fixed columns)
    constants: Column<Fixed>,     // fixed 2
    table: Column<Fixed>,         // fixed 3
    basic: [Column<Advice>; 2],   // advice 0, 1
    lookup: Column<Advice>,       // advice 2
    phase1: Option<Column<Advice>>, // advice 3 (two_phase only)
    challenge: Option<Challenge>,
    instance: Column<Instance>,
}

/// One satisfiable instance, fully explicit.  Rows are the USABLE rows (0 .. n − (blinding_factors + 1)); `create_proof` fills the rest
/// of every advice column from its rng.
#[derive(Clone, Default)]
pub struct ShapeCircuit {
    pub two_phase: bool,
    pub usable: usize,
    pub fixed: [Vec<Fr>; 4],          // q_0, q_1, constants, table (usable rows; upstream leaves the unusable rows zero)
    pub advice: [Vec<Fr>; 3],         // a_0, a_1 (gates), a_2 (lookup input)
    pub instance: Vec<Fr>,
    /// copy constraints as (permutation column, row, permutation column, row); permutation columns in the order they are enabled below:
    /// advice 0, 1, 2 (, 3), fixed 2 (constants), instance 0
    pub copies: Vec<(usize, usize, usize, usize)>,
}

impl ShapeCircuit {
    /// blinding_factors = 6 for this constraint system (four rotations of one advice column, + 2): 57 usable rows at k = 6
    pub fn build(two_phase: bool) -> Self {
        let n = 1usize << K;
        let usable = n - 7;
        let mut st = 0xC1C0_17u64 + two_phase as u64;
        let mut small = |bound: u64| Fr::from(splitmix_next(&mut st) % bound);
        let gates: Vec<usize> = (0..usable - 3).step_by(4).collect();
        let mut fixed = [vec![Fr::ZERO; usable], vec![Fr::ZERO; usable], vec![Fr::ZERO; usable], vec![Fr::ZERO; usable]];
        for &r in &gates {
            fixed[0][r] = Fr::ONE;
            fixed[1][r] = Fr::ONE;
        }
        for i in 0..8 {
            fixed[2][i] = Fr::from(100 + i as u64); // the constants column
        }
        for i in 0..usable {
            fixed[3][i] = Fr::from(i as u64); // the lookup table: 0 .. usable − 1 (0 again on the unusable rows)
        }
        let mut advice = [vec![Fr::ZERO; usable], vec![Fr::ZERO; usable], vec![Fr::ZERO; usable]];
        for i in 0..usable {
            advice[0][i] = small(1 << 40);
            advice[1][i] = small(1 << 40);
            advice[2][i] = small(usable as u64); // in the table
        }
        let instance: Vec<Fr> = (0..4).map(|j| Fr::from(1000 + j as u64)).collect();
        let perm_instance = if two_phase { 5 } else { 4 };
        let perm_constants = perm_instance - 1;
        let mut copies = Vec::new();
        for g in 0..4 {
            // b input of column-0 gate g  ==  lookup cell g
            advice[0][gates[g] + 1] = advice[2][g];
            copies.push((0, gates[g] + 1, 2, g));
            // b input of column-1 gate g  ==  constant g
            advice[1][gates[g] + 1] = fixed[2][g];
            copies.push((1, gates[g] + 1, perm_constants, g));
        }
        for (j, g) in (4..8).enumerate() {
            // c input of column-0 gate g  ==  public input j
            advice[0][gates[g] + 2] = instance[j];
            copies.push((0, gates[g] + 2, perm_instance, j));
        }
        // gate outputs d = a + b c, column by column; then the copies that take a column-0 OUTPUT into column 1, then column 1's outputs
        for &r in &gates {
            advice[0][r + 3] = advice[0][r] + advice[0][r + 1] * advice[0][r + 2];
        }
        for g in 8..11 {
            advice[1][gates[g] + 2] = advice[0][gates[g] + 3];
            copies.push((1, gates[g] + 2, 0, gates[g] + 3));
        }
        for &r in &gates {
            advice[1][r + 3] = advice[1][r] + advice[1][r + 1] * advice[1][r + 2];
        }
        ShapeCircuit { two_phase, usable, fixed, advice, instance, copies }
    }

    pub fn describe(&self) -> Json {
        let col = |v: &Vec<Fr>| v.iter().map(hex_fr).collect::<Vec<_>>();
        json!({
            "k": K, "two_phase": self.two_phase, "usable_rows": self.usable,
            "fixed": self.fixed.iter().map(col).collect::<Vec<_>>(),
            "advice": self.advice.iter().map(col).collect::<Vec<_>>(),
            "instance": [col(&self.instance)],
            "copies": self.copies.iter().map(|c| vec![c.0, c.1, c.2, c.3]).collect::<Vec<_>>(),
        })
    }
}

impl Circuit<Fr> for ShapeCircuit {
    type Config = ShapeConfig;
    type FloorPlanner = SimpleFloorPlanner;
    type Params = bool; // two_phase [UPSTREAM-RECALL: the fork is built with `circuit-params` — the reference's circuits declare `type Params`, src/sha256_bit_circuit.rs:47]

    fn without_witnesses(&self) -> Self {
        self.clone()
    }
    fn params(&self) -> bool {
        self.two_phase
    }
    fn configure_with_params(meta: &mut ConstraintSystem<Fr>, two_phase: bool) -> ShapeConfig {
        // column creation order = the shape's column indices
        let q = [meta.fixed_column(), meta.fixed_column()];
        let constants = meta.fixed_column();
        let table = meta.fixed_column();
        let basic = [meta.advice_column(), meta.advice_column()];
        let lookup = meta.advice_column();
        let (phase1, challenge) = if two_phase {
            (Some(meta.advice_column_in(SecondPhase)), Some(meta.challenge_usable_after(FirstPhase)))
        } else {
            (None, None)
        };
        let instance = meta.instance_column();
        // gates first (queries in the shape's order: q, a(0), a(1), a(2), a(3)), column by column
        for c in 0..2 {
            meta.create_gate("vertical", |m| {
                let s = m.query_fixed(q[c], Rotation::cur());
                let a = m.query_advice(basic[c], Rotation::cur());
                let b = m.query_advice(basic[c], Rotation(1));
                let cc = m.query_advice(basic[c], Rotation(2));
                let d = m.query_advice(basic[c], Rotation(3));
                vec![s * (a + b * cc - d)]
            });
        }
        if let (Some(a3), Some(ch)) = (phase1, challenge) {
            meta.create_gate("phase 1", |m| {
                let s = m.query_fixed(q[0], Rotation::cur());
                let x = m.query_advice(a3, Rotation::cur());
                let a0 = m.query_advice(basic[0], Rotation::cur());
                let c = m.query_challenge(ch);
                vec![s * (x - a0 * c)]
            });
        }
        // the range lookup: lookup-advice in the table column
        meta.lookup_any("range", |m| {
            let a = m.query_advice(lookup, Rotation::cur());
            let t = m.query_fixed(table, Rotation::cur());
            vec![(a, t)]
        });
        // equality last: permutation columns advice 0, 1, 2 (, 3), constants, instance
        meta.enable_equality(basic[0]);
        meta.enable_equality(basic[1]);
        meta.enable_equality(lookup);
        if let Some(a3) = phase1 {
            meta.enable_equality(a3);
        }
        meta.enable_equality(constants);
        meta.enable_equality(instance);
        ShapeConfig { q, constants, table, basic, lookup, phase1, challenge, instance }
    }
    fn configure(_meta: &mut ConstraintSystem<Fr>) -> ShapeConfig {
        unreachable!("configure_with_params is the entry point under circuit-params")
    }

    fn synthesize(&self, cfg: ShapeConfig, mut layouter: impl Layouter<Fr>) -> Result<(), Error> {
        let ch = cfg.challenge.map(|c| layouter.get_challenge(c));
        let inst_cells = layouter.assign_region(
            || "shape",
            |mut region| {
                // [UPSTREAM-RECALL] the axiom fork's Region: assign_advice(column, offset, Value<F>) -> AssignedCell, assign_fixed(column,
                // offset, F) -> Cell, constrain_equal(Cell, Cell) — no annotation closures, no Result (upstream PSE: closures and Results)
                let fixed_cols = [cfg.q[0], cfg.q[1], cfg.constants, cfg.table];
                let mut fixed_cells = vec![Vec::new(); 4];
                for (j, col) in fixed_cols.iter().enumerate() {
                    for (i, v) in self.fixed[j].iter().enumerate() {
                        fixed_cells[j].push(region.assign_fixed(*col, i, *v));
                    }
                }
                let adv_cols = [cfg.basic[0], cfg.basic[1], cfg.lookup];
                let mut adv_cells = vec![Vec::new(); 4];
                for (j, col) in adv_cols.iter().enumerate() {
                    for (i, v) in self.advice[j].iter().enumerate() {
                        adv_cells[j].push(region.assign_advice(*col, i, Value::known(*v)).cell());
                    }
                }
                if let (Some(a3), Some(chv)) = (cfg.phase1, ch) {
                    for (i, v) in self.advice[0].iter().enumerate() {
                        adv_cells[3].push(region.assign_advice(a3, i, chv.map(|c| *v * c)).cell());
                    }
                }
                let n_adv = if self.two_phase { 4 } else { 3 };
                let mut to_instance = Vec::new();
                for &(ca, ra, cb, rb) in &self.copies {
                    let cell_of = |c: usize, r: usize| if c < n_adv { Some(adv_cells[c][r]) } else if c == n_adv { Some(fixed_cells[2][r]) } else { None };
                    match (cell_of(ca, ra), cell_of(cb, rb)) {
                        (Some(x), Some(y)) => region.constrain_equal(x, y), // [UPSTREAM-RECALL] Cell by value, unit result (halo2-lib's halo2-axiom branch calls it so)
                        (Some(x), None) => to_instance.push((x, rb)),
                        _ => unreachable!("an instance cell on the left of a copy"),
                    }
                }
                Ok(to_instance)
            },
        )?;
        for (cell, row) in inst_cells {
            layouter.constrain_instance(cell, cfg.instance, row); // [UPSTREAM-RECALL] unit in the fork (src/sha256_bit_circuit.rs:94 of the reference calls it so)
        }
        Ok(())
    }
}

impl snark_verifier_sdk::CircuitExt<Fr> for ShapeCircuit {
    fn num_instance(&self) -> Vec<usize> {
        vec![self.instance.len()]
    }
    fn instances(&self) -> Vec<Vec<Fr>> {
        vec![self.instance.clone()]
    }
}

// ------------------------------------------------------------------------------------------------------------------------------ proofs
fn describe_cs(pk: &ProvingKey<G1Affine>) -> Json {
    let cs = pk.get_vk().cs();
    json!({
        "degree": cs.degree(), "blinding_factors": cs.blinding_factors(),
        "num_fixed_columns": cs.num_fixed_columns(), "num_advice_columns": cs.num_advice_columns(), "num_instance_columns": cs.num_instance_columns(),
        "num_selectors": cs.num_selectors(),
        "advice_queries": cs.advice_queries().iter().map(|(c, r)| json!([c.index(), r.0])).collect::<Vec<_>>(),
        "fixed_queries": cs.fixed_queries().iter().map(|(c, r)| json!([c.index(), r.0])).collect::<Vec<_>>(),
        "instance_queries": cs.instance_queries().iter().map(|(c, r)| json!([c.index(), r.0])).collect::<Vec<_>>(),
        "permutation_columns": cs.permutation().get_columns().iter().map(|c| json!([format!("{:?}", c.column_type()), c.index()])).collect::<Vec<_>>(),
        "permutation_columns_note": "column types print as upstream's Debug of Any (Advice carries its phase)",
    })
}

/// One `create_proof` under transcript `T`, rng = CountingRng(RNG_SEED); upstream's own verifier run on the result.
fn prove_with<E, TW, TR>(params: &ParamsKZG<Bn256>, pk: &ProvingKey<G1Affine>, circuit: &ShapeCircuit, writer: TW, reader: impl FnOnce(Vec<u8>) -> TR, finalize: impl FnOnce(TW) -> Vec<u8>) -> Json
where
    E: EncodedChallenge<G1Affine>,
    TW: TranscriptWrite<G1Affine, E>,
    TR: halo2_proofs::transcript::TranscriptRead<G1Affine, E>,
{
    let mut rng = CountingRng::new(RNG_SEED);
    let mut rec = Rec { inner: writer, challenges: Vec::new() };
    let inst: Vec<&[Fr]> = vec![circuit.instance.as_slice()]; // (&[&inst] below: the shape snark-verifier-sdk's gen_proof passes)
    create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, E, _, _, _>(params, pk, &[circuit.clone()], &[&inst], &mut rng, &mut rec).expect("create_proof");
    let challenges = rec.challenges.clone();
    let proof = finalize(rec.inner);
    let mut tr = reader(proof.clone());
    let strategy = SingleStrategy::new(params);
    let verified = verify_proof::<KZGCommitmentScheme<Bn256>, VerifierSHPLONK<'_, Bn256>, E, _, _>(params.verifier_params(), pk.get_vk(), strategy, &[&inst], &mut tr).is_ok();
    json!({ "proof": hex::encode(&proof), "challenges": challenges.iter().map(hex_fr).collect::<Vec<_>>(), "rng_u64_drawn": rng.u64_drawn, "verified": verified })
}

fn one_circuit(params: &ParamsKZG<Bn256>, two_phase: bool) -> (Json, ProvingKey<G1Affine>, ShapeCircuit) {
    use snark_verifier_sdk::halo2::{PoseidonTranscript, POSEIDON_SPEC};
    use snark_verifier_sdk::snark_verifier::loader::native::NativeLoader;
    use snark_verifier_sdk::snark_verifier::system::halo2::transcript::evm::EvmTranscript;

    let circuit = ShapeCircuit::build(two_phase);
    let vk = keygen_vk(params, &circuit).expect("keygen_vk");
    let pk = keygen_pk(params, vk, &circuit).expect("keygen_pk");
    let blake = prove_with::<Challenge255<G1Affine>, _, _>(
        params, &pk, &circuit,
        Blake2bWrite::<Vec<u8>, G1Affine, Challenge255<G1Affine>>::init(vec![]),
        |p| Blake2bRead::<_, G1Affine, Challenge255<G1Affine>>::init(io::Cursor::new(p)),
        |w| w.finalize(),
    );
    let poseidon = prove_with(
        params, &pk, &circuit,
        PoseidonTranscript::<NativeLoader, Vec<u8>>::from_spec(vec![], POSEIDON_SPEC.clone()),
        |p| PoseidonTranscript::<NativeLoader, io::Cursor<Vec<u8>>>::from_spec(io::Cursor::new(p), POSEIDON_SPEC.clone()),
        |w| w.finalize(),
    );
    let evm = prove_with(
        params, &pk, &circuit,
        EvmTranscript::<G1Affine, NativeLoader, Vec<u8>, Vec<u8>>::new(vec![]),
        |p| EvmTranscript::<G1Affine, NativeLoader, io::Cursor<Vec<u8>>, Vec<u8>>::new(io::Cursor::new(p)),
        |w| w.finalize(),
    );
    let doc = json!({
        "circuit": circuit.describe(),
        "constraint_system": describe_cs(&pk),
        // VerifyingKey::transcript_repr: what vk.hash_into absorbs first (a Blake2b hash of the pinned vk's Debug form: not reproducible
        // without Rust, so the consumer TAKES it from here as the key's transcript representation)
        "vk_transcript_repr": hex_fr(&pk.get_vk().transcript_repr()),
        "fixed_commitments": pk.get_vk().fixed_commitments().iter().map(|p| hex::encode(group::GroupEncoding::to_bytes(p).as_ref())).collect::<Vec<_>>(),
        "permutation_commitments": pk.get_vk().permutation().commitments().iter().map(|p| hex::encode(group::GroupEncoding::to_bytes(p).as_ref())).collect::<Vec<_>>(),
        "proofs": { "blake2b": blake, "poseidon": poseidon, "evm": evm },
    });
    (doc, pk, circuit)
}

pub fn emit() -> Json {
    let params = ParamsKZG::<Bn256>::setup(K, TrapdoorRng { first: Some(SRS_TRAPDOOR) });
    let (small, pk, circuit) = one_circuit(&params, false);
    let (two_phase, _, _) = one_circuit(&params, true);
    // ---- on-disk formats (the reference's calls: gen_srs / read_pk / gen_snark_shplonk(.., Some(path)) in
    // /root/reference/src/bin/cli.rs:222,312,320,478-483)
    let mut params_bytes = Vec::new();
    params.write(&mut params_bytes).expect("ParamsKZG::write");
    let mut pk_bytes = Vec::new();
    pk.write(&mut pk_bytes, SerdeFormat::RawBytesUnchecked).expect("ProvingKey::write");
    let snark = snark_verifier_sdk::halo2::gen_snark_shplonk(&params, &pk, circuit.clone(), None::<&str>);
    let snark_bytes = bincode::serialize(&snark).expect("bincode Snark");
    let protocol_len = bincode::serialize(&snark.protocol).expect("bincode protocol").len();
    let files = json!({
        "params": hex::encode(&params_bytes),
        "pk": hex::encode(&pk_bytes),
        "pk_n_selectors": pk.get_vk().cs().num_selectors(),
        "snark": hex::encode(&snark_bytes),
        "snark_protocol_len": protocol_len,
        "snark_proof": hex::encode(&snark.proof),
        "snark_instances": snark.instances.iter().map(|c| c.iter().map(hex_fr).collect::<Vec<_>>()).collect::<Vec<_>>(),
        "snark_note": "gen_snark_shplonk draws its own rng: the proof inside differs from proofs.poseidon, the LAYOUT is what is pinned",
    });
    json!({
        "srs_trapdoor": format!("{:x}", SRS_TRAPDOOR), "rng_seed": RNG_SEED,
        "rng_spec": "SplitMix64 sequence (state += 0x9E3779B97F4A7C15 before each output); an Fr::random = 8 consecutive u64, little-endian 512-bit value mod r",
        "evaluate_h_note": "Evaluator::evaluate_h is pub(in crate::plonk): its output is pinned through the quotient commitments inside every proof and upstream's verify_proof (verified)",
        "small": small, "two_phase": two_phase, "files": files,
    })
}
