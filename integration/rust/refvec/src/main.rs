//! Reference-held vectors for the hot path, produced by the crates the reference pins (NOT by libzkhip):
//!     cargo run --release > …/tests/golden/reference_vectors.json
//! Inputs are the repository's seeded synthetic values (oracle/pyref.py `synth_raw253`: SplitMix64 words, top bits masked to 253), so
//! the test suite regenerates them and compares the library's outputs with what upstream computed.  Settles the open items of
//! SURVEY.md §8(c): Fr::ZETA, the compressed-point flag bits, Blake2b / Keccak / Poseidon transcript encodings.
use ff::{Field, PrimeField, WithSmallOrderMulGroup};
use group::{Curve, GroupEncoding};
use halo2_proofs::arithmetic::{best_fft, best_multiexp};
use halo2_proofs::poly::EvaluationDomain;
use halo2_proofs::transcript::{Blake2bWrite, Challenge255, Transcript, TranscriptWrite, TranscriptWriterBuffer};
use halo2curves::bn256::{Fr, G1Affine, G1};
use serde_json::json;

mod prover_vectors;

fn splitmix64(x: u64) -> u64 {
    let mut z = x.wrapping_add(0x9E3779B97F4A7C15);
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}
/// oracle/pyref.py synth_raw253(seed, idx): four SplitMix64 words, the top one masked to 61 bits; taken as the RAW Montgomery limbs
fn synth_raw253(seed: u64, idx: u64) -> Fr {
    let mut w = [0u64; 4];
    for (limb, slot) in w.iter_mut().enumerate() {
        *slot = splitmix64(seed.wrapping_add((idx * 4 + limb as u64).wrapping_mul(0x2545F4914F6CDD1D)));   // pyref.synth_word
    }
    w[3] &= (1u64 << 61) - 1;
    unsafe { std::mem::transmute::<[u64; 4], Fr>(w) } // raw limbs = Montgomery form, as the repo's synthetic tables define them
}
fn hex_fr(x: &Fr) -> String {
    let mut b = x.to_repr().as_ref().to_vec();
    b.reverse();
    hex::encode(b)
}
fn hex_pt(p: &G1Affine) -> serde_json::Value {
    json!({ "compressed": hex::encode(p.to_bytes().as_ref()), "x": format!("{:?}", p.x), "y": format!("{:?}", p.y) })
}

fn main() {
    let g = G1Affine::generator();
    // --- field constants and encodings
    let constants = json!({
        "zeta": hex_fr(&Fr::ZETA), "delta": hex_fr(&Fr::DELTA), "root_of_unity": hex_fr(&Fr::ROOT_OF_UNITY), "s": Fr::S,
        "generator_compressed": hex::encode(g.to_bytes().as_ref()),
        "identity_compressed": hex::encode(G1Affine::default().to_bytes().as_ref()),
        "two_g": hex_pt(&(G1::from(g) + G1::from(g)).to_affine()),
        "neg_g_compressed": hex::encode((-g).to_bytes().as_ref()),
    });
    // --- MSM: n points [s^i] G (s = 0x1D5C0FFEE as in the tests), seeded scalars
    let s = Fr::from(0x1D5C0FFEEu64);
    let mut msm = Vec::new();
    for (k, seed) in [(8u32, 11u64), (12, 12)] {
        let n = 1usize << k;
        let mut bases = Vec::with_capacity(n);
        let mut cur = Fr::ONE;
        for _ in 0..n {
            bases.push((g * cur).to_affine());
            cur *= s;
        }
        let coeffs: Vec<Fr> = (0..n as u64).map(|i| synth_raw253(seed, i)).collect();
        let sum = best_multiexp(&coeffs, &bases).to_affine();
        msm.push(json!({ "k": k, "seed": seed, "srs_trapdoor": "1d5c0ffee", "sum": hex_pt(&sum) }));
    }
    // --- FFT and EvaluationDomain
    let mut fft = Vec::new();
    for (k, seed) in [(4u32, 21u64), (10, 22)] {
        let n = 1usize << k;
        let mut a: Vec<Fr> = (0..n as u64).map(|i| synth_raw253(seed, i)).collect();
        let mut omega = Fr::ROOT_OF_UNITY;
        for _ in k..Fr::S {
            omega = omega.square();
        }
        best_fft(&mut a, omega, k);
        fft.push(json!({ "k": k, "seed": seed, "omega": hex_fr(&omega), "first": hex_fr(&a[0]), "second": hex_fr(&a[1]), "last": hex_fr(&a[n - 1]),
                         "all": if k <= 4 { a.iter().map(hex_fr).collect::<Vec<_>>() } else { vec![] } }));
    }
    let dom = EvaluationDomain::<Fr>::new(4, 6);
    let mut lag = dom.empty_lagrange();
    for (i, v) in lag.iter_mut().enumerate() {
        *v = synth_raw253(31, i as u64);
    }
    let coeff = dom.lagrange_to_coeff(lag);
    let ext = dom.coeff_to_extended(coeff.clone());
    let domain = json!({ "j": 4, "k": 6, "extended_k": dom.extended_k(), "seed": 31,
                         "coeff": coeff.iter().map(hex_fr).collect::<Vec<_>>(), "extended": ext.iter().map(hex_fr).collect::<Vec<_>>() });
    // --- transcripts: one fixed sequence under Blake2b (halo2) and Poseidon / Keccak (snark-verifier)
    let p1 = (g * Fr::from(5u64)).to_affine();
    let sc = Fr::from(0x1234567890ABCDEFu64);
    let blake = {
        let mut t = Blake2bWrite::<Vec<u8>, G1Affine, Challenge255<G1Affine>>::init(vec![]);
        t.common_scalar(Fr::from(7u64)).unwrap();
        t.write_point(p1).unwrap();
        t.write_scalar(sc).unwrap();
        let c1: Fr = *t.squeeze_challenge_scalar::<()>();
        let c2: Fr = *t.squeeze_challenge_scalar::<()>();
        json!({ "c1": hex_fr(&c1), "c2": hex_fr(&c2), "proof": hex::encode(t.finalize()) })
    };
    let poseidon = {
        use snark_verifier_sdk::halo2::{PoseidonTranscript, POSEIDON_SPEC};
        use snark_verifier_sdk::snark_verifier::loader::native::NativeLoader;
        let mut t = PoseidonTranscript::<NativeLoader, Vec<u8>>::from_spec(vec![], POSEIDON_SPEC.clone());
        t.common_scalar(Fr::from(7u64)).unwrap();
        t.write_point(p1).unwrap();
        t.write_scalar(sc).unwrap();
        let c1: Fr = *t.squeeze_challenge_scalar::<()>();
        let c2: Fr = *t.squeeze_challenge_scalar::<()>();
        json!({ "c1": hex_fr(&c1), "c2": hex_fr(&c2), "proof": hex::encode(t.finalize()) })
    };
    let evm = {
        use snark_verifier_sdk::snark_verifier::system::halo2::transcript::evm::EvmTranscript;
        let mut t = EvmTranscript::<G1Affine, _, _, _>::new(vec![]);
        t.common_scalar(Fr::from(7u64)).unwrap();
        t.write_point(p1).unwrap();
        t.write_scalar(sc).unwrap();
        let c1: Fr = *t.squeeze_challenge_scalar::<()>();
        let c2: Fr = *t.squeeze_challenge_scalar::<()>();
        json!({ "c1": hex_fr(&c1), "c2": hex_fr(&c2), "proof": hex::encode(t.finalize()) })
    };
    let out = json!({ "source": "halo2curves e185711 / halo2_proofs 4b42325 / snark-verifier-sdk 7011e8c (the reference's pins)",
                      "constants": constants, "msm": msm, "fft": fft, "domain": domain,
                      // whole create_proof runs (small(6), two_phase(6)) under the three transcripts with a replayable rng, the verifying key's
                      // transcript representation, and the bytes of ParamsKZG::write / ProvingKey::write / bincode(Snark): prover_vectors.rs
                      "prover": prover_vectors::emit(),
                      "transcripts": { "sequence": "common_scalar(7), write_point(5 G), write_scalar(0x1234567890abcdef), squeeze, squeeze",
                                       "blake2b": blake, "poseidon": poseidon, "evm": evm } });
    println!("{}", serde_json::to_string_pretty(&out).unwrap());
}
