//! Prints one JSON object with the values this repository's oracle restates from memory of the crates behind the reference
//! (DCMMC/halo2-scaffold: Cargo.toml:13,16,18; src/scaffold.rs:14; examples/standard_plonk.rs:26-50).
//!     cargo run --release > probe.json && python3 tools/compare_probe.py probe.json
//! Every key is compared with tests/golden/recall_expectations.json; keys under "informational" have no expectation (they
//! depend on the crate's Debug text or on how it consumes its rng) and are printed for the record.
//! Compiled by nobody so far: there is no Rust toolchain where this repository is built.
use halo2_proofs::{
    arithmetic::{best_fft, best_multiexp},
    circuit::Value,
    halo2curves::{
        bn256::{Bn256, Fq, Fr, G1Affine, G1},
        group::{ff::PrimeField, prime::PrimeCurveAffine, Curve, Group, GroupEncoding},
        FieldExt,
    },
    plonk::{create_proof, keygen_pk, keygen_vk},
    poly::{
        commitment::{Blind, ParamsProver},
        kzg::{
            commitment::{KZGCommitmentScheme, ParamsKZG},
            multiopen::ProverSHPLONK,
        },
        EvaluationDomain,
    },
    transcript::{Blake2bWrite, Challenge255, EncodedChallenge, Transcript, TranscriptWrite, TranscriptWriterBuffer},
};
use halo2_base::{
    gates::{GateChip, GateInstructions, RangeChip, RangeInstructions},
    AssignedValue, Context,
    QuantumCell::{Constant, Existing, Witness},
};
use halo2_scaffold::circuits::standard_plonk::StandardPlonk;
use halo2_scaffold::scaffold::gen_key;
use rand::SeedableRng;
use rand_chacha::ChaCha20Rng;

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{:02x}", x)).collect()
}
fn repr(x: &Fr) -> String {
    hex(x.to_repr().as_ref())
}
/// the four 64-bit words an Fr / Fq occupies in memory (the Montgomery form the C ABI exchanges), least significant first
fn limbs_fr(x: Fr) -> Vec<String> {
    let l: [u64; 4] = unsafe { std::mem::transmute(x) };
    l.iter().map(|w| format!("{:016x}", w)).collect()
}
fn limbs_fq(x: Fq) -> Vec<String> {
    let l: [u64; 4] = unsafe { std::mem::transmute(x) };
    l.iter().map(|w| format!("{:016x}", w)).collect()
}
fn limbs_g1(p: &G1Affine) -> Vec<String> {
    let mut v = limbs_fq(p.x);
    v.extend(limbs_fq(p.y));
    v
}
fn point(p: &G1Affine) -> String {
    hex(p.to_bytes().as_ref())
}
fn list(v: Vec<String>) -> String {
    format!("[{}]", v.iter().map(|s| format!("\"{}\"", s)).collect::<Vec<_>>().join(", "))
}

/// the instruction sequence of the reference's examples/halo2_lib.rs:14-60 (x^2 + 72 three ways; x and the first result public)
fn halo2_lib_body(ctx: &mut Context<Fr>, x: Fr, make_public: &mut Vec<AssignedValue<Fr>>) {
    let x = ctx.load_witness(x);
    make_public.push(x);
    let gate = GateChip::<Fr>::default();
    let x_sq = gate.mul(ctx, x, x);
    let c = Fr::from(72);
    let out = gate.add(ctx, x_sq, Constant(c));
    make_public.push(out);
    let val = *x.value() * x.value() + c;
    ctx.assign_region_last([Constant(c), Existing(x), Existing(x), Witness(val)], [0]);
    gate.mul_add(ctx, x, x, Constant(c));
}
/// the instruction sequence of examples/range.rs:10-34 (x public; range_check(x, 64); x + x)
fn range_body(ctx: &mut Context<Fr>, x: Fr, make_public: &mut Vec<AssignedValue<Fr>>) {
    let lookup_bits = std::env::var("LOOKUP_BITS").unwrap().parse().unwrap();
    let x = ctx.load_witness(x);
    make_public.push(x);
    let range = RangeChip::default(lookup_bits);
    range.range_check(ctx, x, 64);
    range.gate().add(ctx, x, x);
}

/// keygen through the reference's scaffold::gen_key at DEGREE / LOOKUP_BITS -> (fixed commitments, permutation commitments, break
/// points of phase 0, the FLEX_GATE_CONFIG_PARAMS builder.config left in the environment)
fn keygen_case(degree: &str, lookup_bits: Option<&str>, range_closure: bool) -> (String, String, String, String) {
    std::env::set_var("DEGREE", degree);
    match lookup_bits {
        Some(b) => std::env::set_var("LOOKUP_BITS", b),
        None => std::env::remove_var("LOOKUP_BITS"),
    }
    let (pk, break_points) =
        if range_closure { gen_key(range_body, Fr::from(0xdeadbeefcafe1234u64)) } else { gen_key(halo2_lib_body, Fr::from(12)) };
    let vk = pk.get_vk();
    (
        list(vk.fixed_commitments().iter().map(point).collect()),
        list(vk.permutation().commitments().iter().map(point).collect()),
        format!("{:?}", break_points[0]),
        std::env::var("FLEX_GATE_CONFIG_PARAMS").unwrap_or_default(),
    )
}

fn main() {
    let mut out: Vec<(String, String)> = vec![];
    let mut put = |k: &str, v: String| out.push((k.to_string(), v));
    let q = |s: String| format!("\"{}\"", s);

    // ---- field constants and layouts (halo2curves::bn256::Fr; reference src/scaffold.rs:14)
    put("fr_five_limbs", list(limbs_fr(Fr::from(5))));
    put("fr_one_limbs", list(limbs_fr(Fr::one())));
    put("fr_five_repr", q(repr(&Fr::from(5))));
    put("fr_root_of_unity_repr", q(repr(&Fr::root_of_unity())));
    put("fr_zeta_repr", q(repr(&<Fr as FieldExt>::ZETA)));
    put("fr_delta_repr", q(repr(&<Fr as FieldExt>::DELTA)));
    put("fr_s", format!("{}", Fr::S));

    // ---- G1: generator, identity, compressed encodings
    let g = G1Affine::generator();
    let id = G1Affine::identity();
    put("g1_generator_limbs", list(limbs_g1(&g)));
    put("g1_identity_limbs", list(limbs_g1(&id)));
    put("g1_generator_bytes", q(point(&g)));
    put("g1_identity_bytes", q(point(&id)));
    let two_g = (G1::generator() + G1::generator()).to_affine();
    put("g1_two_g_bytes", q(point(&two_g)));
    let minus_g = (-G1::generator()).to_affine();
    put("g1_minus_g_bytes", q(point(&minus_g)));

    // ---- best_multiexp / best_fft on a fixed 2^4 input: s_i = 1000003 i + 7, P_i = (i + 1) G, a_i = 3 i + 1
    let scalars: Vec<Fr> = (0..16u64).map(|i| Fr::from(1000003 * i + 7)).collect();
    let bases: Vec<G1Affine> = (0..16u64).map(|i| (G1::generator() * Fr::from(i + 1)).to_affine()).collect();
    put("msm16_bytes", q(point(&best_multiexp(&scalars, &bases).to_affine())));
    let k = 4u32;
    let domain = EvaluationDomain::<Fr>::new(3, k);
    let mut a: Vec<Fr> = (0..16u64).map(|i| Fr::from(3 * i + 1)).collect();
    let omega = domain.get_omega();
    put("omega_k4_repr", q(repr(&omega)));
    best_fft(&mut a, omega, k);
    put("fft16_repr", list(a.iter().map(repr).collect()));

    // ---- EvaluationDomain::coeff_to_extended (the coset generator and the extended root of unity in one observable)
    let coeffs: Vec<Fr> = (0..16u64).map(|i| Fr::from(3 * i + 1)).collect();
    let ext = domain.coeff_to_extended(domain.coeff_from_vec(coeffs));
    put("coeff_to_extended_k4_j3_repr", list(ext.iter().map(repr).collect()));
    put("extended_k_k4_j3", format!("{}", domain.extended_k()));

    // ---- transcript: one point, one scalar, one challenge
    let mut t = Blake2bWrite::<_, G1Affine, Challenge255<_>>::init(vec![]);
    t.write_point(g).unwrap();
    t.write_scalar(Fr::from(5)).unwrap();
    let c: Challenge255<G1Affine> = t.squeeze_challenge();
    put("transcript_challenge_repr", q(repr(&c.get_scalar())));
    put("transcript_bytes", q(hex(&t.finalize())));

    // ---- the SRS the scaffold generates: gen_srs(k) = ParamsKZG::setup(k, ChaCha20Rng::from_seed(Default::default()))
    // (reference src/scaffold.rs:119,174,271 through halo2-base utils::fs)
    let k5 = 5u32;
    let params = ParamsKZG::<Bn256>::setup(k5, ChaCha20Rng::from_seed(Default::default()));
    put("srs5_g", list(params.get_g()[..3].iter().map(point).collect()));
    let d5 = EvaluationDomain::<Fr>::new(3, k5);
    let lagrange: Vec<String> = (0..3usize)
        .map(|i| {
            let mut e = vec![Fr::zero(); 1 << k5];
            e[i] = Fr::one();
            point(&params.commit_lagrange(&d5.lagrange_from_vec(e), Blind::default()).to_affine())
        })
        .collect();
    put("srs5_g_lagrange", list(lagrange));

    // ---- keygen of the reference's circuit at its own size (examples/standard_plonk.rs:26-34)
    let circuit = StandardPlonk { x: Value::unknown() };
    let vk = keygen_vk(&params, &circuit).expect("vk should not fail");
    put("vk5_fixed_commitments", list(vk.fixed_commitments().iter().map(point).collect()));
    put("vk5_permutation_commitments", list(vk.permutation().commitments().iter().map(point).collect()));
    let pinned = format!("{:?}", vk.pinned());
    let pk = keygen_pk(&params, vk, &circuit).expect("pk should not fail");

    // ---- one proof under a seeded rng (examples/standard_plonk.rs:37-50 with OsRng replaced)
    let circuit = StandardPlonk { x: Value::known(Fr::from(0xc0ffee)) };
    let mut transcript = Blake2bWrite::<_, _, Challenge255<_>>::init(vec![]);
    create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, Blake2bWrite<Vec<u8>, G1Affine, Challenge255<_>>, _>(
        &params,
        &pk,
        &[circuit],
        &[&[]],
        ChaCha20Rng::from_seed([7u8; 32]),
        &mut transcript,
    )
    .expect("prover should not fail");
    let proof = transcript.finalize();
    put("proof5_len", format!("{}", proof.len()));

    // ---- the halo2-lib builders through the reference's own scaffold::gen_key (src/scaffold.rs:95-155; it reads DEGREE / LOOKUP_BITS
    // from the environment, takes the column counts from builder.config(k, Some(9)) and the SRS from gen_srs(k) — the same
    // fixed-seed SRS as above, cached under ./params): the verifying key's commitments pin halo2-base's layout conventions, the
    // break points the rule that ends a gate column.  The closures are the reference's two examples.
    let (f, p, b, c1) = keygen_case("5", None, false);
    put("halo2lib_k5_fixed_commitments", f);
    put("halo2lib_k5_permutation_commitments", p);
    put("halo2lib_k5_break_points_phase0", b);
    let (f, p, b, c2) = keygen_case("7", Some("4"), true);
    put("range_k7_bits4_fixed_commitments", f);
    put("range_k7_bits4_permutation_commitments", p);
    put("range_k7_bits4_break_points_phase0", b);
    let (f, p, b, c3) = keygen_case("5", Some("4"), true);
    put("range_k5_bits4_fixed_commitments", f);
    put("range_k5_bits4_permutation_commitments", p);
    put("range_k5_bits4_break_points_phase0", b);
    // the Gate closure under the Range builder (LOOKUP_BITS set, nothing looked up: src/scaffold.rs:44-48 switches on the variable alone)
    let (f, p, b, c4) = keygen_case("6", Some("4"), false);
    put("halo2lib_range_builder_k6_bits4_fixed_commitments", f);
    put("halo2lib_range_builder_k6_bits4_permutation_commitments", p);
    put("halo2lib_range_builder_k6_bits4_break_points_phase0", b);
    let flex_config = vec![format!("\"halo2lib_range_builder_k6_bits4\": {:?}", c4), format!("\"halo2lib_k5\": {:?}", c1), format!("\"range_k7_bits4\": {:?}", c2), format!("\"range_k5_bits4\": {:?}", c3)];

    // informational: no expectation exists for these
    let info = format!(
        "{{\"vk5_pinned_debug\": {:?}, \"proof5_hex\": \"{}\", \"flex_gate_config_params\": {{{}}}}}",
        pinned,
        hex(&proof),
        flex_config.join(", ")
    );
    put("informational", info);

    println!("{{");
    for (i, (k, v)) in out.iter().enumerate() {
        println!("  \"{}\": {}{}", k, v, if i + 1 < out.len() { "," } else { "" });
    }
    println!("}}");
}
