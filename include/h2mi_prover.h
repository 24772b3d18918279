/* h2mi_prover.h — the device-resident Halo2/KZG prover of libh2mi.so behind a phase-level C ABI.
 *
 * What it replaces.  The reference's callers make ONE call,
 *     create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, Blake2bWrite<..>, _>(
 *         &params, &pk, &[circuit], &[instances], rng, &mut transcript)
 * (reference examples/standard_plonk.rs:40-50, src/scaffold.rs:190-200, 322-331), after keygen_vk / keygen_pk
 * (examples/standard_plonk.rs:33-34, src/scaffold.rs:132,135,284,287).  Inside the crate that call alternates between
 * the transcript (absorb commitments, squeeze a challenge) and heavy vector work that depends on the challenge just
 * squeezed.  This header cuts create_proof at exactly those points:
 *
 *     caller (Rust fork / C++ / Python)                      libh2mi.so
 *     ---------------------------------                      -----------------------------------------------------------
 *     witness generation (synthesize)      --- cells --->    h2mi_prover_advice      columns to HBM, blinding rows, commitments
 *     transcript: write points, squeeze theta
 *                                          --- theta --->    h2mi_prover_lookups     permuted input / table columns, commitments
 *     write points, squeeze beta, gamma    --- beta,gamma -> h2mi_prover_products    grand products, random polynomial, commitments
 *     write points, squeeze y              --- y --------->  h2mi_prover_quotient    evaluate_h / (X^n - 1), h pieces, commitments
 *     write points, squeeze x              --- x --------->  h2mi_prover_evaluations every queried evaluation, in write order
 *     write scalars, squeeze y', v         --- y', v ---->   h2mi_prover_shplonk_quotient   h(X) of ProverSHPLONK, commitment
 *     write point, squeeze u               --- u --------->  h2mi_prover_shplonk_open       L(X) / (X - u), commitment
 *     write point, finalize
 *
 * The caller owns witness generation and the Blake2b transcript (and therefore vk.transcript_repr, which the crate derives
 * from its own Debug text: nothing in the library depends on it).  The library owns every pass over a vector: workspaces,
 * streams, MSM batching, flush / join order, the coefficient and extended-coset forms, SHPLONK's rotation sets.  Only
 * 64-byte points and 32-byte scalars cross the boundary after the witness.
 *
 * The constraint system is DATA (h2mi_constraint_system).  Shapes accepted are the ones the reference proves:
 *   H2MI_GATES_STANDARD_PLONK  src/circuits/standard_plonk.rs:29-48 — q_a a + q_b b + q_c c + q_ab a b + constant
 *   H2MI_GATES_FLEX_VERTICAL   halo2-base's FlexGate through scaffold::prove (src/scaffold.rs:246-366, 379-485):
 *                              per gate column q (a + a(wX) a(w^2 X) - a(w^3 X)); up to H2MI_MAX_GATES (32) gate columns, up to
 *                              H2MI_MAX_LOOKUPS (8) single-expression lookups (a lookup-advice column, or q_lookup * a), one instance
 *                              column, up to H2MI_MAX_PERM (64) equality-enabled columns: every column count
 *                              `builder.config(k, Some(minimum_rows))` takes for a circuit that fills a few dozen columns
 * Every function returns H2MI_OK or a negative H2MI_E* code (h2mi.h); no exception crosses the boundary.  Field elements
 * and points use the layouts of h2mi.h (4 / 8 uint64 limbs, Montgomery form).  A prover object is used by one thread at a time.
 */
#ifndef H2MI_PROVER_H
#define H2MI_PROVER_H

#include "h2mi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define H2MI_COL_ADVICE 0u
#define H2MI_COL_FIXED 1u
#define H2MI_COL_INSTANCE 2u
typedef struct {
  uint32_t kind;  /* H2MI_COL_* */
  uint32_t index; /* index within its kind */
} h2mi_column;
typedef struct {
  uint32_t column;  /* index within the kind the query list belongs to */
  int32_t rotation; /* Rotation(r): the query opens the column at omega^r x */
} h2mi_query;

#define H2MI_GATES_STANDARD_PLONK 1u
#define H2MI_GATES_FLEX_VERTICAL 2u

#define H2MI_MAX_GATES H2MI_FLEX_MAX_GATES     /* 32 */
#define H2MI_MAX_PERM H2MI_FLEX_MAX_PERM       /* 64 */
#define H2MI_MAX_LOOKUPS H2MI_FLEX_MAX_LOOKUPS /* 8 */
#define H2MI_MAX_ADVICE 64
#define H2MI_MAX_FIXED 64
#define H2MI_MAX_QUERIES 192

typedef struct {
  h2mi_column input;        /* the lookup's input column (advice) */
  int32_t selector_fixed;   /* >= 0: the input expression is fixed[selector_fixed] * input (halo2-base's single-column q_lookup); -1: the column itself */
  uint32_t table_fixed;     /* the fixed column holding the table */
} h2mi_lookup;

/* ConstraintSystem<Fr> after configure(): the numbers create_proof reads off `pk.vk.cs` */
typedef struct {
  uint32_t k;                 /* rows = 2^k */
  uint32_t n_advice, n_fixed; /* at most H2MI_MAX_ADVICE / H2MI_MAX_FIXED */
  uint32_t n_instance;        /* 0 or 1 instance columns */
  uint32_t degree;            /* cs.degree(): extended domain 2^ceil(log2((degree - 1) n)), degree - 1 h pieces, permutation chunks of degree - 2 */
  uint32_t blinding_factors;  /* cs.blinding_factors() */
  uint32_t gates;             /* H2MI_GATES_* */
  uint32_t n_gates;           /* FLEX_VERTICAL: vertical gates (1 .. H2MI_MAX_GATES); STANDARD_PLONK: ignored (advice 0..2, fixed 0..4) */
  uint32_t gate_advice[H2MI_MAX_GATES];   /* advice column of gate g */
  uint32_t gate_selector[H2MI_MAX_GATES]; /* fixed column holding its selector */
  uint32_t n_perm;                         /* equality-enabled columns, in permutation-argument order */
  h2mi_column perm_columns[H2MI_MAX_PERM];
  uint32_t n_lookups;
  h2mi_lookup lookups[H2MI_MAX_LOOKUPS];
  uint32_t n_advice_queries, n_fixed_queries; /* in creation order: the order create_proof writes their evaluations */
  h2mi_query advice_queries[H2MI_MAX_QUERIES];
  h2mi_query fixed_queries[H2MI_MAX_QUERIES];
} h2mi_constraint_system;

/* assigned cells of one column: `count` values at `rows` (any order, each row once), or at rows 0 .. count - 1 when rows
 * is NULL; every other row of the column is zero.  values: count x 4 limbs.  H2MI_CELLS_CANONICAL in `flags`: the values
 * are canonical little-endian integers below r and the device converts them (for hosts without Montgomery arithmetic). */
#define H2MI_CELLS_CANONICAL 1u
typedef struct {
  const uint32_t* rows;
  const uint64_t* values;
  size_t count;
  uint32_t flags;
} h2mi_column_cells;

typedef struct h2mi_pk_s* h2mi_pk_t;         /* keygen_pk's ProvingKey, resident in HBM */
typedef struct h2mi_prover_s* h2mi_prover_t; /* the buffers, streams and phase state of one create_proof at a time; reused from proof to proof */

/* ---- keygen_vk + keygen_pk (examples/standard_plonk.rs:33-34; src/scaffold.rs:284,287) ----------------------------------------
 * fixed: cs->n_fixed columns as the circuit's synthesize() (without witnesses) assigns them — selectors included, as the columns keygen
 * appends for them; a lookup's table column is the whole table.  copies: n_copies x 4 uint32 = (left column, left row, right column,
 * right row) per constrain_equal call, in call order, columns as indices into cs->perm_columns (permutation/keygen.rs Assembly::copy;
 * the order decides the sigma polynomials).  g_lagrange_handle: the FULL Lagrange SRS (h2mi_bases_register*) of 2^k points — keygen
 * commits the fixed and sigma columns against it.  flags: H2MI_KEYGEN_VK_ONLY builds only what keygen_vk returns (the commitments).
 * H2MI_ERANGE: a fixed cell or a copy constraint on a row at or beyond 2^k - blinding_factors - 1 (the crate's NotEnoughRowsAvailable).
 * The pk holds: fixed / sigma columns in Lagrange, coefficient and extended-coset form, l_0 / l_last / l_active cosets, the support
 * of the copy constraints, each lookup table's sorted distinct values. */
#define H2MI_KEYGEN_VK_ONLY 1u
int h2mi_prover_keygen(const h2mi_constraint_system* cs, uint64_t g_lagrange_handle, const h2mi_column_cells* fixed, const uint32_t* copies,
                       size_t n_copies, unsigned flags, h2mi_pk_t* pk_out);
int h2mi_prover_pk_release(h2mi_pk_t pk); /* H2MI_EINVAL while a prover created against it is alive */
/* VerifyingKey::{fixed_commitments, permutation.commitments}: affine points (8 limbs each); either pointer may be NULL */
int h2mi_prover_vk_commitments(h2mi_pk_t pk, uint64_t* fixed_out /* n_fixed x 8 */, uint64_t* permutation_out /* n_perm x 8 */);

/* ---- one prover per (pk, SRS) ----------------------------------------------------------------------------------------------
 * g_handle / g_lagrange_handle: the base sets commitments are made against.  base_lo, base_count: they hold bases
 * [base_lo, base_lo + base_count) of the 2^k — the whole SRS (0, 2^k), or one rank's contiguous slice of it in the
 * one-process-per-GPU deployment: every commitment is then this rank's PARTIAL point and a combiner must be set. */
int h2mi_prover_create(h2mi_pk_t pk, uint64_t g_handle, uint64_t g_lagrange_handle, size_t base_lo, size_t base_count, h2mi_prover_t* prover_out);
int h2mi_prover_destroy(h2mi_prover_t prover);
/* sliced SRS: the phase's commitments are written as 96-byte Jacobian partial points to d_partial + 96 slot (slot < the largest of
 * h2mi_prover_counts' advice / lookups / products / quotient — the size both buffers must have, in points; 8 covers the reference's
 * StandardPlonk and one-column halo2-lib shapes); when a phase
 * reads its points back the library joins its MSM pipeline, calls combine(ctx, count) — which must leave the sums over all ranks of
 * slots 0 .. count - 1 at d_combined + 96 slot, ordered on the library's stream (h2mi_library_stream) or complete on return: an RCCL
 * all-gather + h2mi_g1_fold_groups_dev — and reads d_combined.  A nonzero return from combine fails the phase with H2MI_EHIP.
 * The callback runs on the calling thread, inside the phase call. */
typedef int (*h2mi_combine_fn)(void* ctx, size_t count);
int h2mi_prover_set_combiner(h2mi_prover_t prover, void* d_partial, void* d_combined, h2mi_combine_fn combine, void* ctx);

/* Where the blinding scalars come from (the crate takes them from its `rng` argument).  Default: counter-based SplitMix64 streams of the
 * 32-bit `seed` given to h2mi_prover_advice — reproducible, what the goldens and benchmarks use, NOT hiding against anyone who can guess
 * 32 bits.  With a key: every blinding scalar of the following proofs is Fr::from_u512 of one ChaCha20 block (RFC 7539 block function with
 * a 64-bit block counter and a 64-bit stream id, the layout of rand_chacha's ChaCha20Rng; one block per scalar, as `Fr::random(rng)`
 * consumes it) under this 256-bit key: block counter = the scalar's index, stream id = nonce << 3 | purpose (1 advice blinding rows,
 * 2 permutation products, 3 the vanishing argument's random polynomial, 4 permuted lookup columns, 5 lookup products), nonce = the `seed`
 * argument of h2mi_prover_advice (then below 2^61: a per-proof counter).  A fork fills the key from its rng once per prover.
 * key = NULL returns to the seeded streams.  Abandons a proof in flight. */
int h2mi_prover_set_rng_key(h2mi_prover_t prover, const uint8_t key[32]);

/* ---- the phases, in create_proof's order.  Every phase must be called once per proof, in this order (H2MI_EINVAL otherwise);
 * h2mi_prover_advice starts a new proof at any time.  points_out receive affine points, 8 limbs each, in the order create_proof writes
 * them to the transcript; the identity comes back as (0, 0) (the crate's transcript refuses it).  Each call returns when its points /
 * scalars are on the host; work that needs no further challenge (coefficient and extended forms of the columns just committed) keeps
 * running on the device behind it. */

/* advice[c]: the witness cells of advice column c (cs->n_advice of them), rows below 2^k - blinding_factors - 1.  instance: the public
 * inputs of the instance column (count values, Montgomery; the caller hashes them into its transcript itself).  seed: stands where the
 * crate takes `rng` — every blinding scalar is drawn from counter-based SplitMix64 streams of this seed (h2mi_fr_random_dev's
 * generator: seed + 1 advice blinding rows, + 2 permutation products, + 3 the vanishing argument's random polynomial, + 4 permuted
 * lookup columns, + 5 lookup products; seed < 2^32), or — after h2mi_prover_set_rng_key — the per-proof nonce of the keyed ChaCha20
 * streams (< 2^61).
 * points_out: n_advice commitments. */
int h2mi_prover_advice(h2mi_prover_t prover, const h2mi_column_cells* advice, const uint64_t* instance, size_t n_instance_values, uint64_t seed,
                       uint64_t* points_out);
/* theta is accepted for the crate's multi-expression lookups and unused by the single-expression ones.
 * points_out: per lookup the permuted input, then the permuted table commitment (2 x n_lookups; nothing without lookups — the call
 * may then be skipped).  H2MI_EUNSAT: a lookup input is not a table value. */
int h2mi_prover_lookups(h2mi_prover_t prover, const uint64_t theta[4], uint64_t* points_out);
/* points_out: the permutation argument's ceil(n_perm / (degree - 2)) grand products, one product per lookup, then the vanishing
 * argument's random polynomial: the order create_proof commits (and writes) them in. */
int h2mi_prover_products(h2mi_prover_t prover, const uint64_t beta[4], const uint64_t gamma[4], uint64_t* points_out);
/* points_out: the degree - 1 pieces of h(X) */
int h2mi_prover_quotient(h2mi_prover_t prover, const uint64_t y[4], uint64_t* points_out);
/* evals_out: every evaluation create_proof writes, in its order: advice queries, fixed queries, the random polynomial, the sigma
 * polynomials, per permutation product z(x), z(omega x) and — all but the last — z(omega^-(blinding_factors + 1) x), per lookup
 * z(x), z(omega x), A'(x), A'(omega^-1 x), S'(x).  h2mi_prover_num_evaluations gives the count (4 limbs each). */
int h2mi_prover_num_evaluations(h2mi_prover_t prover, size_t* count_out);
int h2mi_prover_evaluations(h2mi_prover_t prover, const uint64_t x[4], uint64_t* evals_out);
/* ProverSHPLONK::create_proof (poly/kzg/multiopen/shplonk/prover.rs) over the queries create_proof collects, cut at its two commitments */
int h2mi_prover_shplonk_quotient(h2mi_prover_t prover, const uint64_t y[4], const uint64_t v[4], uint64_t point_out[8]);
int h2mi_prover_shplonk_open(h2mi_prover_t prover, const uint64_t u[4], uint64_t point_out[8]);

/* number of points the phases return, so that a caller can size buffers from the constraint system alone */
typedef struct {
  uint32_t advice, lookups, products, quotient, evaluations;
} h2mi_prover_counts;
int h2mi_prover_get_counts(h2mi_prover_t prover, h2mi_prover_counts* out);

/* ---- device-resident intermediates, for callers that check or reuse them (the test-suite evaluates the quotient identity on them):
 * d_ptr_out / count_out receive the vector's address and its length in field elements.  Valid until the prover / pk is destroyed;
 * contents are those of the last proof. */
enum {
  H2MI_BUF_ADVICE = 0, H2MI_BUF_ADVICE_POLY, H2MI_BUF_ADVICE_COSET, H2MI_BUF_INSTANCE,
  H2MI_BUF_PERM_Z, H2MI_BUF_PERM_Z_POLY, H2MI_BUF_PERM_Z_COSET,
  H2MI_BUF_LOOKUP_PERMUTED_INPUT, H2MI_BUF_LOOKUP_PERMUTED_TABLE, H2MI_BUF_LOOKUP_Z,
  H2MI_BUF_RANDOM_POLY, H2MI_BUF_H /* (degree - 1) n coefficients: piece i at i n */, H2MI_BUF_H_POLY,
  H2MI_BUF_SHPLONK_H, H2MI_BUF_SHPLONK_H2,
  H2MI_PKBUF_FIXED = 64, H2MI_PKBUF_FIXED_POLY, H2MI_PKBUF_FIXED_COSET, H2MI_PKBUF_SIGMA, H2MI_PKBUF_SIGMA_POLY, H2MI_PKBUF_SIGMA_COSET,
  H2MI_PKBUF_L0_COSET, H2MI_PKBUF_L_LAST_COSET, H2MI_PKBUF_L_ACTIVE_COSET
};
int h2mi_prover_buffer(h2mi_prover_t prover, uint32_t kind, uint32_t index, void** d_ptr_out, size_t* count_out);
int h2mi_prover_pk_buffer(h2mi_pk_t pk, uint32_t kind, uint32_t index, void** d_ptr_out, size_t* count_out);

#ifdef __cplusplus
}
#endif
#endif /* H2MI_PROVER_H */
