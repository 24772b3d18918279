/* h2mi.h — C ABI of libh2mi.so: BN254 MSM + NTT backend for AMD Instinct MI355X (gfx950).
 *
 * Drop-in boundary for the hot path of the Halo2/KZG prover that DCMMC/halo2-scaffold drives through
 * create_proof() (reference examples/standard_plonk.rs:41-49, src/scaffold.rs:191-199,322-331).  The
 * reference itself has no FFI seam; the seam is Cargo's [patch] of its halo2_proofs dependency
 * (reference Cargo.toml:13,16), whose arithmetic::{best_multiexp, best_fft}, poly::EvaluationDomain and
 * poly::kzg::commitment::ParamsKZG call the functions below (binding shown in INTEGRATION.md).
 *
 * Data layouts are those of halo2curves::bn256 (reference src/scaffold.rs:14):
 *   Fr, Fq    4 x uint64 little-endian limbs, MONTGOMERY form (R = 2^256), fully reduced
 *   G1Affine  {x, y} = 8 x uint64; the identity is (0, 0)
 *   G1        {x, y, z} = 12 x uint64 Jacobian (x/z^2, y/z^3); the identity has z = 0
 *
 * Conventions: every function returns H2MI_OK (0) or a negative H2MI_E* code; h2mi_strerror() names
 * it.  No C++ exception crosses this boundary.  Pointers named d_* are DEVICE pointers (HBM of the
 * process's GPU); all others are host pointers owned by the caller for the duration of the call.
 * Multi-GPU, two ways: one process per GPU (h2mi_init picks the GPU; the job combines the 96-byte partial MSM
 * results itself: RCCL all-gather + h2mi_g1_fold_groups_dev), or ONE process driving n GPUs
 * (h2mi_init_devices: the bases are sharded at registration and every MSM folds its partial results internally).
 * Calls are serialised by an internal mutex, so any thread may call.
 * There is NO CPU fallback: without a usable GPU every compute entry point returns H2MI_ENODEV.
 */
#ifndef H2MI_H
#define H2MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define H2MI_OK 0
#define H2MI_EINVAL (-1)   /* bad argument (null pointer, size 0 where not allowed, log_n out of range) */
#define H2MI_ENODEV (-2)   /* no usable GPU / h2mi_init not called                                      */
#define H2MI_ENOMEM (-3)   /* device or host allocation failed                                          */
#define H2MI_EHIP (-4)     /* a HIP runtime call or kernel launch failed                                */
#define H2MI_EHANDLE (-5)  /* unknown or released bases handle                                          */
#define H2MI_ERANGE (-6)   /* n larger than the registered base count / unsupported size                */
#define H2MI_EUNSAT (-7)   /* h2mi_prover.h: the witness does not satisfy the circuit where the prover can see it
                              (a lookup input that is not a table value: the crate's Error::ConstraintSystemFailure) */

#define H2MI_MAX_LOG_N 28  /* largest NTT the device path accepts: the field's two-adicity (2^28 x 32 B = 8 GiB per buffer) */

typedef void* h2mi_stream_t; /* a hipStream_t, or NULL for the library's own stream */

/* ---- lifecycle ---------------------------------------------------------------------------------- */
/* Select GPU `device` (ordinal visible to this process) and create the library's stream.  Idempotent
 * for the same device.  Replaces nothing in the reference (which has no device); the Rust shim calls
 * it once from a `std::sync::Once`. */
int h2mi_init(int device);
/* One prover process, n devices (SURVEY.md 8b `h2mi_init(n_devices)`; the reference is one process calling
 * create_proof once, src/scaffold.rs:246-366): devices 0 .. n-1, device 0 the primary.  Afterwards
 *   h2mi_bases_register{,_dev}   shard the base set: one contiguous slice (and its window tables) per device;
 *   h2mi_msm_bn254_g1{,_dev}     launch every slice from the calling thread and fold the 96-byte partial results on
 *                                the primary device (host form: at once; _dev form, stream = NULL only: at the next
 *                                h2mi_join / h2mi_sync / h2mi_memcpy_d2h, like every queued MSM);
 * transforms, polynomial helpers and every d_* pointer stay on the primary device (NTT is single-GPU by design).
 * H2MI_VIRTUAL_DEVICES=1 lets n exceed the number of GPUs (entry i runs on GPU i mod count): a one-GPU rehearsal of
 * the sharding, scalar distribution and fold.  Mutually exclusive with h2mi_init. */
int h2mi_init_devices(int n_devices);
int h2mi_device_count(void); /* devices this process drives (0 before init) */
void h2mi_shutdown(void);
const char* h2mi_strerror(int code);
const char* h2mi_version(void);

/* ---- device memory (so a host can keep vectors resident between calls) ---------- */
int h2mi_malloc(size_t bytes, void** d_ptr);
int h2mi_free(void* d_ptr);
int h2mi_memcpy_h2d(void* d_dst, const void* src, size_t bytes);
/* queued on the library's stream without waiting for it (pageable `src` is staged by the runtime before the call
 * returns): for the small patches a prover writes into device-resident columns (assigned cells, blinding rows) */
int h2mi_memcpy_h2d_async(void* d_dst, const void* src, size_t bytes);
/* `count` 32-byte field elements (Montgomery limbs, host memory) written to `count` device addresses (16-byte aligned) by ONE kernel
 * launch per 64 cells — the cells travel in the launch's arguments, so `values` may be reused at once.  For the handful of assigned
 * cells and blinding rows a prover patches into zeroed columns (create_proof's `advice[column][row] = value`): a dozen 32-byte
 * hipMemcpyAsync calls cost ~10 us each on the library stream in front of a phase's first commitment.  Stream-ordered on `stream`
 * (NULL = the library stream); cells given twice keep one of the values. */
int h2mi_fr_patch_cells_dev(void* const* d_cells, const uint64_t* values, size_t count, h2mi_stream_t stream);
int h2mi_memcpy_d2h(void* dst, const void* d_src, size_t bytes);
int h2mi_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes);
int h2mi_memset_zero(void* d_ptr, size_t bytes); /* asynchronous on the library's stream */
int h2mi_sync(void); /* wait for all work queued on the library's streams */
/* device-side join: later work on the library's stream waits for every MSM queued so far to be complete.
 * An MSM queued on the library's stream (h2mi_msm_bn254_g1_dev with stream = NULL) only runs its bucket
 * partition and accumulation at once; the latency-bound bucket reduction is deferred and run for all MSMs
 * queued since the last join as one batch, so the 96-byte results are written only by h2mi_join / h2mi_sync /
 * h2mi_memcpy_d2h / h2mi_msm_flush — or when half of a handle's workspace slots are pending: two MSMs for base sets
 * above 2^17 points (four slots), four below (eight slots).  A prover calls h2mi_join where the transcript
 * needs the commitments of a phase.  MSMs on a caller-provided stream complete in order on that stream.
 * Base sets of at most 2^14 points defer their whole accumulate-and-finish launch the same way (H2MI_MSM_GENERAL below). */
int h2mi_join(void);
/* start the deferred bucket reductions of the MSMs queued so far NOW, on the library's reduction stream, without ordering
 * the library's stream behind them (h2mi_join still does that, later): a prover that has queued the commitments of a phase
 * and goes on to queue work that does not need them (the columns' transforms) calls this first, so the reductions run
 * beside that work instead of starting when the host reaches h2mi_join. */
int h2mi_msm_flush(void);
/* ---- side streams.  Every *_dev entry point takes a stream (NULL = the library's own, on which calls execute in issue
 * order).  A second stream lets work that does not depend on the next transcript challenge — e.g. the coefficient and
 * extended forms of the advice columns, needed only by evaluate_h — run beside the library stream's chain instead of
 * queueing behind it.  h2mi_stream_wait(waiter, signaller) makes everything queued later on `waiter` start after
 * everything queued so far on `signaller` (NULL = the library stream on either side); it does not block the host. */
int h2mi_stream_create(h2mi_stream_t* stream_out);
int h2mi_stream_destroy(h2mi_stream_t stream);
int h2mi_stream_wait(h2mi_stream_t waiter, h2mi_stream_t signaller);

/* the library's own stream (a hipStream_t), so that a host can order foreign work — an RCCL collective, its own
 * kernels — against the library's without a host synchronisation */
int h2mi_library_stream(void** stream_out);

/* ---- bases (the KZG SRS): ParamsKZG::{g, g_lagrange}, SURVEY.md 8a row a5 ------------------------
 * Replaces the `&params.g` / `&params.g_lagrange` slices that ParamsKZG::commit / commit_lagrange pass
 * to best_multiexp (reached from reference examples/standard_plonk.rs:33,34,41-49).  Registration
 * copies the n affine points to HBM and builds the per-window multiples table 2^(c*w) * P_i (all
 * windows then share one bucket set), once; commits afterwards move only scalars. */
int h2mi_bases_register(const uint64_t* bases /* n*8 limbs */, size_t n, uint64_t* handle_out);
int h2mi_bases_register_dev(const void* d_bases /* n*64 B */, size_t n, uint64_t* handle_out);
int h2mi_bases_release(uint64_t handle);
/* window bits c, window count W, bucket count, registered n */
int h2mi_bases_info(uint64_t handle, uint32_t* c, uint32_t* windows, uint32_t* buckets, uint64_t* n);

/* ---- MSM: halo2_proofs::arithmetic::best_multiexp(coeffs, bases) -> C::Curve ---------------------
 * Replaces best_multiexp for C = bn256::G1Affine; computes sum_i s_i * P_i over
 * the first n registered bases.  Scalars are Fr values exactly as they sit in memory (Montgomery).
 * The result is a Jacobian representative of the exact group element (the caller batch-normalises
 * before hashing, as create_proof does); identity is returned as (0, R, 0) = G1::identity().
 * Which representative (X : Y : Z) comes back is NOT reproducible between two calls on the same input:
 * the order of additions inside a bucket follows arrival order in the bucket partition (the reference's
 * representative likewise depends on its thread count).  h2mi_msm_set_canonical(1) makes every MSM end
 * with a normalisation to Z = 1 (bit-reproducible output, about 0.17 ms more latency per MSM).
 * The reference asserts coeffs.len() == bases.len(); here n > registered n returns H2MI_ERANGE. */
int h2mi_msm_bn254_g1(uint64_t handle_or_0, const uint64_t* bases_or_null /* used when handle==0 */,
                      const uint64_t* scalars /* n*4 limbs */, size_t n, uint64_t out_jacobian[12]);
/* handle == 0 (best_multiexp on a plain slice of bases) does NOT rebuild the window tables on every call: the bases
 * are uploaded, fingerprinted on the device over every byte and looked up in a cache of the four most recent ad-hoc
 * registrations, so repeated calls with the same slice pay the 64 B x n upload only.  Register explicitly
 * (h2mi_bases_register) to avoid that too.  h2mi_msm_adhoc_builds: how many ad-hoc registrations have been built. */
int h2mi_msm_adhoc_builds(uint64_t* builds_out);
/* device-resident form: scalars and the 96-byte result live in HBM; asynchronous on `stream`. */
int h2mi_msm_bn254_g1_dev(uint64_t handle, const void* d_scalars, size_t n, void* d_out_jacobian,
                          h2mi_stream_t stream);
/* `count` MSMs of n scalars each over ONE registered base set — the commitments of a prover phase (create_proof commits a phase's advice
 * columns, then its grand products, then the quotient's pieces, each group before one challenge: reference examples/standard_plonk.rs:41-49
 * through halo2_proofs' create_proof).  Result j goes to d_out_jacobian + 96 j; the results are the ones `count` calls of
 * h2mi_msm_bn254_g1_dev in the same order would produce.  For base sets of up to 2^17 points on the library stream the partition and the
 * accumulation of up to four MSMs run as ONE set of launches (the host could not issue a small MSM's eight launches as fast as the device
 * ran them: DESIGN.md 4.1); larger base sets, caller streams and sharded handles take the loop.  The scalars must stay untouched until work
 * queued on `stream` after this call would run (as for the single form).  flags:
 *   H2MI_MSM_SPARSE   the caller's promise that the columns are SPARSE — mostly zeros, or one value repeated almost everywhere (witness
 *                     columns of a padded circuit, permutation / lookup grand products): the kernels of such an MSM are short at every
 *                     size, so the batched launches are used above 2^17 points as well (narrow windows; 20-bit windows take the loop).
 *                     Results do not depend on the promise; dense columns passed with it only lose the overlap between one MSM's partition
 *                     and the previous one's accumulation.
 *   H2MI_MSM_INORDER  the group is all its phase commits and its points are read back next (a lone commitment: count = 1): partition,
 *                     accumulation AND bucket reductions run in order on one stream, nothing is deferred to the join — the library's
 *                     three-stream split overlaps CONSECUTIVE MSMs and costs a lone one ~50 us of stream hops.  A dense group above 2^17
 *                     points keeps the pipelined loop, whose overlap is worth more; sharded handles take the ordinary path.
 *   H2MI_MSM_GENERAL  take the general (bucket) pipeline even for a base set that has the latency path's table.  Base sets of at most 2^14
 *                     points take a latency path of their own (narrow windows against a table of every digit multiple — up to 2 GB per
 *                     handle, skipped when it does not fit —, two short kernels: DESIGN.md 4.1), which wins a lone commitment and a phase
 *                     of four; above 2^13 points the general pipeline wins once MSMs stream back to back.  Without this flag the library
 *                     switches such a handle over by itself after four MSMs have been issued without a join (h2mi_join / h2mi_sync /
 *                     h2mi_memcpy_d2h); a caller that knows it streams says so up front.  Results do not depend on the path. */
#define H2MI_MSM_SPARSE 1u
#define H2MI_MSM_INORDER 2u
#define H2MI_MSM_GENERAL 4u
int h2mi_msm_bn254_g1_phase_dev(uint64_t handle, const void* const* d_scalars, size_t count, size_t n, void* d_out_jacobian, unsigned flags,
                                h2mi_stream_t stream);
/* 1: MSM results are normalised to Z = 1 on the device (reproducible bits); 0 (default): raw sum. */
int h2mi_msm_set_canonical(int on);
/* number of bucket insertions (non-zero signed digits) the last MSM on this handle performed, and the
 * running-sum reduction adds — the numerator of "G1-adds/s" (SURVEY.md 8d).  Synchronises. */
int h2mi_msm_last_stats(uint64_t handle, uint64_t* bucket_adds, uint64_t* reduce_adds);
/* sum of k Jacobian points (k*12 limbs, host) -> one Jacobian point: the combine step of the sliced
 * multi-GPU MSM (the fold `results.iter().fold(identity, |a, b| a + b)` of best_multiexp). */
int h2mi_g1_sum_jacobian(const uint64_t* points, size_t k, uint64_t out_jacobian[12]);
/* the same fold for k MSMs at once: points[r*k + j] is rank r's partial result of MSM j -> out[j] */
int h2mi_g1_fold_groups(const uint64_t* points /* world*k*12 */, size_t world, size_t k, uint64_t* out /* k*12 */);
/* the same fold on device-resident points (e.g. the buffer an RCCL all-gather filled), asynchronous on `stream`:
 * the per-phase combine of a sliced multi-GPU prover never leaves HBM */
int h2mi_g1_fold_groups_dev(const void* d_points /* world*k*96 B */, size_t world, size_t k, void* d_out /* k*96 B */, h2mi_stream_t stream);
/* batch Jacobian -> affine (G1::batch_normalize, used by create_proof before transcript writes) */
int h2mi_g1_batch_normalize(const uint64_t* jac /* k*12 */, size_t k, uint64_t* affine_out /* k*8 */);
int h2mi_g1_batch_normalize_dev(const void* d_jac /* k*96 B */, size_t k, void* d_affine_out /* k*64 B */, h2mi_stream_t stream);

/* ---- NTT: halo2_proofs::arithmetic::best_fft(a, omega, log_n) ------------------------------------
 * Replaces best_fft for G = bn256::Fr: in-place DFT out[i] = sum_j a[j] *
 * omega^(i*j), natural order in and out.  The reference asserts a.len() == 1 << log_n; the caller
 * passes log_n and the buffer must hold 2^log_n elements.  log_n in [0, H2MI_MAX_LOG_N]. */
int h2mi_ntt_bn254_fr(uint64_t* a /* n*4 limbs, in place */, const uint64_t omega[4], uint32_t log_n);
/* EvaluationDomain helpers fused into the first / last pass:
 *   a[i] *= pre_scale_base^i  (distribute_powers_zeta with base = g_coset), then the DFT,
 *   then a[i] *= post_scale   (the ifft divisor n^-1).  Either pointer may be NULL. */
int h2mi_ntt_ext_bn254_fr(uint64_t* a, uint32_t log_n, const uint64_t omega[4],
                          const uint64_t* pre_scale_base_or_null, const uint64_t* post_scale_or_null);
/* device-resident form, asynchronous on `stream`; omega / scale constants are host pointers. */
int h2mi_ntt_bn254_fr_dev(void* d_a, uint32_t log_n, const uint64_t omega[4],
                          const uint64_t* pre_scale_base_or_null, const uint64_t* post_scale_or_null,
                          h2mi_stream_t stream);
/* out-of-place, zero-extending form: reads src_len <= 2^log_n elements from d_src (the rest count as zero),
 * writes the 2^log_n results to d_dst; d_src is left untouched and must not overlap d_dst.  This is
 * EvaluationDomain::coeff_to_extended without the clone and the zero padding (src_len = n, log_n =
 * extended_k, pre = g_coset), and lagrange_to_coeff on a column the prover still needs in Lagrange form. */
int h2mi_ntt_bn254_fr_oop_dev(const void* d_src, size_t src_len, void* d_dst, uint32_t log_n, const uint64_t omega[4],
                              const uint64_t* pre_scale_base_or_null, const uint64_t* post_scale_or_null,
                              h2mi_stream_t stream);
/* a[i] *= base^i on the device (EvaluationDomain::distribute_powers_zeta after an inverse coset NTT) */
int h2mi_fr_scale_powers_dev(void* d_a, size_t n, const uint64_t base[4], const uint64_t* post_scale_or_null,
                             h2mi_stream_t stream);

/* ---- opening-argument helpers on device-resident coefficient vectors ------------------
 * halo2_proofs::arithmetic::eval_polynomial(poly, point): out = sum_i poly[i] * point^i  (32 B at d_out) */
int h2mi_fr_eval_poly_dev(const void* d_poly, size_t n, const uint64_t point[4], void* d_out, h2mi_stream_t stream);
/* the same for `count` <= 24 polynomials of n coefficients at ONE point (create_proof evaluates every queried
 * column at x): one launch; result k at d_out + 32 k */
int h2mi_fr_eval_polys_dev(const void* const* d_polys, size_t count, size_t n, const uint64_t point[4], void* d_out, h2mi_stream_t stream);
/* every evaluation a proof writes, in ONE call: `ngroups` groups of polynomials (d_polys holds them group after group, group g has
 * group_counts[g] members) opened at points[g] (ngroups x 4 limbs); d_out receives the values in d_polys' order.  Up to 4 points and 24
 * polynomials share one power-table launch, one evaluation launch and one row sum (create_proof opens at x, omega x and omega^-(b+1) x:
 * nine launches through the per-point form); larger requests are served group by group. */
int h2mi_fr_eval_polys_multi_dev(const void* const* d_polys, const size_t* group_counts, const uint64_t* points, size_t ngroups, size_t n, void* d_out,
                                 h2mi_stream_t stream);
/* builds the cached power tables of `count` <= 32 bases (count x 4 limbs) at the size the helpers use for n-coefficient vectors, the
 * missing ones in one launch: with an opening argument's roots and their inverses known up front, the divisions that follow find their
 * tables instead of building them one launch at a time.  Purely a scheduling hint: results never depend on it. */
int h2mi_fr_powtab_prefetch_dev(const uint64_t* bases, size_t count, size_t n, h2mi_stream_t stream);
/* halo2_proofs::arithmetic::kate_division(a, b): quotient of a(X) by (X - b), n - 1 coefficients at d_out
 * (the remainder a(b) is dropped, as in the crate).  The caller passes b^-1 (one CPU inversion). */
int h2mi_fr_kate_division_dev(const void* d_poly, size_t n, const uint64_t b[4], const uint64_t b_inv[4], void* d_out,
                              h2mi_stream_t stream);
/* division by prod_{i < m} (X - roots[i]), m <= 4 distinct roots, of a polynomial they all vanish at — what SHPLONK's
 * div_by_vanishing does with one kate_division per point (poly/kzg/multiopen/shplonk/prover.rs [RECALL]) — in ONE round:
 * d_out[j] = sum_i weights[i] * (a / (X - roots[i]))[j] for j < n - 1, with weights[i] = 1 / prod_{k != i} (roots[i] - roots[k])
 * computed by the caller (partial fractions; the m quotients are independent, the chain of m dependent divisions is not).
 * roots, roots_inv, weights: m x 4 limbs, Montgomery.  d_out[n - 1] is left untouched; the top m - 1 written coefficients
 * come out as exact zeros.
 * Footprint: the m quotients are materialised side by side in the calling stream's scratch vector before they are summed —
 * m x (n + n / 512 + 1024) field elements (m = 4, n = 2^22: 0.5 GB; n = 2^25: 4 GB), one such scratch per stream that
 * issues divisions (SHPLONK's three lanes).  The scratch is grow-only; its first growth synchronises the device. */
int h2mi_fr_kate_division_multi_dev(const void* d_poly, size_t n, const uint64_t* roots, const uint64_t* roots_inv, const uint64_t* weights,
                                    size_t m, void* d_out, h2mi_stream_t stream);
/* out[i] = sum_k scalars[k] * polys[k][i], count <= 24 (the challenge-weighted sums of SHPLONK) */
int h2mi_fr_lincomb_dev(const void* const* d_polys, const uint64_t* scalars /* count*4 */, size_t count, size_t n, void* d_out,
                        h2mi_stream_t stream);

/* poly[i] += head[i] for i < count <= 16: the low-degree remainder terms R(X) / r = R(u) that SHPLONK subtracts
 * from a (linear combination of) opened polynomial(s) — poly/kzg/multiopen/shplonk/prover.rs
 * `quotient_contribution` / `linearisation_contribution`. */
int h2mi_fr_add_head_dev(void* d_poly, const uint64_t* head /* count*4 */, size_t count, h2mi_stream_t stream);
/* out[i] = a[i] * b[i] over n elements (in place allowed): the row values of a product of columns, e.g. the lookup input
 * q_lookup * a of halo2-base's single-advice-column range check */
int h2mi_fr_mul_dev(const void* d_a, const void* d_b, size_t n, void* d_out, h2mi_stream_t stream);
/* out[i] = value for i < n (Lagrange vectors such as l_active of keygen_pk) */
int h2mi_fr_fill_dev(void* d_out, size_t n, const uint64_t value[4], h2mi_stream_t stream);
/* Seeded stand-in for the prover's sweeps of `Scalar::random(rng)` (blinding rows; the vanishing argument's random
 * polynomial, plonk/vanishing/prover.rs `Argument::commit`; the reference passes OsRng, examples/standard_plonk.rs:48,
 * so its bytes are not reproducible — SURVEY.md 0).  Counter-based SplitMix64: element i has limbs
 * splitmix64(seed << 32 | 4 (start + i) + j), j = 0..3, top limb masked to 62 bits, reduced once; seed < 2^32. */
int h2mi_fr_random_dev(void* d_out, size_t n, uint64_t seed, uint64_t start, h2mi_stream_t stream);
/* the same sweep from a 256-bit key, for proofs that have to hide their witness: element i = Fr::from_u512 of ChaCha20 block start + i
 * (RFC 7539 block function; 64-bit block counter, 64-bit stream id `stream_id`: the layout of rand_chacha's ChaCha20Rng, whose eight
 * next_u64 per Fr::random are one block [RECALL halo2curves]); key: 32 bytes, host memory.  h2mi_prover_set_rng_key (h2mi_prover.h) uses it. */
int h2mi_fr_random_chacha_dev(void* d_out, size_t n, const uint8_t key[32], uint64_t stream_id, uint64_t start, h2mi_stream_t stream);

/* ---- quotient numerator for the reference's StandardPlonk circuit -------------------------
 * halo2_proofs plonk/evaluation.rs `evaluate_h` + vanishing division, specialised to the circuit of reference
 * src/circuits/standard_plonk.rs (one degree-3 gate over 3 advice + 5 fixed columns, 3 permutation sets of one
 * column).  All vectors are extended-domain evaluations (2^extended_k elements) resident in HBM. t_inv holds the
 * 2^(extended_k - k) values of (X^n - 1)^-1 on the coset.  The result (h on the extended coset, already divided)
 * goes to d_h_out; h2mi_ntt_bn254_fr_dev + h2mi_fr_scale_powers_dev bring it to coefficients. */
typedef struct {
  const void* advice[3];  /* a, b, c */
  const void* fixed[5];   /* q_a, q_b, q_c, q_ab, constant */
  const void* sigma[3];   /* permutation polynomials of a, b, c */
  const void* z[3];       /* permutation products */
  const void* l0;
  const void* l_last;
  const void* l_active;
} h2mi_standard_plonk_cosets;
int h2mi_plonk_evaluate_h_standard_dev(const h2mi_standard_plonk_cosets* cosets, uint32_t k, uint32_t extended_k,
                                       uint32_t blinding_factors, const uint64_t beta[4], const uint64_t gamma[4],
                                       const uint64_t y[4], const uint64_t delta[4], const uint64_t zeta[4],
                                       const uint64_t extended_omega[4], const uint64_t* t_inv /* 2^(extended_k-k) x 4 */,
                                       void* d_h_out, h2mi_stream_t stream);

/* the permutation argument's grand-product column for one chunk of m <= 64 columns (plonk/permutation/prover.rs,
 * the z vectors are produced where they are consumed): z[0] = start (one if NULL),
 *   z[i+1] = z[i] * prod_j (v_j[i] + beta delta^(c_j) omega^i + gamma) / prod_j (v_j[i] + beta sigma_j[i] + gamma),  i < usable_rows;
 * rows usable_rows+1 .. 2^k-1 of d_z (the blinding rows) are left untouched; z[usable_rows] also goes to d_last
 * (32 B, device) as the next chunk's start.  beta_delta_pows[j] = beta * delta^(index of column j in the argument).
 * One field inversion per call (prefix / suffix products of the denominators); a zero denominator makes the
 * result meaningless, where the crate would panic.  Asynchronous on `stream`. */
int h2mi_plonk_permutation_product_dev(const void* const* d_values, const void* const* d_sigmas, uint32_t m, uint32_t k,
                                       uint32_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4],
                                       const uint64_t* beta_delta_pows /* m*4 */, const uint64_t omega[4],
                                       const void* d_start_or_null, void* d_z, void* d_last_or_null, h2mi_stream_t stream);

/* every set of the permutation argument in one pass: m <= 64 (H2MI_FLEX_MAX_PERM) columns in argument order, chunked by chunk_len
 * (= cs.degree() - 2) into ceil(m / chunk_len) sets; d_z[s] receives rows 0 .. usable_rows of set s, each set
 * starting at the previous set's last value (plonk/permutation/prover.rs chains them through `last_z`); blinding
 * rows untouched.  One scan over the concatenated sets instead of one call (twelve launches) per set. */
int h2mi_plonk_permutation_products_dev(const void* const* d_values, const void* const* d_sigmas, uint32_t m, uint32_t chunk_len, uint32_t k,
                                        uint32_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4],
                                        const uint64_t* beta_delta_pows /* m*4 */, const uint64_t omega[4], void* const* d_z,
                                        h2mi_stream_t stream);
/* The same with the support of the copy constraints known (a keygen-time fact): `d_active` holds, sorted ascending, the
 * positions set * usable_rows + row (uint32) at which some column of the set has sigma != the identity permutation — the
 * only rows whose ratio can differ from one.  Numerators, denominators, the inversion and the scans then run over these
 * n_active positions only and every z row is filled from the prefix product of the positions before it: same values, work
 * proportional to the constrained cells plus one write per row (the halo2-lib examples constrain 10 .. 10^4 of 2^20 rows).
 * n_active = 0: every product is one. */
int h2mi_plonk_permutation_products_sparse_dev(const void* const* d_values, const void* const* d_sigmas, uint32_t m, uint32_t chunk_len, uint32_t k,
                                               uint32_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4],
                                               const uint64_t* beta_delta_pows, const uint64_t omega[4], const void* d_active, uint32_t n_active,
                                               void* const* d_z, h2mi_stream_t stream);

/* ---- lookup argument (plonk/lookup/prover.rs), single-expression lookups: the range check the reference's
 * RangeWithInstanceCircuitBuilder configures with LOOKUP_BITS (src/scaffold.rs:44-48,434-485; examples/range.rs:10-34).
 * commit_permuted's permute_expression_pair on the device: d_permuted_input[0 .. usable_rows) = the usable input rows
 * sorted (Fr's Ord: canonical integer order), d_permuted_table = the table rearranged against it (the input value where
 * a run starts, the unconsumed table values — ascending — on the repeated rows, last repeated row first); rows beyond
 * usable_rows (the blinding rows) are left untouched.  The fixed table is described by its distinct usable values in
 * ascending order (canonical little-endian integers AND the same values in Montgomery form) and their multiplicities,
 * prepared once at keygen: a proof needs no sort, only a counting sort against them.  not_in_table_out (mandatory; the
 * call synchronises on it) receives the number of inputs that are not table values — the crate fails the proof then, and
 * the permuted columns are meaningless: H2MI_EINVAL without it. */
int h2mi_plonk_lookup_permute_dev(const void* d_input, const void* d_table_sorted_canonical, const void* d_table_sorted_mont,
                                  const void* d_table_mult /* u32 x n_unique */, uint32_t n_unique, uint32_t k, uint32_t usable_rows,
                                  void* d_permuted_input, void* d_permuted_table, uint64_t* not_in_table_out, h2mi_stream_t stream);
/* extended-coset form of an instance column from the proving key's l_0 coset, without transforms: the column holds `count`
 * <= 16 public inputs on rows 0 .. count - 1 (what the scaffold's builders constrain: src/scaffold.rs:411, 480) and zeros
 * elsewhere, so its coset values are sum_r values[r] * l0_coset[(j - r 2^(extended_k - k)) mod 2^extended_k] — the same field
 * elements coeff_to_extended(lagrange_to_coeff(column)) yields.  values: count x 4 limbs, Montgomery. */
int h2mi_plonk_instance_coset_dev(const void* d_l0_coset, uint32_t k, uint32_t extended_k, const uint64_t* values, size_t count, void* d_out,
                                  h2mi_stream_t stream);
/* commit_product: z[0] = 1, z[i+1] = z[i] (a_i + beta)(t_i + gamma) / ((a'_i + beta)(s'_i + gamma)), i < usable_rows;
 * blinding rows untouched; one field inversion per call.  From 4096 usable rows the product runs over the rows whose ratio can
 * differ from one ((a_i, t_i) != (a'_i, s'_i): all but ~2^17 of 2^22 rows of a range check are skipped) when they are at most
 * a quarter of all rows; the call reads that count back, i.e. it waits for the work queued on `stream` before it (the library's
 * other streams keep running) */
int h2mi_plonk_lookup_product_dev(const void* d_input, const void* d_table, const void* d_permuted_input, const void* d_permuted_table, uint32_t k,
                                  uint32_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4], void* d_z, h2mi_stream_t stream);
/* evaluate_h + vanishing division for the halo2-lib constraint systems [halo2-base shapes restated from memory]: gate
 * q (a + a(wX) a(w^2 X) - a(w^3 X)), permutation argument over n_perm <= 4 columns in chunks of chunk_len (= cs.degree()
 * - 2 = 1 .. 3), and — has_lookup (the Range builder, extended domain 4n) — one lookup in `table` of either
 * `lookup_selector` * a (halo2-base with a single advice column: a complex selector on the looked-up cells' own rows,
 * lookup degree 5, chunk_len 3) or, when `lookup_selector` is NULL, of the dedicated column `lookup_advice` (degree 4,
 * chunk_len 2); without has_lookup (the Gate builder: degree 3, chunk_len 1, extended domain 2n) the lookup pointers are
 * unused.  All vectors are extended-coset evaluations. */
typedef struct {
  const void* a;               /* the gate's advice column */
  const void* lookup_advice;   /* the lookup input column (NULL when lookup_selector is given) */
  const void* lookup_selector; /* q_lookup: the lookup input is q_lookup * a (NULL: lookup_advice) */
  const void* q;               /* gate selector */
  const void* table;           /* fixed lookup table */
  const void* perm_value[4];   /* equality-enabled columns in argument order */
  const void* perm_sigma[4];
  const void* perm_z[4];       /* ceil(n_perm / chunk_len) grand products */
  const void* lookup_permuted_input;
  const void* lookup_permuted_table;
  const void* lookup_z;
  const void* l0;
  const void* l_last;
  const void* l_active;
  uint32_t n_perm;
  uint32_t chunk_len;          /* 1, 2 or 3 */
  uint32_t has_lookup;         /* 0: gate + permutation terms only */
} h2mi_range_cosets;
int h2mi_plonk_evaluate_h_range_dev(const h2mi_range_cosets* cosets, uint32_t k, uint32_t extended_k, uint32_t blinding_factors,
                                    const uint64_t beta[4], const uint64_t gamma[4], const uint64_t y[4], const uint64_t delta[4],
                                    const uint64_t zeta[4], const uint64_t extended_omega[4], const uint64_t* t_inv /* 2^(extended_k-k) x 4 */,
                                    void* d_h_out, h2mi_stream_t stream);

/* The general form of the halo2-base constraint systems (round 4; the column limits below from round 5): what
 * `builder.config(k, Some(minimum_rows))` (src/scaffold.rs:268) configures when the cells overflow ONE advice column — n_gates <=
 * H2MI_FLEX_MAX_GATES gate advice columns, each with its own selector and vertical gate; the permutation argument over n_perm <=
 * H2MI_FLEX_MAX_PERM columns (constants, the gate columns, the lookup-advice columns, the instance column) in chunks of chunk_len =
 * cs.degree() - 2; n_lookups <= H2MI_FLEX_MAX_LOOKUPS single-expression lookups
 * whose input is a lookup-advice column (lookup_input_b NULL) or the product of two columns (the single-column selector form).
 * Terms in evaluate_h's order: gates, permutation, lookups.  All vectors are extended-coset evaluations; t_inv as above.
 * Slower per point than the specialised entry above (every operand is converted to the multiplier's radix on load), and written
 * independently of its level bookkeeping: for a shape both accept the two must agree (tests/test_gpu_flex.py). */
#define H2MI_FLEX_MAX_GATES 32
#define H2MI_FLEX_MAX_PERM 64
#define H2MI_FLEX_MAX_LOOKUPS 8
typedef struct {
  uint32_t n_gates;
  const void* gate_a[H2MI_FLEX_MAX_GATES];
  const void* gate_q[H2MI_FLEX_MAX_GATES];
  uint32_t n_perm, chunk_len;
  const void* perm_value[H2MI_FLEX_MAX_PERM];
  const void* perm_sigma[H2MI_FLEX_MAX_PERM];
  const void* perm_z[H2MI_FLEX_MAX_PERM]; /* ceil(n_perm / chunk_len) grand products */
  uint32_t n_lookups;
  const void* lookup_input[H2MI_FLEX_MAX_LOOKUPS];
  const void* lookup_input_b[H2MI_FLEX_MAX_LOOKUPS]; /* NULL, or a second factor of the input expression */
  const void* lookup_table[H2MI_FLEX_MAX_LOOKUPS];
  const void* lookup_permuted_input[H2MI_FLEX_MAX_LOOKUPS];
  const void* lookup_permuted_table[H2MI_FLEX_MAX_LOOKUPS];
  const void* lookup_z[H2MI_FLEX_MAX_LOOKUPS];
  const void* l0;
  const void* l_last;
  const void* l_active;
} h2mi_flex_cosets;
int h2mi_plonk_evaluate_h_flex_dev(const h2mi_flex_cosets* cosets, uint32_t k, uint32_t extended_k, uint32_t blinding_factors,
                                   const uint64_t beta[4], const uint64_t gamma[4], const uint64_t y[4], const uint64_t delta[4],
                                   const uint64_t zeta[4], const uint64_t extended_omega[4], const uint64_t* t_inv /* 2^(extended_k-k) x 4 */,
                                   void* d_h_out, h2mi_stream_t stream);

/* ---- SRS generation helper: ParamsKZG::setup's g[i] = s_i * G   ------------------
 * d_scalars: n Fr (Montgomery).  d_out_affine: n G1Affine.  Fixed-base windowed multiplication of the
 * generator (1, 2) with on-device normalisation. */
int h2mi_g1_fixed_base_mul_dev(const void* d_scalars, size_t n, void* d_out_affine, h2mi_stream_t stream);
/* out[i] = base^i for i < n (the powers-of-s vector of ParamsKZG::setup), Montgomery in and out */
int h2mi_fr_powers_dev(void* d_out, size_t n, const uint64_t base[4], h2mi_stream_t stream);
/* halo2_proofs::arithmetic::best_fft for G = bn256::G1 — the group-valued transform ParamsKZG::setup runs over the monomial SRS to get
 * the Lagrange one (reference examples/standard_plonk.rs:29; `best_fft(&mut g_lagrange_projective, root.invert(), k)` then a scaling by
 * n^-1): out[i] = post_scale * sum_j omega^(i j) * in[j], natural order in and out.  Points are G1Affine on both sides (n x 64 B, (0, 0) =
 * identity); in place allowed.  omega: a 2^log_n-th root of unity; post_scale (Montgomery) may be NULL.  Every butterfly multiplies a
 * point by a 254-bit twiddle, so the cost is (n / 2) log n scalar multiplications (seconds at DEGREE 20 - 22, where the crate takes
 * minutes): keygen-time work.  Uses 144 B x n of scratch for the call and synchronises `stream` before returning.  log_n <= 26. */
int h2mi_fft_bn254_g1_dev(const void* d_affine_in, void* d_affine_out, uint32_t log_n, const uint64_t omega[4], const uint64_t* post_scale_or_null,
                          h2mi_stream_t stream);

/* ---- wire encodings: what create_proof writes to the transcript (reference call sites:
 * Blake2bWrite::init / finalize around create_proof, examples/standard_plonk.rs:40-50 and src/scaffold.rs:190-200,
 * 321-332; Blake2bRead::init for verify_proof, examples/standard_plonk.rs:56, src/scaffold.rs:221,352) and what
 * ParamsKZG::{write,read} keep on disk (the SRS the scaffold obtains through gen_srs, src/scaffold.rs:119,174,271,
 * cached under params/, .gitignore:17-18).  Conventions restated from halo2curves 0.3.x [RECALL, see
 * csrc/h2mi_serde.hip]: field elements as 32 little-endian canonical bytes (Fr::to_repr); G1Affine as 32
 * bytes = x with flags in byte 31 (0x40: y odd, 0x80: point at infinity).  field: 0 = Fq, 1 = Fr.
 * The *_from_* / decompress forms report the number of invalid encodings (>= modulus, not on the curve,
 * inconsistent flags) through invalid_out — such entries decode to zero / the identity — and synchronise. */
int h2mi_fe_to_repr_dev(int field, const void* d_in, size_t n, void* d_out32, h2mi_stream_t stream);
int h2mi_fe_from_repr_dev(int field, const void* d_in32, size_t n, void* d_out, uint64_t* invalid_out);
int h2mi_g1_compress_dev(const void* d_affine, size_t n, void* d_out32, h2mi_stream_t stream);
int h2mi_g1_decompress_dev(const void* d_in32, size_t n, void* d_affine_out, uint64_t* invalid_out);
/* host-pointer forms (synchronous): G1Affine::to_bytes / from_bytes over n points */
int h2mi_g1_compress(const uint64_t* affine /* n*8 */, size_t n, uint8_t* out32 /* n*32 */);
int h2mi_g1_decompress(const uint8_t* in32 /* n*32 */, size_t n, uint64_t* affine_out /* n*8 */, uint64_t* invalid_out);

/* ---- profiling: per-kernel device time measured with HIP events on the launching stream ---------- */
int h2mi_profile_enable(int on);         /* 1 = record events around every kernel launch */
/* restrict event recording to kernels whose name starts with `prefix` (NULL or "" = all kernels) */
int h2mi_profile_filter(const char* prefix);
int h2mi_profile_reset(void);
/* total milliseconds and launch count for kernels whose name starts with `prefix`; synchronises */
int h2mi_profile_query(const char* prefix, double* total_ms, uint64_t* launches);
/* every recorded launch in order, one text line each: "<kernel> <start ms after the first recorded launch> <duration ms>";
 * needed_out (may be NULL) receives the buffer size the full text needs; synchronises */
int h2mi_profile_dump(char* buf, size_t cap, size_t* needed_out);

#ifdef __cplusplus
}
#endif
#endif /* H2MI_H */
