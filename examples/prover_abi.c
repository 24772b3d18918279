/* prover_abi.c — the reference's examples/standard_plonk.rs (lines 25-50) from PLAIN C over the two C headers alone:
 * include/h2mi.h (SRS generation and registration) and include/h2mi_prover.h (keygen + the seven phase calls).  What a binding in any
 * language has to bring is all here, in ~300 lines: the circuit's cells (src/circuits/standard_plonk.rs:79-112), a Blake2b transcript
 * (halo2_proofs::transcript: personalisation "Halo2-Transcript", prefix bytes 0 / 1 / 2, Challenge255) and single-element Montgomery
 * arithmetic for the handful of scalars that cross it.  No vector work, no scheduling: that is the library's.
 *
 * Prints the verifying key's bytes and the proof; tests/test_gpu_prover_abi.py compares both with the committed golden proofs
 * (tests/golden/standard_plonk_proofs.json), i.e. with the oracle prover and the C++ / Python hosts.  vk.transcript_repr is the same
 * stand-in those hosts use (Blake2b "Halo2-Verify-Key" over k, the degree and the compressed commitments: README.md).
 *
 * Usage: prover_abi [k [srs_secret_hex [witness_hex [seed]]]]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/h2mi.h"
#include "../include/h2mi_prover.h"

__extension__ typedef unsigned __int128 u128; /* a GCC / Clang extension; the only one in this file */
typedef struct { uint64_t l[4]; } fe; /* 4 x 64-bit little-endian limbs */

/* ---- single-element arithmetic modulo r (scalars) and q (point coordinates): generic CIOS Montgomery product ---------------------- */
static const uint64_t R_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t R_INV = 0xc2e1f593efffffffULL;
static const fe R_ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
static const fe R_R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
static const uint64_t Q_MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t Q_INV = 0x87d20782e4866389ULL;

static fe mont_mul(const fe* a, const fe* b, const uint64_t mod[4], uint64_t inv) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * inv;
    c = (u128)m * mod[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * mod[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  int ge = t[4] != 0;
  if (!ge) {
    ge = 1;
    for (int i = 3; i >= 0; i--) {
      if (t[i] > mod[i]) break;
      if (t[i] < mod[i]) { ge = 0; break; }
    }
  }
  if (ge) {
    u128 bo = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)t[i] - mod[i] - (uint64_t)bo;
      t[i] = (uint64_t)d;
      bo = (d >> 64) & 1;
    }
  }
  fe r;
  memcpy(r.l, t, 32);
  return r;
}
static fe fr_mul(fe a, fe b) { return mont_mul(&a, &b, R_MOD, R_INV); }
static fe fr_add(fe a, fe b) {
  u128 c = 0;
  fe t;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.l[i] + b.l[i];
    t.l[i] = (uint64_t)c;
    c >>= 64;
  }
  int ge = 1;
  for (int i = 3; i >= 0; i--) {
    if (t.l[i] > R_MOD[i]) break;
    if (t.l[i] < R_MOD[i]) { ge = 0; break; }
  }
  if (ge) {
    u128 bo = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)t.l[i] - R_MOD[i] - (uint64_t)bo;
      t.l[i] = (uint64_t)d;
      bo = (d >> 64) & 1;
    }
  }
  return t;
}
static fe fr_neg(fe a) {
  fe t = {{0, 0, 0, 0}};
  if ((a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0) return t;
  u128 bo = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)R_MOD[i] - a.l[i] - (uint64_t)bo;
    t.l[i] = (uint64_t)d;
    bo = (d >> 64) & 1;
  }
  return t;
}
static fe fr_from_raw(fe raw) { return fr_mul(raw, R_R2); } /* any 256-bit integer -> its residue in Montgomery form */
static fe fr_from_u64(uint64_t v) { fe a = {{v, 0, 0, 0}}; return fr_from_raw(a); }
static fe fr_pow(fe a, const uint64_t e[4]) {
  fe r = R_ONE;
  for (int i = 255; i >= 0; i--) {
    r = fr_mul(r, r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = fr_mul(r, a);
  }
  return r;
}
static fe fr_inv(fe a) { const uint64_t e[4] = {R_MOD[0] - 2, R_MOD[1], R_MOD[2], R_MOD[3]}; return fr_pow(a, e); }
static fe fr_omega(uint32_t k) { /* ROOT_OF_UNITY = 7^((r - 1) / 2^28), squared 28 - k times (EvaluationDomain::new) */
  uint64_t m1[4] = {R_MOD[0] - 1, R_MOD[1], R_MOD[2], R_MOD[3]}, e[4];
  for (int i = 0; i < 4; i++) e[i] = (m1[i] >> 28) | (i < 3 ? m1[i + 1] << 36 : 0);
  fe w = fr_pow(fr_from_u64(7), e);
  for (uint32_t i = k; i < 28; i++) w = fr_mul(w, w);
  return w;
}
static void to_canonical(const fe* a, const uint64_t mod[4], uint64_t inv, uint8_t out[32]) { /* Montgomery -> 32 little-endian bytes */
  const fe one = {{1, 0, 0, 0}};
  fe c = mont_mul(a, &one, mod, inv);
  memcpy(out, c.l, 32);
}
static fe fr_from_wide(const uint8_t b[64]) { /* Fr::from_bytes_wide: 64 little-endian bytes reduced mod r */
  fe lo, hi;
  memcpy(lo.l, b, 32);
  memcpy(hi.l, b + 32, 32);
  return fr_add(fr_mul(lo, R_R2), fr_mul(fr_mul(hi, R_R2), R_R2));
}

/* ---- Blake2b-512 with a personalisation (RFC 7693), clonable by value ------------------------------------------------------------------- */
typedef struct { uint64_t h[8], t; uint8_t buf[128]; size_t fill; } blake2b;
static const uint64_t B2_IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                  0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static void b2_compress(blake2b* s, int last) {
  static const uint8_t SIGMA[12][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
  uint64_t m[16], v[16];
  memcpy(m, s->buf, 128);
  for (int i = 0; i < 8; i++) { v[i] = s->h[i]; v[i + 8] = B2_IV[i]; }
  v[12] ^= s->t;
  if (last) v[14] = ~v[14];
#define B2_G(a, b, c, d, x, y)                                   \
  v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32);      \
  v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24);      \
  v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16);      \
  v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63)
  for (int r = 0; r < 12; r++) {
    const uint8_t* g = SIGMA[r];
    B2_G(0, 4, 8, 12, m[g[0]], m[g[1]]);   B2_G(1, 5, 9, 13, m[g[2]], m[g[3]]);
    B2_G(2, 6, 10, 14, m[g[4]], m[g[5]]);  B2_G(3, 7, 11, 15, m[g[6]], m[g[7]]);
    B2_G(0, 5, 10, 15, m[g[8]], m[g[9]]);  B2_G(1, 6, 11, 12, m[g[10]], m[g[11]]);
    B2_G(2, 7, 8, 13, m[g[12]], m[g[13]]); B2_G(3, 4, 9, 14, m[g[14]], m[g[15]]);
  }
#undef B2_G
  for (int i = 0; i < 8; i++) s->h[i] ^= v[i] ^ v[i + 8];
}
static void b2_init(blake2b* s, const char* personal) {
  uint8_t param[64] = {0};
  param[0] = 64; param[2] = 1; param[3] = 1;
  memcpy(param + 48, personal, strlen(personal) < 16 ? strlen(personal) : 16);
  for (int i = 0; i < 8; i++) { uint64_t w; memcpy(&w, param + 8 * i, 8); s->h[i] = B2_IV[i] ^ w; }
  s->t = 0; s->fill = 0; memset(s->buf, 0, 128);
}
static void b2_update(blake2b* s, const void* data, size_t len) {
  const uint8_t* p = (const uint8_t*)data;
  while (len) {
    if (s->fill == 128) { s->t += 128; b2_compress(s, 0); s->fill = 0; }
    size_t take = 128 - s->fill < len ? 128 - s->fill : len;
    memcpy(s->buf + s->fill, p, take);
    s->fill += take; p += take; len -= take;
  }
}
static void b2_digest(const blake2b* s, uint8_t out[64]) { /* of everything absorbed so far; the state keeps absorbing */
  blake2b c = *s;
  c.t += c.fill;
  memset(c.buf + c.fill, 0, 128 - c.fill);
  b2_compress(&c, 1);
  memcpy(out, c.h, 64);
}

/* ---- the transcript (Blake2bWrite + Challenge255 for G1Affine) -------------------------------------------------------------------------- */
typedef struct { blake2b st; uint8_t proof[4096]; size_t len; } transcript;
static void die(const char* what, int rc) { fprintf(stderr, "prover_abi: %s: %s\n", what, h2mi_strerror(rc)); exit(2); }
static void check(int rc, const char* what) { if (rc != H2MI_OK) die(what, rc); }
static void tr_common_scalar(transcript* t, const fe* s) {
  uint8_t b[33];
  b[0] = 2;
  to_canonical(s, R_MOD, R_INV, b + 1);
  b2_update(&t->st, b, 33);
}
static void tr_write_scalar(transcript* t, const fe* s) {
  tr_common_scalar(t, s);
  to_canonical(s, R_MOD, R_INV, t->proof + t->len);
  t->len += 32;
}
static void tr_write_point(transcript* t, const uint64_t affine[8]) {
  if (!(affine[0] | affine[1] | affine[2] | affine[3] | affine[4] | affine[5] | affine[6] | affine[7])) {
    fprintf(stderr, "prover_abi: cannot write points at infinity to the transcript\n");
    exit(2);
  }
  uint8_t b[65];
  fe x, y;
  memcpy(x.l, affine, 32);
  memcpy(y.l, affine + 4, 32);
  b[0] = 1;
  to_canonical(&x, Q_MOD, Q_INV, b + 1);
  to_canonical(&y, Q_MOD, Q_INV, b + 33);
  b2_update(&t->st, b, 65);
  memcpy(t->proof + t->len, b + 1, 32); /* G1Affine::to_bytes: x, the parity of y in bit 6 of the last byte */
  t->proof[t->len + 31] |= (uint8_t)((b[33] & 1) << 6);
  t->len += 32;
}
static fe tr_squeeze(transcript* t) {
  const uint8_t zero = 0;
  uint8_t d[64];
  b2_update(&t->st, &zero, 1);
  b2_digest(&t->st, d);
  return fr_from_wide(d);
}
static void print_hex(const char* label, const uint8_t* b, size_t n) {
  printf("%s ", label);
  for (size_t i = 0; i < n; i++) printf("%02x", b[i]);
  printf("\n");
}
static fe fe_from_hex(const char* h) {
  if (h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) h += 2;
  char buf[65];
  size_t len = strlen(h);
  if (len > 64) { h += len - 64; len = 64; }
  memset(buf, '0', 64);
  memcpy(buf + 64 - len, h, len);
  buf[64] = 0;
  fe raw;
  for (int i = 0; i < 4; i++) {
    char w[17];
    memcpy(w, buf + 64 - 16 * (i + 1), 16);
    w[16] = 0;
    raw.l[i] = strtoull(w, NULL, 16);
  }
  return fr_from_raw(raw);
}

int main(int argc, char** argv) {
  const uint32_t k = argc > 1 ? (uint32_t)atoi(argv[1]) : 5; /* `let k = 5;` */
  const fe s = fe_from_hex(argc > 2 ? argv[2] : "5ec2e7");
  const fe x = fe_from_hex(argc > 3 ? argv[3] : "c0ffee");
  const uint64_t seed = argc > 4 ? strtoull(argv[4], NULL, 10) : 11;
  const size_t n = (size_t)1 << k;
  check(h2mi_init(0), "h2mi_init");

  /* let params = ParamsKZG::<Bn256>::setup(k, rng): g = s^i G, g_lagrange = L_i(s) G, registered as MSM base sets */
  void *pw = NULL, *d_g = NULL, *d_gl = NULL;
  uint64_t hg = 0, hgl = 0;
  const fe w_inv = fr_inv(fr_omega(k)), n_inv = fr_inv(fr_from_u64(n));
  check(h2mi_malloc(n * 32, &pw), "malloc");
  check(h2mi_malloc(n * 64, &d_g), "malloc");
  check(h2mi_malloc(n * 64, &d_gl), "malloc");
  check(h2mi_fr_powers_dev(pw, n, s.l, NULL), "powers of s");
  check(h2mi_g1_fixed_base_mul_dev(pw, n, d_g, NULL), "g");
  check(h2mi_ntt_bn254_fr_dev(pw, k, w_inv.l, NULL, n_inv.l, NULL), "lagrange scalars");
  check(h2mi_g1_fixed_base_mul_dev(pw, n, d_gl, NULL), "g_lagrange");
  check(h2mi_sync(), "sync");
  check(h2mi_bases_register_dev(d_g, n, &hg), "register g");
  check(h2mi_bases_register_dev(d_gl, n, &hgl), "register g_lagrange");

  /* StandardPlonkConfig::configure (src/circuits/standard_plonk.rs:29-48) as numbers */
  h2mi_constraint_system cs;
  memset(&cs, 0, sizeof(cs));
  cs.k = k; cs.n_advice = 3; cs.n_fixed = 5; cs.degree = 3; cs.blinding_factors = 5;
  cs.gates = H2MI_GATES_STANDARD_PLONK;
  cs.n_perm = 3; cs.n_advice_queries = 3; cs.n_fixed_queries = 5;
  for (uint32_t j = 0; j < 3; j++) {
    cs.perm_columns[j].kind = H2MI_COL_ADVICE; cs.perm_columns[j].index = j;
    cs.advice_queries[j].column = j;
  }
  for (uint32_t j = 0; j < 5; j++) cs.fixed_queries[j].column = j;
  /* what synthesize() assigns (:79-112): q_c = -1 and q_ab = 1 on rows 1, 2; constant = 72 on row 2; x copied into a and b of rows 1, 2 */
  const fe c72 = fr_from_u64(72), minus_one = fr_neg(R_ONE);
  const uint32_t rows12[2] = {1, 2}, row2[1] = {2};
  const fe qc[2] = {minus_one, minus_one}, qab[2] = {R_ONE, R_ONE};
  h2mi_column_cells fixed[5];
  memset(fixed, 0, sizeof(fixed));
  fixed[2].rows = rows12; fixed[2].values = qc[0].l;  fixed[2].count = 2;
  fixed[3].rows = rows12; fixed[3].values = qab[0].l; fixed[3].count = 2;
  fixed[4].rows = row2;   fixed[4].values = c72.l;    fixed[4].count = 1;
  const uint32_t copies[16] = {0, 1, 0, 0, 1, 1, 0, 0, 0, 2, 0, 0, 1, 2, 0, 0}; /* constrain_equal(new cell, (a, 0)) in call order */
  h2mi_pk_t pk = NULL;
  check(h2mi_prover_keygen(&cs, hgl, fixed, copies, 4, 0, &pk), "keygen");

  /* the verifying key: k, degree, the eight compressed commitments; transcript_repr = the hosts' stand-in (see the header comment) */
  uint64_t commitments[8 * 8];
  uint8_t vk_bytes[8 + 8 * 32];
  check(h2mi_prover_vk_commitments(pk, commitments, commitments + 5 * 8), "vk commitments");
  const uint32_t k32 = k, deg32 = cs.degree;
  memcpy(vk_bytes, &k32, 4);
  memcpy(vk_bytes + 4, &deg32, 4);
  check(h2mi_g1_compress(commitments, 8, vk_bytes + 8), "compress");
  blake2b vh;
  uint8_t d64[64];
  const uint64_t vk_len = sizeof(vk_bytes);
  b2_init(&vh, "Halo2-Verify-Key");
  b2_update(&vh, &vk_len, 8);
  b2_update(&vh, vk_bytes, sizeof(vk_bytes));
  b2_digest(&vh, d64);
  const fe transcript_repr = fr_from_wide(d64);

  /* create_proof: the witness, then seven calls with the transcript in between */
  const fe xx = fr_mul(x, x);
  const fe a_col[3] = {x, x, x}, b_col[2] = {x, x}, c_col[2] = {xx, fr_add(xx, c72)};
  h2mi_column_cells advice[3];
  memset(advice, 0, sizeof(advice));
  advice[0].values = a_col[0].l; advice[0].count = 3; /* rows 0 .. 2 */
  advice[1].rows = rows12; advice[1].values = b_col[0].l; advice[1].count = 2;
  advice[2].rows = rows12; advice[2].values = c_col[0].l; advice[2].count = 2;
  h2mi_prover_t prover = NULL;
  h2mi_prover_counts counts;
  check(h2mi_prover_create(pk, hg, hgl, 0, n, &prover), "prover_create");
  check(h2mi_prover_get_counts(prover, &counts), "counts");
  transcript tr;
  b2_init(&tr.st, "Halo2-Transcript");
  tr.len = 0;
  memset(tr.proof, 0, sizeof(tr.proof));
  tr_common_scalar(&tr, &transcript_repr); /* vk.hash_into */
  uint64_t pts[8 * 8];
  check(h2mi_prover_advice(prover, advice, NULL, 0, seed, pts), "advice");
  for (uint32_t i = 0; i < counts.advice; i++) tr_write_point(&tr, pts + 8 * i);
  (void)tr_squeeze(&tr); /* theta: drawn even without lookups */
  const fe beta = tr_squeeze(&tr), gamma = tr_squeeze(&tr);
  check(h2mi_prover_products(prover, beta.l, gamma.l, pts), "products");
  for (uint32_t i = 0; i < counts.products; i++) tr_write_point(&tr, pts + 8 * i);
  const fe y = tr_squeeze(&tr);
  check(h2mi_prover_quotient(prover, y.l, pts), "quotient");
  for (uint32_t i = 0; i < counts.quotient; i++) tr_write_point(&tr, pts + 8 * i);
  const fe xc = tr_squeeze(&tr);
  fe evals[64];
  if (counts.evaluations > 64) die("evaluations", H2MI_ERANGE);
  check(h2mi_prover_evaluations(prover, xc.l, evals[0].l), "evaluations");
  for (uint32_t i = 0; i < counts.evaluations; i++) tr_write_scalar(&tr, &evals[i]);
  const fe sy = tr_squeeze(&tr), sv = tr_squeeze(&tr); /* ProverSHPLONK: y, v */
  check(h2mi_prover_shplonk_quotient(prover, sy.l, sv.l, pts), "shplonk quotient");
  tr_write_point(&tr, pts);
  const fe su = tr_squeeze(&tr);
  check(h2mi_prover_shplonk_open(prover, su.l, pts), "shplonk open");
  tr_write_point(&tr, pts);

  print_hex("vk", vk_bytes, sizeof(vk_bytes));
  print_hex("proof", tr.proof, tr.len);
  printf("proof_bytes %zu\n", tr.len);
  check(h2mi_prover_destroy(prover), "prover_destroy");
  check(h2mi_prover_pk_release(pk), "pk_release");
  h2mi_bases_release(hg);
  h2mi_bases_release(hgl);
  h2mi_free(pw); h2mi_free(d_g); h2mi_free(d_gl);
  h2mi_shutdown();
  return 0;
}
