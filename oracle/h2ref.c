/* h2ref.c — plain-C CPU restatement of the reference stack's MSM and NTT algorithms.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the checker / the timed CPU baseline ("kind": "port").  The product (halo2-scaffold_amd/) never
 * links or loads it.
 *
 * PARITY UNPINNED: /root/reference contains no MSM/NTT/field code and no golden vectors; the
 * arithmetic lives in un-vendored crates (halo2_proofs @ PSE tag v2023_02_02, reference Cargo.toml:13;
 * halo2curves 0.3.x) and no Rust toolchain exists here.  The algorithms below restate, from the
 * published structure of those crates:
 *   h2ref_msm   halo2_proofs::arithmetic::best_multiexp / multiexp_serial: scalars to canonical bytes,
 *               window c = 3 (n < 32), else ceil(ln n); (256/c)+1 segments processed high to low with
 *               c doublings between segments; buckets None/Affine/Projective; running-sum reduction;
 *               the slice is split into one contiguous chunk per thread and the partial points are
 *               folded by addition.
 *   h2ref_ntt   halo2_proofs::arithmetic::best_fft: bit-reversal permutation, n/2 twiddles by
 *               repeated multiplication, log n rounds of radix-2 butterflies; threads as the crate's
 *               recursive_butterfly_arithmetic uses them: independent sub-transforms in parallel, the
 *               rounds that combine them serial per block (round 5; rounds 1-4 split every round across
 *               all threads, which lost to one thread at 2^16).
 *   field       halo2curves::bn256::{Fq, Fr}: 4 x 64-bit limbs, Montgomery form R = 2^256, CIOS
 *               multiplication, results always fully reduced.
 *   curve       halo2curves::bn256::{G1Affine, G1}: y^2 = x^3 + 3, Jacobian coordinates,
 *               dbl-2009-l / add-2007-bl / madd-2007-bl with complete special-case handling.
 * The reference call sites these are reached from: examples/standard_plonk.rs:29,33,34,41-49 and
 * src/scaffold.rs:119,132,135,174,191-199,271,284,287,322-331.
 * It is validated against oracle/bn254.py (independent big-integer formulas) in tests/test_oracle.py.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { fe x, y; } aff;       /* (0,0) = identity */
typedef struct { fe x, y, z; } jac;    /* z = 0 identity  */

typedef struct { uint64_t m[4]; uint64_t inv; fe one; fe r2; } fparams;

static const fparams FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0x87d20782e4866389ULL,
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}};
static const fparams FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0xc2e1f593efffffffULL,
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}};

/* ---- field ------------------------------------------------------------------------------------- */
static inline int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int geq_mod(const uint64_t t[4], const uint64_t m[4]) {
  for (int i = 3; i >= 0; i--) {
    if (t[i] > m[i]) return 1;
    if (t[i] < m[i]) return 0;
  }
  return 1;
}
static inline void sub_mod_raw(uint64_t t[4], const uint64_t m[4]) {
  u128 b = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)t[i] - m[i] - (uint64_t)b;
    t[i] = (uint64_t)d;
    b = (d >> 64) & 1;
  }
}
static inline void fe_add(fe* r, const fe* a, const fe* b, const fparams* P) {
  u128 c = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; i++) {
    c += (u128)a->l[i] + b->l[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  if (geq_mod(t, P->m)) sub_mod_raw(t, P->m); /* a + b < 2^255: no carry out */
  memcpy(r->l, t, 32);
}
static inline void fe_sub(fe* r, const fe* a, const fe* b, const fparams* P) {
  u128 bo = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)bo;
    t[i] = (uint64_t)d;
    bo = (d >> 64) & 1;
  }
  if (bo) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)t[i] + P->m[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  memcpy(r->l, t, 32);
}
static inline void fe_neg(fe* r, const fe* a, const fparams* P) {
  fe z = {{0, 0, 0, 0}};
  fe_sub(r, &z, a, P);
}
static inline void fe_mul(fe* r, const fe* a, const fe* b, const fparams* P) {
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
    uint64_t m = t[0] * P->inv;
    c = (u128)m * P->m[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * P->m[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || geq_mod(t, P->m)) sub_mod_raw(t, P->m);
  memcpy(r->l, t, 32);
}
static inline void fe_sqr(fe* r, const fe* a, const fparams* P) { fe_mul(r, a, a, P); }
static void fe_from_mont(fe* r, const fe* a, const fparams* P) {
  fe one = {{1, 0, 0, 0}};
  fe_mul(r, a, &one, P);
}
static void fe_pow(fe* r, const fe* a, const uint64_t e[4], const fparams* P) {
  fe acc = P->one;
  for (int i = 255; i >= 0; i--) {
    fe_sqr(&acc, &acc, P);
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, a, P);
  }
  *r = acc;
}
static void fe_inv(fe* r, const fe* a, const fparams* P) {
  uint64_t e[4] = {P->m[0] - 2, P->m[1], P->m[2], P->m[3]};
  fe_pow(r, a, e, P);
}

/* ---- curve ------------------------------------------------------------------------------------- */
static inline int aff_is_id(const aff* p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void jac_set_id(jac* r) {
  memset(r, 0, sizeof(*r));
  r->y = FQ.one;
}
static void jac_double(jac* r, const jac* p) {
  if (fe_is_zero(&p->z)) { *r = *p; return; }
  fe a, b, c, d, e, f, t, x3, y3, z3;
  fe_sqr(&a, &p->x, &FQ);
  fe_sqr(&b, &p->y, &FQ);
  fe_sqr(&c, &b, &FQ);
  fe_add(&d, &p->x, &b, &FQ);
  fe_sqr(&d, &d, &FQ);
  fe_sub(&d, &d, &a, &FQ);
  fe_sub(&d, &d, &c, &FQ);
  fe_add(&d, &d, &d, &FQ);
  fe_add(&e, &a, &a, &FQ);
  fe_add(&e, &e, &a, &FQ);
  fe_sqr(&f, &e, &FQ);
  fe_mul(&z3, &p->z, &p->y, &FQ);
  fe_add(&z3, &z3, &z3, &FQ);
  fe_add(&t, &d, &d, &FQ);
  fe_sub(&x3, &f, &t, &FQ);
  fe_add(&c, &c, &c, &FQ);
  fe_add(&c, &c, &c, &FQ);
  fe_add(&c, &c, &c, &FQ);
  fe_sub(&t, &d, &x3, &FQ);
  fe_mul(&y3, &e, &t, &FQ);
  fe_sub(&y3, &y3, &c, &FQ);
  r->x = x3; r->y = y3; r->z = z3;
}
static void jac_add(jac* r, const jac* p, const jac* q) {
  if (fe_is_zero(&p->z)) { *r = *q; return; }
  if (fe_is_zero(&q->z)) { *r = *p; return; }
  fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
  fe_sqr(&z1z1, &p->z, &FQ);
  fe_sqr(&z2z2, &q->z, &FQ);
  fe_mul(&u1, &p->x, &z2z2, &FQ);
  fe_mul(&u2, &q->x, &z1z1, &FQ);
  fe_mul(&s1, &p->y, &q->z, &FQ);
  fe_mul(&s1, &s1, &z2z2, &FQ);
  fe_mul(&s2, &q->y, &p->z, &FQ);
  fe_mul(&s2, &s2, &z1z1, &FQ);
  if (fe_eq(&u1, &u2)) {
    if (fe_eq(&s1, &s2)) { jac_double(r, p); return; }
    jac_set_id(r);
    return;
  }
  fe_sub(&h, &u2, &u1, &FQ);
  fe_add(&i, &h, &h, &FQ);
  fe_sqr(&i, &i, &FQ);
  fe_mul(&j, &h, &i, &FQ);
  fe_sub(&rr, &s2, &s1, &FQ);
  fe_add(&rr, &rr, &rr, &FQ);
  fe_mul(&v, &u1, &i, &FQ);
  fe_sqr(&x3, &rr, &FQ);
  fe_sub(&x3, &x3, &j, &FQ);
  fe_sub(&x3, &x3, &v, &FQ);
  fe_sub(&x3, &x3, &v, &FQ);
  fe_sub(&t, &v, &x3, &FQ);
  fe_mul(&y3, &rr, &t, &FQ);
  fe_mul(&t, &s1, &j, &FQ);
  fe_add(&t, &t, &t, &FQ);
  fe_sub(&y3, &y3, &t, &FQ);
  fe_add(&z3, &p->z, &q->z, &FQ);
  fe_sqr(&z3, &z3, &FQ);
  fe_sub(&z3, &z3, &z1z1, &FQ);
  fe_sub(&z3, &z3, &z2z2, &FQ);
  fe_mul(&z3, &z3, &h, &FQ);
  r->x = x3; r->y = y3; r->z = z3;
}
static void jac_from_aff(jac* r, const aff* p) {
  if (aff_is_id(p)) { jac_set_id(r); return; }
  r->x = p->x; r->y = p->y; r->z = FQ.one;
}
static void jac_add_mixed(jac* r, const jac* p, const aff* q) {
  if (aff_is_id(q)) { *r = *p; return; }
  if (fe_is_zero(&p->z)) { jac_from_aff(r, q); return; }
  fe z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
  fe_sqr(&z1z1, &p->z, &FQ);
  fe_mul(&u2, &q->x, &z1z1, &FQ);
  fe_mul(&s2, &q->y, &p->z, &FQ);
  fe_mul(&s2, &s2, &z1z1, &FQ);
  if (fe_eq(&p->x, &u2)) {
    if (fe_eq(&p->y, &s2)) { jac_double(r, p); return; }
    jac_set_id(r);
    return;
  }
  fe_sub(&h, &u2, &p->x, &FQ);
  fe_sqr(&hh, &h, &FQ);
  fe_add(&i, &hh, &hh, &FQ);
  fe_add(&i, &i, &i, &FQ);
  fe_mul(&j, &h, &i, &FQ);
  fe_sub(&rr, &s2, &p->y, &FQ);
  fe_add(&rr, &rr, &rr, &FQ);
  fe_mul(&v, &p->x, &i, &FQ);
  fe_sqr(&x3, &rr, &FQ);
  fe_sub(&x3, &x3, &j, &FQ);
  fe_sub(&x3, &x3, &v, &FQ);
  fe_sub(&x3, &x3, &v, &FQ);
  fe_sub(&t, &v, &x3, &FQ);
  fe_mul(&y3, &rr, &t, &FQ);
  fe_mul(&t, &p->y, &j, &FQ);
  fe_add(&t, &t, &t, &FQ);
  fe_sub(&y3, &y3, &t, &FQ);
  fe_add(&z3, &p->z, &h, &FQ);
  fe_sqr(&z3, &z3, &FQ);
  fe_sub(&z3, &z3, &z1z1, &FQ);
  fe_sub(&z3, &z3, &hh, &FQ);
  r->x = x3; r->y = y3; r->z = z3;
}
static void jac_to_aff(aff* r, const jac* p) {
  if (fe_is_zero(&p->z)) { memset(r, 0, sizeof(*r)); return; }
  fe zi, zi2, zi3;
  fe_inv(&zi, &p->z, &FQ);
  fe_sqr(&zi2, &zi, &FQ);
  fe_mul(&zi3, &zi2, &zi, &FQ);
  fe_mul(&r->x, &p->x, &zi2, &FQ);
  fe_mul(&r->y, &p->y, &zi3, &FQ);
}

/* ---- multiexp_serial (one thread's chunk) ------------------------------------------------------- */
typedef struct { int kind; aff a; jac j; } bucket; /* 0 None, 1 Affine, 2 Projective */

static size_t get_at(size_t segment, size_t c, const uint8_t bytes[32]) {
  size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
  if (skip_bytes >= 32) return 0;
  uint8_t v[8] = {0};
  for (size_t i = 0; i < 8 && skip_bytes + i < 32; i++) v[i] = bytes[skip_bytes + i];
  uint64_t tmp;
  memcpy(&tmp, v, 8);
  tmp >>= skip_bits - skip_bytes * 8;
  tmp %= ((uint64_t)1 << c);
  return (size_t)tmp;
}

static void multiexp_serial(const fe* coeffs, const aff* bases, size_t n, jac* acc) {
  uint8_t* repr = (uint8_t*)malloc(n * 32);
  for (size_t i = 0; i < n; i++) {
    fe c;
    fe_from_mont(&c, &coeffs[i], &FR); /* to_repr(): canonical little-endian bytes */
    memcpy(repr + 32 * i, c.l, 32);
  }
  size_t c;
  if (n < 4) c = 1;
  else if (n < 32) c = 3;
  else c = (size_t)ceil(log((double)n));
  size_t segments = 256 / c + 1;
  size_t nb = ((size_t)1 << c) - 1;
  bucket* buckets = (bucket*)malloc(nb * sizeof(bucket));
  for (size_t seg = segments; seg-- > 0;) {
    for (size_t k = 0; k < c; k++) jac_double(acc, acc);
    for (size_t b = 0; b < nb; b++) buckets[b].kind = 0;
    for (size_t i = 0; i < n; i++) {
      size_t d = get_at(seg, c, repr + 32 * i);
      if (d == 0) continue;
      bucket* B = &buckets[d - 1];
      if (B->kind == 0) { B->kind = 1; B->a = bases[i]; }
      else if (B->kind == 1) { jac t; jac_from_aff(&t, &B->a); jac_add_mixed(&B->j, &t, &bases[i]); B->kind = 2; }
      else { jac_add_mixed(&B->j, &B->j, &bases[i]); }
    }
    jac running;
    jac_set_id(&running);
    for (size_t b = nb; b-- > 0;) {
      if (buckets[b].kind == 1) jac_add_mixed(&running, &running, &buckets[b].a);
      else if (buckets[b].kind == 2) jac_add(&running, &running, &buckets[b].j);
      jac_add(acc, acc, &running);
    }
  }
  free(buckets);
  free(repr);
}

typedef struct { const fe* coeffs; const aff* bases; size_t n; jac acc; } msm_job;
static void* msm_worker(void* arg) {
  msm_job* j = (msm_job*)arg;
  jac_set_id(&j->acc);
  multiexp_serial(j->coeffs, j->bases, j->n, &j->acc);
  return NULL;
}

/* best_multiexp: scalars n*4 limbs (Montgomery Fr), bases n*8 limbs (affine Montgomery Fq) */
void h2ref_msm(const uint64_t* scalars, const uint64_t* bases, size_t n, int threads, uint64_t out_jac[12]) {
  const fe* co = (const fe*)scalars;
  const aff* ba = (const aff*)bases;
  jac total;
  jac_set_id(&total);
  if (threads < 1) threads = 1;
  if (n > (size_t)threads) {
    size_t chunk = n / (size_t)threads;
    size_t nchunks = (n + chunk - 1) / chunk;
    msm_job* jobs = (msm_job*)malloc(nchunks * sizeof(msm_job));
    pthread_t* th = (pthread_t*)malloc(nchunks * sizeof(pthread_t));
    for (size_t k = 0; k < nchunks; k++) {
      size_t lo = k * chunk, hi = lo + chunk > n ? n : lo + chunk;
      jobs[k].coeffs = co + lo; jobs[k].bases = ba + lo; jobs[k].n = hi - lo;
      pthread_create(&th[k], NULL, msm_worker, &jobs[k]);
    }
    for (size_t k = 0; k < nchunks; k++) {
      pthread_join(th[k], NULL);
      jac_add(&total, &total, &jobs[k].acc);
    }
    free(jobs);
    free(th);
  } else {
    multiexp_serial(co, ba, n, &total);
  }
  memcpy(out_jac, &total, 96);
}

/* ---- best_fft ---------------------------------------------------------------------------------- */
typedef struct { fe* a; const fe* tw; size_t n, chunk, twiddle_chunk, lo, hi; } fft_job;
static void* fft_round_worker(void* arg) {
  fft_job* J = (fft_job*)arg;
  size_t half = J->chunk / 2;
  for (size_t blk = J->lo; blk < J->hi; blk++) {
    fe* left = J->a + blk * J->chunk;
    fe* right = left + half;
    for (size_t i = 0; i < half; i++) {
      fe t;
      if (i == 0) t = right[0];
      else fe_mul(&t, &right[i], &J->tw[i * J->twiddle_chunk], &FR);
      fe u = left[i];
      fe_add(&left[i], &u, &t, &FR);
      fe_sub(&right[i], &u, &t, &FR);
    }
  }
  return NULL;
}
/* one thread's sub-transform: every round whose blocks fit inside region [region * sub, (region + 1) * sub) */
typedef struct { fe* a; const fe* tw; size_t n, sub, region; } fft_sub_job;
static void* fft_sub_worker(void* arg) {
  fft_sub_job* S = (fft_sub_job*)arg;
  for (size_t chunk = 2; chunk <= S->sub; chunk *= 2) {
    fft_job J = {S->a, S->tw, S->n, chunk, S->n / chunk, S->region * S->sub / chunk, (S->region + 1) * S->sub / chunk};
    fft_round_worker(&J);
  }
  return NULL;
}

static size_t bitreverse(size_t n, size_t l) {
  size_t r = 0;
  for (size_t i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; }
  return r;
}

void h2ref_ntt(uint64_t* data, const uint64_t omega[4], uint32_t log_n, int threads) {
  fe* a = (fe*)data;
  size_t n = (size_t)1 << log_n;
  if (threads < 1) threads = 1;
  for (size_t k = 0; k < n; k++) {
    size_t rk = bitreverse(k, log_n);
    if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; }
  }
  if (n < 2) return;
  fe* tw = (fe*)malloc((n / 2) * sizeof(fe));
  fe w = FR.one, om;
  memcpy(om.l, omega, 32);
  for (size_t i = 0; i < n / 2; i++) { tw[i] = w; fe_mul(&w, &w, &om, &FR); }
  /* The crate's split (best_fft with log_n > log_threads: recursive_butterfly_arithmetic under rayon::join): the two halves of a
   * block are independent sub-transforms run in parallel, the butterflies that combine them are ONE serial loop.  With
   * 2^d <= threads workers that is 2^d independent sub-transforms of n / 2^d points, one per thread, then d top rounds in which
   * round j has 2^(d-j) blocks, each combined serially by its own thread (the last round: one thread, n / 2 butterflies). */
  uint32_t d = 0;
  while (((size_t)2 << d) <= (size_t)threads) d++;
  if (log_n <= d) d = 0; /* "log_n <= log_threads": the serial iterative loop */
  /* rayon hands a sub-transform to a pooled worker for nothing; a pthread costs tens of microseconds: keep 2^12 points per thread */
  while (d > 0 && log_n < d + 12) d--;
  size_t sub = n >> d, regions = (size_t)1 << d;
  pthread_t* th = (pthread_t*)malloc(regions * sizeof(pthread_t));
  if (d == 0) {
    fft_sub_job S = {a, tw, n, n, 0};
    fft_sub_worker(&S);
  } else {
    fft_sub_job* subs = (fft_sub_job*)malloc(regions * sizeof(fft_sub_job));
    for (size_t g = 0; g < regions; g++) {
      fft_sub_job S = {a, tw, n, sub, g};
      subs[g] = S;
      pthread_create(&th[g], NULL, fft_sub_worker, &subs[g]);
    }
    for (size_t g = 0; g < regions; g++) pthread_join(th[g], NULL);
    free(subs);
    fft_job* jobs = (fft_job*)malloc(regions * sizeof(fft_job));
    for (size_t chunk = 2 * sub; chunk <= n; chunk *= 2) {
      size_t nblocks = n / chunk;
      for (size_t b = 0; b < nblocks; b++) {
        fft_job J = {a, tw, n, chunk, n / chunk, b, b + 1};
        jobs[b] = J;
        if (nblocks > 1) pthread_create(&th[b], NULL, fft_round_worker, &jobs[b]);
      }
      if (nblocks > 1) for (size_t b = 0; b < nblocks; b++) pthread_join(th[b], NULL);
      else fft_round_worker(&jobs[0]);
    }
    free(jobs);
  }
  free(th);
  free(tw);
}

/* ---- helpers for tests / fixtures --------------------------------------------------------------- */
/* elementwise field ops: field 0 = Fq, 1 = Fr ; op 0 mul, 1 add, 2 sub, 4 inv, 5 from_mont, 6 to_mont */
void h2ref_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  const fparams* P = field ? &FR : &FQ;
  for (size_t i = 0; i < n; i++) {
    const fe* x = (const fe*)(a + 4 * i);
    const fe* y = b ? (const fe*)(b + 4 * i) : x;
    fe r;
    switch (op) {
      case 0: fe_mul(&r, x, y, P); break;
      case 1: fe_add(&r, x, y, P); break;
      case 2: fe_sub(&r, x, y, P); break;
      case 4: fe_inv(&r, x, P); break;
      case 5: fe_from_mont(&r, x, P); break;
      case 6: fe_mul(&r, x, &P->r2, P); break;
      default: fe_neg(&r, x, P); break;
    }
    memcpy(out + 4 * i, r.l, 32);
  }
}

typedef struct { const fe* s; aff* out; size_t lo, hi; } gen_job;
static void* gen_worker(void* arg) {
  gen_job* J = (gen_job*)arg;
  aff g;
  g.x = FQ.one;
  fe_add(&g.y, &FQ.one, &FQ.one, &FQ);
  /* 4-bit fixed-window table of the generator */
  jac tab[16];
  jac_set_id(&tab[0]);
  for (int d = 1; d < 16; d++) jac_add_mixed(&tab[d], &tab[d - 1], &g);
  for (size_t i = J->lo; i < J->hi; i++) {
    fe c;
    fe_from_mont(&c, &J->s[i], &FR);
    jac acc;
    jac_set_id(&acc);
    for (int nib = 63; nib >= 0; nib--) {
      for (int k = 0; k < 4; k++) jac_double(&acc, &acc);
      unsigned d = (unsigned)((c.l[nib >> 4] >> ((nib & 15) * 4)) & 15);
      if (d) jac_add(&acc, &acc, &tab[d]);
    }
    jac_to_aff(&J->out[i], &acc);
  }
  return NULL;
}
/* out[i] = scalars[i] * G (affine) — SRS / test-base generation on the CPU */
void h2ref_g1_mul_gen(const uint64_t* scalars, size_t n, int threads, uint64_t* out_affine) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = (int)n;
  pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
  gen_job* jobs = (gen_job*)malloc((size_t)threads * sizeof(gen_job));
  for (int t = 0; t < threads; t++) {
    gen_job J = {(const fe*)scalars, (aff*)out_affine, n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads};
    jobs[t] = J;
    pthread_create(&th[t], NULL, gen_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}

/* Jacobian (12 limbs) -> affine (8 limbs), k points */
void h2ref_normalize(const uint64_t* jacp, size_t k, uint64_t* out_affine) {
  for (size_t i = 0; i < k; i++) jac_to_aff((aff*)(out_affine + 8 * i), (const jac*)(jacp + 12 * i));
}
/* sum of k Jacobian points */
void h2ref_sum(const uint64_t* jacp, size_t k, uint64_t out[12]) {
  jac t;
  jac_set_id(&t);
  for (size_t i = 0; i < k; i++) jac_add(&t, &t, (const jac*)(jacp + 12 * i));
  memcpy(out, &t, 96);
}

/* ---- Fr vector helpers for the large-size oracle prover (oracle/vec.py, oracle/fastflex.py) -------------------------
 * The Python-integer oracle prover (oracle/flex.py) takes hours at 2^20 rows; these are the same element-wise
 * definitions (halo2_proofs::arithmetic::{eval_polynomial, kate_division, parallelize} loops and the batch inversion of
 * plonk/permutation/prover.rs) over (n,4)-limb Montgomery arrays so that it can produce golden proofs at the sizes
 * BASELINE.json names.  Test infrastructure; validated element for element against oracle/bn254.py in tests/test_oracle_fast.py. */
typedef struct { int op; const fe* a; const fe* b; int b_scalar; fe* out; size_t lo, hi; } vec_job;
static void* vec_worker(void* arg) {
  vec_job* J = (vec_job*)arg;
  for (size_t i = J->lo; i < J->hi; i++) {
    const fe* y = J->b_scalar ? J->b : &J->b[i];
    fe r;
    switch (J->op) {
      case 0: fe_mul(&r, &J->a[i], y, &FR); break;
      case 1: fe_add(&r, &J->a[i], y, &FR); break;
      case 2: fe_sub(&r, &J->a[i], y, &FR); break;
      default: fe_sub(&r, y, &J->a[i], &FR); break; /* 3: b - a */
    }
    J->out[i] = r;
  }
  return NULL;
}
/* out[i] = a[i] (op) b[i] or b[0]; op 0 mul, 1 add, 2 sub, 3 reversed sub.  out may alias a or b. */
void h2ref_vec_binop(int op, const uint64_t* a, const uint64_t* b, int b_scalar, uint64_t* out, size_t n, int threads) {
  if (threads < 1) threads = 1;
  if (n < 4096) threads = 1;
  pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
  vec_job* jobs = (vec_job*)malloc((size_t)threads * sizeof(vec_job));
  for (int t = 0; t < threads; t++) {
    vec_job J = {op, (const fe*)a, (const fe*)b, b_scalar, (fe*)out, n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads};
    jobs[t] = J;
    if (threads == 1) vec_worker(&jobs[0]);
    else pthread_create(&th[t], NULL, vec_worker, &jobs[t]);
  }
  if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}
/* sum_i a[i] b[i] */
void h2ref_vec_dot(const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
  fe acc;
  memset(&acc, 0, sizeof acc);
  for (size_t i = 0; i < n; i++) {
    fe t;
    fe_mul(&t, (const fe*)(a + 4 * i), (const fe*)(b + 4 * i), &FR);
    fe_add(&acc, &acc, &t, &FR);
  }
  memcpy(out, acc.l, 32);
}
/* arithmetic::eval_polynomial: Horner from the top coefficient */
void h2ref_vec_horner(const uint64_t* a, size_t n, const uint64_t x[4], uint64_t out[4]) {
  fe acc, X;
  memset(&acc, 0, sizeof acc);
  memcpy(X.l, x, 32);
  for (size_t i = n; i-- > 0;) {
    fe_mul(&acc, &acc, &X, &FR);
    fe_add(&acc, &acc, (const fe*)(a + 4 * i), &FR);
  }
  memcpy(out, acc.l, 32);
}
/* arithmetic::kate_division: q[i-1] = a[i] + b q[i] from the top; out has n entries, out[n-1] = 0 */
void h2ref_vec_kate(const uint64_t* a, size_t n, const uint64_t b[4], uint64_t* out) {
  fe tmp, B;
  memset(&tmp, 0, sizeof tmp);
  memcpy(B.l, b, 32);
  if (n) memset(out + 4 * (n - 1), 0, 32);
  for (size_t i = n; i-- > 1;) {
    fe t;
    fe_mul(&t, &tmp, &B, &FR);
    fe_add(&tmp, &t, (const fe*)(a + 4 * i), &FR);
    memcpy(out + 4 * (i - 1), tmp.l, 32);
  }
}
/* out[i] = start * base^i */
void h2ref_vec_powers(const uint64_t base[4], const uint64_t start[4], size_t n, uint64_t* out) {
  fe w, B;
  memcpy(w.l, start, 32);
  memcpy(B.l, base, 32);
  for (size_t i = 0; i < n; i++) {
    memcpy(out + 4 * i, w.l, 32);
    fe_mul(&w, &w, &B, &FR);
  }
}
/* out[0] = start, out[i+1] = out[i] * a[i] for i < n - 1 … out has `n_out` entries, uses a[0 .. n_out-2] */
void h2ref_vec_running_product(const uint64_t* a, const uint64_t start[4], size_t n_out, uint64_t* out) {
  fe w;
  memcpy(w.l, start, 32);
  for (size_t i = 0; i < n_out; i++) {
    memcpy(out + 4 * i, w.l, 32);
    if (i + 1 < n_out) fe_mul(&w, &w, (const fe*)(a + 4 * i), &FR);
  }
}
/* Montgomery batch inversion (one field inversion), zero stays zero; out must not alias a */
void h2ref_vec_batch_inv(const uint64_t* a, uint64_t* out, size_t n) {
  fe acc = FR.one;
  for (size_t i = 0; i < n; i++) {
    memcpy(out + 4 * i, acc.l, 32);
    if (!fe_is_zero((const fe*)(a + 4 * i))) fe_mul(&acc, &acc, (const fe*)(a + 4 * i), &FR);
  }
  fe inv;
  fe_inv(&inv, &acc, &FR);
  for (size_t i = n; i-- > 0;) {
    const fe* x = (const fe*)(a + 4 * i);
    if (fe_is_zero(x)) { memset(out + 4 * i, 0, 32); continue; }
    fe t;
    fe_mul(&t, &inv, (const fe*)(out + 4 * i), &FR);
    fe_mul(&inv, &inv, x, &FR);
    memcpy(out + 4 * i, t.l, 32);
  }
}
