// libh2mi.so — the PLONK-specific vector kernels of create_proof: the permutation argument's and the lookup
// argument's grand products (plonk/permutation/prover.rs, plonk/lookup/prover.rs commit_product) and the quotient numerator
// (plonk/evaluation.rs evaluate_h + the division by X^n - 1) for the reference's StandardPlonk circuit
// (src/circuits/standard_plonk.rs:29-48) and the halo2-lib builders' constraint systems (src/scaffold.rs:379-485).
// Element-wise / scan work over HBM-resident vectors; reached from csrc/h2mi_prover.cpp.  Power tables and scratch vectors
// come from h2mi_ntt.hip (h2mi_fr_tables.h).  Split out of h2mi_ntt.hip in round 5; the kernels are unchanged.
#include <algorithm>
#include <vector>

#include "h2mi_fr_tables.h"
#include "scan.cuh"

namespace h2 {



// ---- quotient numerator of the reference's StandardPlonk circuit -------------------------
// halo2_proofs plonk/evaluation.rs `evaluate_h` specialised to src/circuits/standard_plonk.rs: one gate
// q_a a + q_b b + q_c c + q_ab a b + constant, three permutation sets of one column each; terms combined with
// powers of y; result already divided by X^n - 1 (its inverse on the coset repeats with period 2^(ext_k-k)).
// Element-wise over the extended domain; all vectors stay in HBM.
//
// Arithmetic: the lazy 29-bit-limb layer (f29.cuh) on the Montgomery-2^256 words as they lie in memory, WITHOUT
// converting them to its own radix.  f29_mul divides by 2^261, so the product of two memory-format values
// x 2^256, y 2^256 is x y 2^256 2^-5: every data-by-data product leaves one stray factor 2^-5.  Call a value "level L"
// when its limbs hold x 2^256 2^(-5 L): data and plain challenges are level 0, mul(level L1, level L2) is level
// L1 + L2 + 1, sums need equal levels.  The stray factors are paid by the HOST: h = sum_i y^(N-1-i) term_i is
// evaluated term by term (as many multiplications as Horner's rule), and the constant y^(N-1-i) for term i is
// handed over already multiplied by the power of 2^5 that brings this term back to level 0 — constants at negative
// levels (level -1 = the Montgomery-2^261 form: a multiplication by it keeps the level).  Terms that share a
// Lagrange factor (l_0, l_last, l_active) are summed before the one multiplication by it.  32 multiplications of
// ~210 instructions per point, against 36 of ~380 in the 32-bit-limb layer this kernel used before.
struct TInv {  // (X^n - 1)^-1 on the extended coset: 2^(extended_k - k) <= 16 distinct values
  fe v[16];
};
struct HConsts {
  fe beta_m1;            // beta at level -1: beta * sigma lands on level 0
  fe beta0, gamma0, one0;  // level 0: the plain Montgomery-2^256 words
  fe one_m2;             // one at level -2: brings a level-1 product (selector * advice) back to level 0
  fe cur[4];             // beta zeta DELTA^j at level 0: times X (level -1, straight from the power table) = level 0
  fe y[20];              // per-term powers of y at the level each use needs (layout: see the kernels / fill_* below)
  fe tinv[16];           // level -1
};
struct PlonkCosets {
  const fe* advice[3];
  const fe* fixed[5];
  const fe* sigma[3];
  const fe* z[3];
  const fe* l0;
  const fe* l_last;
  const fe* l_active;
};
__device__ __forceinline__ f29 hc(const fe& c) { return f29_unpack(c.v); }
__device__ __forceinline__ f29 hmul(const f29& a_lazy, const f29& b_norm) { return f29_mul<F9>(a_lazy, b_norm); }
// a - b + 2p, normalized (b normalized, value < 2p)
__device__ __forceinline__ f29 hsub(const f29& a, const f29& b) { return f29_normalize(f29_sub(a, b, F9::K2)); }
__device__ __forceinline__ void hstore(fe* dst, const f29& acc_lazy, const fe& tinv) {  // acc: lazy sum of <= 7 normalized values
  fe o;
  f29_pack(f29_reduce_canonical<F9>(f29_mul<F9>(f29_normalize(acc_lazy), hc(tinv))), o.v);
  fe_store(dst, o);
}

// y[] layout: [0] gate's q.a terms (level -2), [1] its q_ab a b term (-3), [2] its constant (-1), all times y^7;
// [3] [4] [5] the l_0 terms 1, 3, 4 (level -2); [6] the l_last term 2 (-3); [7] [8] [9] the permutation terms 5, 6, 7 (-3)
__global__ void __launch_bounds__(256) k_evaluate_h_standard_plonk(PlonkCosets c, uint32_t ext_k, uint32_t k, uint32_t last_rot, HConsts h,
                                                                    const fe* xlo, const fe* xhi, uint32_t xh, fe* out) {
  const uint32_t size = 1u << ext_k, rot = 1u << (ext_k - k);
  uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= size) return;
  const uint32_t r_next = (idx + rot) & (size - 1), r_last = (idx + size - last_rot * rot) & (size - 1);
  const f29 adv[3] = {load_unpack(&c.advice[0][idx]), load_unpack(&c.advice[1][idx]), load_unpack(&c.advice[2][idx])};
  // Sums of products at one level share ONE Montgomery reduction (f29_mul3 / f29_mul2, round 3): 33 products and 20 reductions per
  // point where every product had its own (33 + 33).  Operands of the shared forms are normalized (loads, constants, products,
  // hsub results); the one lazy operand — a three-term sum, limbs < 1.5 * 2^30 — sits where f29_mul2 allows it.
  // gate
  const f29 g1 = f29_mul3<F9>(load_unpack(&c.fixed[0][idx]), adv[0], load_unpack(&c.fixed[1][idx]), adv[1], load_unpack(&c.fixed[2][idx]), adv[2]);  // level 1
  const f29 g2 = hmul(hmul(load_unpack(&c.fixed[3][idx]), adv[0]), adv[1]);                          // level 2
  f29 acc = f29_mul3<F9>(g1, hc(h.y[0]), g2, hc(h.y[1]), load_unpack(&c.fixed[4][idx]), hc(h.y[2]));  // level 0
  const f29 zs[3] = {load_unpack(&c.z[0][idx]), load_unpack(&c.z[1][idx]), load_unpack(&c.z[2][idx])};
  const f29 one = hc(h.one0), gamma = hc(h.gamma0);
  // l_0 terms: (1 - z_0), (z_1 - z_0(w^last X)), (z_2 - z_1(w^last X))
  const f29 s0 = f29_mul3<F9>(hsub(one, zs[0]), hc(h.y[3]), hsub(zs[1], load_unpack(&c.z[0][r_last])), hc(h.y[4]),
                              hsub(zs[2], load_unpack(&c.z[1][r_last])), hc(h.y[5]));
  // l_last term: z_2^2 - z_2 = z_2 (z_2 - 1)
  const f29 tl = hmul(hmul(zs[2], hsub(zs[2], one)), hc(h.y[6]));
  // permutation terms: z_m(wX) (a_m + beta sigma_m + gamma) - z_m (a_m + beta DELTA^m X + gamma), X = zeta * extended_omega^idx
  const f29 X = pow2tab(xlo, xhi, xh, idx);
  auto term = [&](const f29& a_m, const f29& z_m, const fe* sigma, const fe* z_col, const fe& cur) {
    const f29 inner_l = f29_add(f29_add(a_m, hmul(load_unpack(&sigma[idx]), hc(h.beta_m1))), gamma);  // lazy: limbs < 1.5 * 2^30, value < 3.1 p
    const f29 inner_r = f29_normalize(f29_add(f29_add(a_m, hmul(X, hc(cur))), gamma));
    const f29 neg_r = f29_sub(f29_zero(), inner_r, F9::K4);                                           // 4p - inner_r: lazy, limbs < 2^30
    return f29_mul2<F9>(inner_l, load_unpack(&z_col[r_next]), neg_r, z_m);                            // left - right, level 1
  };
  const f29 d0 = term(adv[0], zs[0], c.sigma[0], c.z[0], h.cur[0]);
  const f29 d1 = term(adv[1], zs[1], c.sigma[1], c.z[1], h.cur[1]);
  const f29 d2 = term(adv[2], zs[2], c.sigma[2], c.z[2], h.cur[2]);
  const f29 sa = f29_mul3<F9>(d0, hc(h.y[7]), d1, hc(h.y[8]), d2, hc(h.y[9]));
  acc = f29_add(acc, f29_mul3<F9>(s0, load_unpack(&c.l0[idx]), tl, load_unpack(&c.l_last[idx]), sa, load_unpack(&c.l_active[idx])));
  hstore(&out[idx], acc, h.tinv[idx & (rot - 1)]);
}

// ---- permutation argument: the grand-product column z -------------------------------------
// create_proof builds, per chunk of columns, z[0] = start, z[i+1] = z[i] * prod_j (v_j[i] + beta delta^j omega^i
// + gamma) / prod_j (v_j[i] + beta sigma_j[i] + gamma) over the usable rows (plonk/permutation/prover.rs
// [RECALL], restated in oracle/plonk.py).  On the device: numerators / denominators per row, ONE field inversion
// for the whole column (prefix and suffix products of the denominators), then a prefix product of the ratios.
// Multiplicative scans over Fr in tiles of 1024 (local scan, scan of the tile totals, apply), forward or reverse.
// Everything between the numerator / denominator kernels and the final write lives in the lazy 29-bit-limb layer:
// the intermediate vectors hold canonical Montgomery-2^261 words, whose products stay in that domain
// (f29_mul(a 2^261, b 2^261) = a b 2^261); the columns themselves are Montgomery-2^256 and are converted once on the
// way in (one multiplication) and once on the way out (the mixed-domain product start * R, or a multiplication by 2^-5).
constexpr uint32_t MS_TILE = 1024;
__device__ __forceinline__ f29 ld261(const fe* p) { return f29_unpack(fe_load(p).v); }
__device__ __forceinline__ fe pack261(const f29& a_lt2p) {
  fe o;
  f29_pack(f29_reduce_canonical<F9>(a_lt2p), o.v);
  return o;
}
__device__ __forceinline__ fe one261() {
  fe o;
  f29_pack(f29_const<F9>(F9::ONE), o.v);
  return o;
}
__global__ void __launch_bounds__(256) k_mulscan_local(const fe* in, size_t n, int reverse, fe* local, fe* totals) {
  __shared__ fe tile[MS_TILE];
  __shared__ fe tprod[256];
  const uint32_t tid = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * MS_TILE;
  for (uint32_t r = 0; r < 4; r++) {  // logical position j; physical index n-1-j for a suffix scan
    size_t j = base + tid + 256 * r;
    tile[tid + 256 * r] = j < n ? fe_load(&in[reverse ? n - 1 - j : j]) : one261();
  }
  __syncthreads();
  const f29 p0 = f29_unpack(tile[4 * tid].v), p1 = f29_mul<F9>(p0, f29_unpack(tile[4 * tid + 1].v)), p2 = f29_mul<F9>(p1, f29_unpack(tile[4 * tid + 2].v)),
            p3 = f29_mul<F9>(p2, f29_unpack(tile[4 * tid + 3].v));
  tprod[tid] = pack261(p3);
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {  // inclusive scan of the 256 thread products (Hillis-Steele)
    fe v = one261();
    if (tid >= d) v = tprod[tid - d];
    __syncthreads();
    if (tid >= d) tprod[tid] = pack261(f29_mul<F9>(f29_unpack(tprod[tid].v), f29_unpack(v.v)));
    __syncthreads();
  }
  f29 left = f29_const<F9>(F9::ONE);
  if (tid) left = f29_unpack(tprod[tid - 1].v);
  __syncthreads();
  tile[4 * tid] = pack261(f29_mul<F9>(left, p0));
  tile[4 * tid + 1] = pack261(f29_mul<F9>(left, p1));
  tile[4 * tid + 2] = pack261(f29_mul<F9>(left, p2));
  tile[4 * tid + 3] = pack261(f29_mul<F9>(left, p3));
  __syncthreads();
  for (uint32_t r = 0; r < 4; r++) {
    size_t j = base + tid + 256 * r;
    if (j < n) fe_store(&local[reverse ? n - 1 - j : j], tile[tid + 256 * r]);
  }
  if (tid == 255) fe_store(&totals[blockIdx.x], tile[MS_TILE - 1]);
}
// exclusive scan of the tile totals: ONE workgroup of 1024 threads, each owning a run of consecutive totals (serial
// product), one Hillis-Steele scan over the 1024 run products, then the runs are walked again.  (Chunks of 256 with a
// running carry took 110 us for the 3072 totals of a three-column product: twelve dependent rounds of eight steps.)
__global__ void __launch_bounds__(1024) k_mulscan_offsets(const fe* totals, uint32_t nblocks, fe* offsets) {
  __shared__ fe tprod[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (nblocks + 1023) / 1024;
  const uint32_t lo = min(tid * per, nblocks), hi = min(lo + per, nblocks);
  f29 run = f29_const<F9>(F9::ONE);
  for (uint32_t b = lo; b < hi; b++) run = f29_mul<F9>(run, ld261(&totals[b]));
  tprod[tid] = pack261(run);
  __syncthreads();
  const uint32_t used = (nblocks + per - 1) / per;  // threads that own a run: the scan need not reach beyond them (six tiles
                                                     // of a sparse grand product: three rounds instead of ten, 0.2 ms -> 0.06)
  for (uint32_t d = 1; d < used; d <<= 1) {
    fe v = one261();
    if (tid >= d) v = tprod[tid - d];
    __syncthreads();
    if (tid >= d) tprod[tid] = pack261(f29_mul<F9>(f29_unpack(tprod[tid].v), f29_unpack(v.v)));
    __syncthreads();
  }
  f29 acc = f29_const<F9>(F9::ONE);
  if (tid) acc = f29_unpack(tprod[tid - 1].v);
  for (uint32_t b = lo; b < hi; b++) {
    fe_store(&offsets[b], pack261(f29_mul<F9>(acc, f29_const<F9>(F9::ONE))));
    acc = f29_mul<F9>(acc, ld261(&totals[b]));
  }
}
__global__ void __launch_bounds__(256) k_mulscan_apply(fe* local, const fe* offsets, size_t n, int reverse) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || j < MS_TILE) return;  // the first tile has offset one
  fe* p = &local[reverse ? n - 1 - j : j];
  fe_store(p, pack261(f29_mul<F9>(ld261(p), ld261(&offsets[j / MS_TILE]))));
}

// up to H2MI_FLEX_MAX_PERM columns by value: 3.1 KB of the 4 KB a launch's arguments may take
struct PermArgs {
  const fe* value[H2MI_FLEX_MAX_PERM];
  const fe* sigma[H2MI_FLEX_MAX_PERM];
  fe beta_delta[H2MI_FLEX_MAX_PERM];  // beta * delta^(column index), Mont256
  fe beta, gamma;
  uint32_t m;
};
// factors of one column at one row (Mont261, lazy sums below 6p): v + beta delta^j omega^i + gamma, v + beta sigma + gamma
__device__ __forceinline__ void perm_factors(const PermArgs& a, uint32_t j, size_t i, const f29& w, f29& numf, f29& denf) {
  const f29 v = f29_from_mont256<F9>(fe_load(&a.value[j][i]).v);
  const f29 g = f29_from_mont256<F9>(a.gamma.v);
  const f29 vg = f29_add(v, g);
  numf = f29_add(vg, f29_mul<F9>(f29_from_mont256<F9>(a.beta_delta[j].v), w));
  denf = f29_add(vg, f29_mul<F9>(f29_from_mont256<F9>(a.beta.v), f29_from_mont256<F9>(fe_load(&a.sigma[j][i]).v)));
}
// rows i < u: num = prod_j (v_j + beta delta^j omega^i + gamma), den = prod_j (v_j + beta sigma_j + gamma); one beyond
__global__ void __launch_bounds__(256) k_perm_numden(PermArgs a, size_t n, uint32_t u, const fe* wlo, const fe* whi, uint32_t wh, fe* num, fe* den) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f29 pn = f29_const<F9>(F9::ONE), pd = pn;
  if (i < u) {
    const f29 w = pow2tab(wlo, whi, wh, (uint32_t)i);  // omega^i (Mont261)
    for (uint32_t j = 0; j < a.m; j++) {
      f29 nf, df;
      perm_factors(a, j, i, w, nf, df);
      pn = f29_mul<F9>(nf, pn);
      pd = f29_mul<F9>(df, pd);
    }
  }
  fe_store(&num[i], pack261(pn));
  fe_store(&den[i], pack261(pd));
}
// the same for every set of a permutation argument at once: t = set * u + i over the concatenated usable rows
// (set = chunk of `chunk` consecutive columns); the running product then chains the sets by itself
__global__ void __launch_bounds__(256) k_perm_numden_sets(PermArgs a, uint32_t chunk, size_t total, uint32_t u, const fe* wlo, const fe* whi, uint32_t wh,
                                                           const uint32_t* active, fe* num, fe* den) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // element of the (compacted) concatenation
  if (e >= total) return;
  const size_t t = active ? active[e] : e;
  const uint32_t set = (uint32_t)(t / u), i = (uint32_t)(t - (size_t)set * u);
  const f29 w = pow2tab(wlo, whi, wh, i);  // omega^i (Mont261)
  f29 pn = f29_const<F9>(F9::ONE), pd = pn;
  for (uint32_t j = set * chunk; j < a.m && j < (set + 1) * chunk; j++) {
    f29 nf, df;
    perm_factors(a, j, i, w, nf, df);
    pn = f29_mul<F9>(nf, pn);
    pd = f29_mul<F9>(df, pd);
  }
  fe_store(&num[e], pack261(pn));
  fe_store(&den[e], pack261(pd));
}
struct ZOut {
  fe* z[H2MI_FLEX_MAX_PERM];
};
// z_set[i] = product of every ratio before (set, i) in the concatenated order: R[set * u + i - 1], one at the very start.
// With an `active` list (sorted positions whose ratio can differ from one, see h2mi_plonk_permutation_products_sparse_dev)
// R holds the prefix products over those positions only: z = R[c - 1], c = number of active positions before (set, i).
__global__ void __launch_bounds__(256) k_perm_write_sets(const fe* R, uint32_t u, ZOut out, const uint32_t* active, uint32_t n_active) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, set = blockIdx.y;
  if (i > u) return;
  size_t t = (size_t)set * u + i;
  if (active) {
    uint32_t lo = 0, hi = n_active;  // first index with active[idx] >= t
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (active[mid] < t) lo = mid + 1;
      else hi = mid;
    }
    t = lo;
  }
  fe o = fe_one<Fr>();
  if (t) {
    if (active) o = fe_load(&R[t - 1]);  // sparse form: R was brought to the memory format once per position (k_perm_to_mont256)
    else f29_to_mont256<F9>(ld261(&R[t - 1]), o.v);
  }
  fe_store(&out.z[set][i], o);
}
// the sparse form's prefix products, Montgomery-2^261 -> the columns' Montgomery-2^256, once per active position instead
// of once per row of every z column (the rows between two positions repeat one value)
__global__ void __launch_bounds__(256) k_perm_to_mont256(fe* R, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  fe o;
  f29_to_mont256<F9>(ld261(&R[i]), o.v);
  fe_store(&R[i], o);
}
// lookup argument's grand product (plonk/lookup/prover.rs commit_product), single-expression lookups:
// num_i = (a_i + beta)(t_i + gamma), den_i = (a'_i + beta)(s'_i + gamma), i < u
__global__ void __launch_bounds__(256) k_lookup_numden(const fe* input, const fe* table, const fe* pin, const fe* ptab, fe beta, fe gamma, uint32_t u,
                                                        fe* num, fe* den) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= u) return;
  const f29 b = f29_from_mont256<F9>(beta.v), g = f29_from_mont256<F9>(gamma.v);
  auto lift = [](const fe* p) { return f29_from_mont256<F9>(fe_load(p).v); };
  // the second operand of a product must be normalized: sums of two values below 2p are carried first
  fe_store(&num[i], pack261(f29_mul<F9>(f29_add(lift(&input[i]), b), f29_normalize(f29_add(lift(&table[i]), g)))));
  fe_store(&den[i], pack261(f29_mul<F9>(f29_add(lift(&pin[i]), b), f29_normalize(f29_add(lift(&ptab[i]), g)))));
}

// Sparse form of the lookup grand product (round 3).  ratio_i = (a_i + beta)(t_i + gamma) / ((a'_i + beta)(s'_i + gamma)) is
// exactly one wherever (a_i, t_i) = (a'_i, s'_i) — for a range check at DEGREE 22 on all but ~2^17 of 2^22 rows (input and
// permuted input are zero outside a handful of limbs, table and permuted table are zero outside 2^16 rows each) — so the
// product only moves at the other rows: they are flagged, compacted into a sorted position list (the scans of scan.cuh), the
// numerators / denominators / one inversion / prefix products run over that list, and k_perm_write_sets fills every row of z
// from the prefix product of the positions before it.  Same column bit for bit as the dense form (which multiplied 4 million
// ones: 3 multiplicative scans, 5.8 ms of the 74 ms range proof), chosen when at most a quarter of the rows are flagged.
__global__ void __launch_bounds__(256) k_lookup_flag(const fe* input, const fe* table, const fe* pin, const fe* ptab, uint32_t u, uint32_t padded,
                                                      uint32_t* flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  uint32_t f = 0;
  if (i < u) f = (fe_eq(fe_load(&input[i]), fe_load(&pin[i])) && fe_eq(fe_load(&table[i]), fe_load(&ptab[i]))) ? 0u : 1u;
  flag[i] = f;
}
__global__ void __launch_bounds__(256) k_lookup_compact(const uint32_t* flag, const uint32_t* pos, uint32_t u, uint32_t* active) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < u && flag[i]) active[pos[i]] = i;
}
__global__ void __launch_bounds__(256) k_lookup_numden_sparse(const fe* input, const fe* table, const fe* pin, const fe* ptab, fe beta, fe gamma,
                                                               const uint32_t* active, uint32_t n_active, fe* num, fe* den) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_active) return;
  const uint32_t i = active[e];
  const f29 b = f29_from_mont256<F9>(beta.v), g = f29_from_mont256<F9>(gamma.v);
  auto lift = [](const fe* p) { return f29_from_mont256<F9>(fe_load(p).v); };
  fe_store(&num[e], pack261(f29_mul<F9>(f29_add(lift(&input[i]), b), f29_normalize(f29_add(lift(&table[i]), g)))));
  fe_store(&den[e], pack261(f29_mul<F9>(f29_add(lift(&pin[i]), b), f29_normalize(f29_add(lift(&ptab[i]), g)))));
}

// ---- quotient numerator of the range-check constraint system (BASELINE config 3) -------------------
// What the reference's RangeWithInstanceCircuitBuilder produces (src/scaffold.rs:434-485) [halo2-base shape restated
// from memory]: one vertical gate q (a + a(wX) a(w^2 X) - a(w^3 X)) on the advice column, a permutation argument over
// n_perm <= 4 equality-enabled columns in chunks of one to three (constraint-system degree 3 .. 5), one single-expression
// lookup in the fixed table of either a lookup-advice column or selector * advice (halo2-base with one advice column).  Terms in evaluate_h's order (gates, permutation, lookups), Horner in y,
// divided by X^n - 1; extended domain 4n.
struct RangeCosets {
  const fe* a;
  const fe* la;   // lookup input: a dedicated lookup-advice column ...
  const fe* ql;   // ... or, when non-null, the selector of the single-advice-column form: input = ql * a
  const fe* q;
  const fe* table;
  const fe* perm_value[4];
  const fe* perm_sigma[4];
  const fe* perm_z[4];
  const fe* lk_input;
  const fe* lk_table;
  const fe* lk_z;
  const fe* l0;
  const fe* l_last;
  const fe* l_active;
  uint32_t n_perm, chunk, has_lookup;
};
// Same arithmetic scheme as k_evaluate_h_standard_plonk (levels, host-scaled powers of y).  h.y[i] belongs to term i
// in evaluate_h's order: 0 the gate (expression at level 2: y[0] at level -3); 1 (1 - z_first) l_0 and the other l_0
// terms (level-0 expressions: -2); 2 the l_last term (level 1: -3); 3 .. the chain terms (l_0); then one term per
// permutation set (expression level = columns in the set: -2 - columns); then the lookup's five: l_0 (-2), l_last (-3),
// the product rule (level 2: -4), l_0 (-2), the ordering rule (level 1: -3).
__device__ __forceinline__ void evaluate_h_range_body(const RangeCosets& c, uint32_t ext_k, uint32_t k, uint32_t last_rot, const HConsts& h,
                                                      const fe* xlo, const fe* xhi, uint32_t xh, fe* out) {
  const uint32_t size = 1u << ext_k, rot = 1u << (ext_k - k);
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= size) return;
  auto at = [&](int r) { return (idx + size + (uint32_t)(r * (int)rot)) & (size - 1); };
  const uint32_t r_next = at(1), r_prev = at(-1), r_last = at(-(int)last_rot);
  const f29 one = hc(h.one0), gamma = hc(h.gamma0), beta = hc(h.beta0);
  // Sums of products at one level share a Montgomery reduction (round 3, as in k_evaluate_h_standard_plonk): Dot gathers the terms of a
  // group (NORMALIZED values, each with its power of y) and multiplies them two at a time (f29_mul2); its branches depend on the
  // constraint system only (uniform).  Differences of two products (left - right of a permutation set, of the lookup) are one f29_mul2 with the second
  // product's first factor negated (4p - x).
  struct Dot {  // pairs (value, index of its y constant), multiplied two at a time: one pending value is all the state (three at a
                // time held four operands across the permutation products: 269 registers, one wavefront per SIMD)
    f29 a0, sum;
    uint32_t i0;
    bool pending;
    const HConsts& h;
    __device__ __forceinline__ Dot(const HConsts& hh) : i0(0), pending(false), h(hh) { sum = f29_zero(); }
    __device__ __forceinline__ void add(const f29& a, uint32_t yi) {
      if (!pending) { a0 = a; i0 = yi; pending = true; }
      else { sum = f29_normalize(f29_add(sum, f29_mul2<F9>(a0, hc(h.y[i0]), a, hc(h.y[yi])))); pending = false; }
    }
    __device__ __forceinline__ f29 result() {  // normalized
      if (pending) sum = f29_normalize(f29_add(sum, f29_mul<F9>(a0, hc(h.y[i0]))));
      pending = false;
      return sum;
    }
  };
  auto neg4 = [](const f29& x_norm) { return f29_sub(f29_zero(), x_norm, F9::K4); };  // 4p - x for normalized x < 4p - 2^232: limbs < 2^30
  // gate: q (a + a(wX) a(w^2 X) - a(w^3 X))
  f29 acc;
  {
    f29 wu = f29_mul2<F9>(load_unpack(&c.a[at(1)]), load_unpack(&c.a[at(2)]), hsub(load_unpack(&c.a[idx]), load_unpack(&c.a[at(3)])), one);  // level 1
    acc = hmul(hmul(wu, load_unpack(&c.q[idx])), hc(h.y[0]));
  }
  const uint32_t sets = (c.n_perm + c.chunk - 1) / c.chunk;
  const f29 z_first = load_unpack(&c.perm_z[0][idx]);
  const f29 z_lastset = load_unpack(&c.perm_z[sets - 1][idx]);
  Dot s0(h), sa(h);                                                                    // l_0 group, l_active group (level -1 after the y factors)
  s0.add(hsub(one, z_first), 1);
  const f29 pl = hmul(z_lastset, hsub(z_lastset, one));                                // l_last group: z (z - 1), level 1
  for (uint32_t s = 1; s < sets; s++) s0.add(hsub(load_unpack(&c.perm_z[s][idx]), load_unpack(&c.perm_z[s - 1][r_last])), 2 + s);
  const f29 X = pow2tab(xlo, xhi, xh, idx);
  const uint32_t p0 = 2 + sets;
  for (uint32_t s = 0; s < sets; s++) {
    f29 left = load_unpack(&c.perm_z[s][r_next]), right = load_unpack(&c.perm_z[s][idx]);
    const uint32_t j0 = c.chunk * s, j1 = min(c.n_perm, c.chunk * (s + 1));
    for (uint32_t j = j0; j + 1 < j1; j++) {
      const f29 val = load_unpack(&c.perm_value[j][idx]);
      left = hmul(f29_add(f29_add(val, hmul(load_unpack(&c.perm_sigma[j][idx]), hc(h.beta_m1))), gamma), left);
      right = hmul(f29_add(f29_add(val, hmul(X, hc(h.cur[j]))), gamma), right);
    }
    {  // the set's last column: both products' final factors in one two-product multiplication, left - right
      const uint32_t j = j1 - 1;
      const f29 val = load_unpack(&c.perm_value[j][idx]);
      const f29 inner_l = f29_add(f29_add(val, hmul(load_unpack(&c.perm_sigma[j][idx]), hc(h.beta_m1))), gamma);  // lazy, limbs < 1.5 * 2^30
      const f29 inner_r = f29_normalize(f29_add(f29_add(val, hmul(X, hc(h.cur[j]))), gamma));
      sa.add(f29_mul2<F9>(inner_l, left, neg4(inner_r), right), p0 + s);
    }
  }
  f29 sl;
  if (c.has_lookup) {
    const uint32_t lb = p0 + sets;
    const f29 ap = load_unpack(&c.lk_input[idx]), sp = load_unpack(&c.lk_table[idx]), zl = load_unpack(&c.lk_z[idx]);
    s0.add(hsub(one, zl), lb);
    sl = f29_mul2<F9>(pl, hc(h.y[2]), hmul(zl, hsub(zl, one)), hc(h.y[lb + 1]));
    {  // z(wX) (A' + beta) (S' + gamma) - z (A + beta) (S + gamma)
      const f29 li = hmul(f29_add(ap, beta), load_unpack(&c.lk_z[r_next]));
      const f29 a_in = c.ql ? hmul(hmul(load_unpack(&c.ql[idx]), load_unpack(&c.a[idx])), hc(h.one_m2)) : load_unpack(&c.la[idx]);
      const f29 tv = hmul(f29_add(a_in, beta), f29_normalize(f29_add(load_unpack(&c.table[idx]), gamma)));
      sa.add(f29_mul2<F9>(f29_add(sp, gamma), li, neg4(tv), zl), lb + 2);
    }
    const f29 a_minus_s = hsub(ap, sp);
    s0.add(a_minus_s, lb + 3);
    sa.add(hmul(a_minus_s, hsub(ap, load_unpack(&c.lk_input[r_prev]))), lb + 4);
  } else {
    sl = hmul(pl, hc(h.y[2]));
  }
  acc = f29_add(acc, f29_mul3<F9>(s0.result(), load_unpack(&c.l0[idx]), sl, load_unpack(&c.l_last[idx]), sa.result(), load_unpack(&c.l_active[idx])));
  hstore(&out[idx], acc, h.tinv[idx & (rot - 1)]);
}
// two register budgets of the same body (round 3): unconstrained it takes 169 VGPRs (two wavefronts per
// SIMD); held to 128 (four wavefronts) it spills 40 dwords to scratch.  Measured inside the range proof at DEGREE 22 on one
// box, alternating: 74.8 / 75.1 ms unconstrained, 75.1 / 75.3 ms at 128 VGPRs — the kernel streams 2^24 rows at ~40 % of HBM and
// the spills cost what the occupancy buys.  The unconstrained form is the product; the other exists in the -DH2MI_AB library (H2MI_EVALH_OCC=4).
__global__ void __launch_bounds__(256) k_evaluate_h_range(RangeCosets c, uint32_t ext_k, uint32_t k, uint32_t last_rot, HConsts h, const fe* xlo,
                                                           const fe* xhi, uint32_t xh, fe* out) {
  evaluate_h_range_body(c, ext_k, k, last_rot, h, xlo, xhi, xh, out);
}
#ifdef H2MI_AB  // the measured loser is compiled into the A/B library only (make ab), not into the product
__global__ void __launch_bounds__(256, 4) k_evaluate_h_range_occ4(RangeCosets c, uint32_t ext_k, uint32_t k, uint32_t last_rot, HConsts h,
                                                                  const fe* xlo, const fe* xhi, uint32_t xh, fe* out) {
  evaluate_h_range_body(c, ext_k, k, last_rot, h, xlo, xhi, xh, out);
}
#endif

// ---- quotient numerator, GENERAL form (round 4): what halo2-base configures when a circuit overflows one advice column --------
// `builder.config(k, Some(minimum_rows))` (src/scaffold.rs:268) then takes num_advice > 1 gate columns — each with its own
// vertical gate q_j (a_j + a_j(wX) a_j(w^2 X) - a_j(w^3 X)) — and, for the Range builder, num_lookup_advice lookup-advice
// columns with one lookup argument each [halo2-base shapes restated from memory].  The specialised kernels above track a stray
// 2^-5 per multiplication as "levels" so that operands can stay in the memory format; this one does not bother: every operand is
// brought to the multiplier's radix when it is loaded (one multiplication more per load), all arithmetic is in ONE domain, and h is
// Horner's rule in y over the terms in evaluate_h's order (gates, permutation, lookups) exactly as the oracle writes them
// (oracle/flex.py prove: _permutation_terms, _lookup_terms).  About half the speed of k_evaluate_h_range per point — a multi-column
// circuit is a small one by construction — and, being independent of the level bookkeeping, a cross-check of it: with one gate and
// the selector form of the lookup input both kernels must produce the same h (tests/test_gpu_flex.py).
struct FlexCosets {
  uint32_t n_gates, n_perm, chunk, n_lookups;
  const fe* gate_a[H2MI_FLEX_MAX_GATES];
  const fe* gate_q[H2MI_FLEX_MAX_GATES];
  const fe* perm_value[H2MI_FLEX_MAX_PERM];
  const fe* perm_sigma[H2MI_FLEX_MAX_PERM];
  const fe* perm_z[H2MI_FLEX_MAX_PERM];
  const fe* lk_in[H2MI_FLEX_MAX_LOOKUPS];
  const fe* lk_in_b[H2MI_FLEX_MAX_LOOKUPS];  // optional second factor of the input expression (selector * advice)
  const fe* lk_table[H2MI_FLEX_MAX_LOOKUPS];
  const fe* lk_pin[H2MI_FLEX_MAX_LOOKUPS];
  const fe* lk_ptab[H2MI_FLEX_MAX_LOOKUPS];
  const fe* lk_z[H2MI_FLEX_MAX_LOOKUPS];
  const fe* l0;
  const fe* l_last;
  const fe* l_active;
};
struct FlexConsts {
  fe beta, gamma, y, delta, zeta;  // Montgomery-2^256 words
  fe tinv[16];
};
__global__ void __launch_bounds__(256) k_evaluate_h_flex(FlexCosets c, uint32_t ext_k, uint32_t k, uint32_t last_rot, FlexConsts h, const fe* xlo,
                                                          const fe* xhi, uint32_t xh, fe* out) {
  const uint32_t size = 1u << ext_k, rot = 1u << (ext_k - k);
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= size) return;
  auto at = [&](int r) { return (idx + size + (uint32_t)(r * (int)rot)) & (size - 1); };
  const uint32_t r_next = at(1), r_prev = at(-1), r_last = at(-(int)last_rot);
  // one domain (Montgomery-2^261), every value normalized and below ~8 p between operations
  auto ld = [](const fe* p) { return f29_from_mont256<F9>(fe_load(p).v); };
  auto cst = [](const fe& v) { return f29_from_mont256<F9>(v.v); };
  auto mul = [](const f29& a, const f29& b) { return f29_mul<F9>(a, b); };                            // b normalized (every value here is)
  auto add = [](const f29& a, const f29& b) { return f29_normalize(f29_add(a, b)); };
  auto sub = [](const f29& a, const f29& b) { return f29_normalize(f29_sub(a, b, F9::K4)); };           // b < 4p - 2^232
  auto red = [](const f29& a) { return f29_mul<F9>(a, f29_const<F9>(F9::ONE)); };                       // back below 1.1 p (ONE = 2^261 mod p)
  const f29 one = f29_const<F9>(F9::ONE), beta = cst(h.beta), gamma = cst(h.gamma), y = cst(h.y), delta = cst(h.delta);
  f29 acc = f29_zero();
  auto horner = [&](const f29& term) { acc = add(mul(acc, y), term); };
  for (uint32_t g = 0; g < c.n_gates; g++) {
    const fe* a = c.gate_a[g];
    horner(mul(sub(add(ld(&a[idx]), mul(ld(&a[at(1)]), ld(&a[at(2)]))), ld(&a[at(3)])), ld(&c.gate_q[g][idx])));
  }
  const f29 l0 = ld(&c.l0[idx]), ll = ld(&c.l_last[idx]), lact = ld(&c.l_active[idx]);
  if (c.n_perm) {
    const uint32_t sets = (c.n_perm + c.chunk - 1) / c.chunk;
    const f29 z_first = ld(&c.perm_z[0][idx]), z_lastset = ld(&c.perm_z[sets - 1][idx]);
    horner(mul(sub(one, z_first), l0));
    horner(mul(red(sub(mul(z_lastset, z_lastset), z_lastset)), ll));
    for (uint32_t s = 1; s < sets; s++) horner(mul(sub(ld(&c.perm_z[s][idx]), ld(&c.perm_z[s - 1][r_last])), l0));
    f29 cur = mul(mul(pow2tab(xlo, xhi, xh, idx), cst(h.zeta)), beta);  // beta * X, X = zeta * extended_omega^idx
    for (uint32_t s = 0; s < sets; s++) {
      f29 left = ld(&c.perm_z[s][r_next]), right = ld(&c.perm_z[s][idx]);
      const uint32_t j0 = c.chunk * s, j1 = min(c.n_perm, c.chunk * (s + 1));
      for (uint32_t j = j0; j < j1; j++) {
        const f29 val = ld(&c.perm_value[j][idx]);
        left = mul(left, add(add(val, mul(beta, ld(&c.perm_sigma[j][idx]))), gamma));
        right = mul(right, add(add(val, cur), gamma));
        cur = mul(cur, delta);
      }
      horner(mul(sub(left, right), lact));
    }
  }
  for (uint32_t l = 0; l < c.n_lookups; l++) {
    f29 a_in = ld(&c.lk_in[l][idx]);
    if (c.lk_in_b[l]) a_in = mul(a_in, ld(&c.lk_in_b[l][idx]));
    const f29 t_in = ld(&c.lk_table[l][idx]), ap = ld(&c.lk_pin[l][idx]), ap_prev = ld(&c.lk_pin[l][r_prev]), sp = ld(&c.lk_ptab[l][idx]);
    const f29 lz = ld(&c.lk_z[l][idx]), lz_next = ld(&c.lk_z[l][r_next]);
    horner(mul(sub(one, lz), l0));
    horner(mul(red(sub(mul(lz, lz), lz)), ll));
    const f29 lhs = mul(mul(lz_next, add(ap, beta)), add(sp, gamma)), rhs = mul(mul(lz, add(a_in, beta)), add(t_in, gamma));
    horner(mul(sub(lhs, rhs), lact));
    const f29 d = sub(ap, sp);
    horner(mul(d, l0));
    horner(mul(red(mul(d, sub(ap, ap_prev))), lact));
  }
  fe o;
  f29_to_mont256<F9>(mul(acc, cst(h.tinv[idx & (rot - 1)])), o.v);
  fe_store(&out[idx], o);
}

// the one inversion on the critical path, by division steps on the 32-bit-limb layer (fe_inv_ds; round 4: the shift / subtract
// Euclid it replaces took 110 us for a lone wavefront).  in = x 2^261 read
// as a Montgomery-2^256 value is (32 x) 2^256; its inverse (x^-1 / 32) 2^256 times 2^10 is x^-1 2^261.
__global__ void k_fr_inv_one(const fe* in, fe* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  fe r = fe_inv_ds<Fr>(fe_load(in));
  for (int i = 0; i < 10; i++) r = fe_dbl<Fr>(r);
  fe_store(out, r);
}
// A handful of constrained cells (the reference's StandardPlonk: 8 positions, x^2 + 72: ~30): the whole sparse grand
// product in ONE workgroup.  R_i = prod_{j<=i} num_j / den_j = PN_i * S_(i+1) / S_0 with PN the prefix products of the
// numerators and S the suffix products of the denominators: the two scans run together (Hillis-Steele), and the ONE
// inversion (division steps, see k_fr_inv_one) is of S_0 — every lane of the first wavefront runs it on the same value, so
// its data-dependent branches are wavefront-uniform.  (Round 2 inverted each lane's own denominator: 64 different branch
// histories in one wavefront, 0.22 ms for 8 cells.)  The result is already in the columns' Montgomery-2^256 form for
// k_perm_write_sets.
constexpr uint32_t PERM_SMALL_MAX = 256;
__global__ void __launch_bounds__(PERM_SMALL_MAX) k_perm_sparse_small(PermArgs a, uint32_t chunk, uint32_t n_active, uint32_t u, const fe* wlo,
                                                                      const fe* whi, uint32_t wh, const uint32_t* active, fe* R) {
  __shared__ fe shn[PERM_SMALL_MAX], shd[PERM_SMALL_MAX + 1], sh_inv;
  const uint32_t tid = threadIdx.x;
  f29 pn = f29_const<F9>(F9::ONE), pd = pn;
  if (tid < n_active) {
    const uint32_t t = active[tid];
    const uint32_t set = t / u, i = t - set * u;
    const f29 w = pow2tab(wlo, whi, wh, i);
    for (uint32_t j = set * chunk; j < a.m && j < (set + 1) * chunk; j++) {
      f29 nf, df;
      perm_factors(a, j, i, w, nf, df);
      pn = f29_mul<F9>(nf, pn);
      pd = f29_mul<F9>(df, pd);
    }
  }
  shn[tid] = pack261(pn);
  shd[tid] = pack261(pd);
  if (tid == 0) shd[PERM_SMALL_MAX] = one261();
  __syncthreads();
  for (uint32_t d = 1; d < n_active; d <<= 1) {  // shn: inclusive prefix products; shd: inclusive suffix products
    fe vn = one261(), vd = one261();
    if (tid >= d) vn = shn[tid - d];
    if (tid + d < PERM_SMALL_MAX) vd = shd[tid + d];
    __syncthreads();
    if (tid >= d) shn[tid] = pack261(f29_mul<F9>(f29_unpack(shn[tid].v), f29_unpack(vn.v)));
    if (tid + d < PERM_SMALL_MAX) shd[tid] = pack261(f29_mul<F9>(f29_unpack(shd[tid].v), f29_unpack(vd.v)));
    __syncthreads();
  }
  if (tid < 64) {  // (S_0 2^261) read as Montgomery-2^256 is (32 S_0) 2^256; its inverse times 2^10 is S_0^-1 2^261
    fe inv = fe_inv_ds<Fr>(shd[0]);
    for (int q = 0; q < 10; q++) inv = fe_dbl<Fr>(inv);
    if (tid == 0) sh_inv = inv;
  }
  __syncthreads();
  if (tid < n_active) {
    f29 r = f29_mul<F9>(f29_unpack(shn[tid].v), f29_unpack(sh_inv.v));
    r = f29_mul<F9>(r, f29_unpack(shd[tid + 1].v));
    fe o;
    f29_to_mont256<F9>(r, o.v);
    fe_store(&R[tid], o);
  }
}
// ratio_i = num_i / den_i = num_i * P_(i-1) * S_(i+1) / P_(n-1)   (P, S: prefix / suffix products of den)
__global__ void __launch_bounds__(256) k_perm_ratio(const fe* num, const fe* P, const fe* S, const fe* inv_total, size_t n, fe* ratio) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f29 r = f29_mul<F9>(ld261(&num[i]), ld261(inv_total));
  if (i) r = f29_mul<F9>(r, ld261(&P[i - 1]));
  if (i + 1 < n) r = f29_mul<F9>(r, ld261(&S[i + 1]));
  fe_store(&ratio[i], pack261(r));
}
// z[0] = start, z[i+1] = start * R_i for i < u (R: inclusive prefix products of the ratios); rows beyond u untouched
__global__ void __launch_bounds__(256) k_perm_write(const fe* R, const fe* start_or_null, uint32_t u, fe* z, fe* last_or_null) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > u) return;
  const fe st = start_or_null ? fe_load(start_or_null) : fe_one<Fr>();
  fe v = st;
  if (i) {  // mixed-domain product: (start 2^256) (R 2^261) / 2^261 = start R 2^256
    f29_pack(f29_reduce_canonical<F9>(f29_mul<F9>(f29_unpack(st.v), ld261(&R[i - 1]))), v.v);
  }
  fe_store(&z[i], v);
  if (i == u && last_or_null) fe_store(last_or_null, v);
}

struct HostY {  // y^0 .. y^(count-1), Montgomery-2^256
  fe p[24];
  HostY(const fe& y, uint32_t count) {
    memcpy(p[0].v, h_canon(f29_const<F9>(F9::TO256)).v, 32);
    for (uint32_t i = 1; i < count && i < 24; i++) p[i] = h_mul256(p[i - 1], y);
  }
};
static void fill_common(HConsts& h, const fe& beta, const fe& gamma, const fe& delta, const fe& zeta, uint32_t n_cur, const uint64_t* t_inv,
                        uint32_t rot) {
  memset(&h, 0, sizeof(h));
  h.beta0 = beta;
  h.beta_m1 = h_level(beta, -1);
  h.gamma0 = gamma;
  h.one0 = h_canon(f29_const<F9>(F9::TO256));
  h.one_m2 = h_level(h.one0, -2);
  fe cur = h_mul256(beta, zeta);
  for (uint32_t j = 0; j < n_cur; j++) {
    h.cur[j] = cur;
    cur = h_mul256(cur, delta);
  }
  for (uint32_t i = 0; i < rot; i++) h.tinv[i] = h_level(host_fe(t_inv + 4 * i), -1);
}

}  // namespace h2

using namespace h2;

extern "C" {

// inclusive multiplicative scan of n elements, in place in `data` (forward prefix or reverse suffix)
static int mulscan(fe* data, size_t n, int reverse, fe* totals, fe* offsets, hipStream_t s) {
  const uint32_t nblocks = ceil_div_u32(n, MS_TILE);
  H2_LAUNCH("k_mulscan_local", k_mulscan_local, nblocks, 256, 0, s, (const fe*)data, n, reverse, data, totals);
  if (nblocks > 1) {
    H2_LAUNCH("k_mulscan_offsets", k_mulscan_offsets, 1, 1024, 0, s, (const fe*)totals, nblocks, offsets);
    H2_LAUNCH("k_mulscan_apply", k_mulscan_apply, ceil_div_u32(n, 256), 256, 0, s, data, (const fe*)offsets, n, reverse);
  }
  return H2MI_OK;
}

int h2mi_plonk_permutation_product_dev(const void* const* d_values, const void* const* d_sigmas, uint32_t m, uint32_t k, uint32_t usable_rows,
                                       const uint64_t beta[4], const uint64_t gamma[4], const uint64_t* beta_delta_pows, const uint64_t omega[4],
                                       const void* d_start_or_null, void* d_z, void* d_last_or_null, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_values || !d_sigmas || !beta || !gamma || !beta_delta_pows || !omega || !d_z || m == 0 || m > H2MI_FLEX_MAX_PERM) return H2MI_EINVAL;
  if (k == 0 || k > H2MI_MAX_LOG_N || usable_rows == 0 || usable_rows >= ((uint64_t)1 << k)) return H2MI_ERANGE;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  const size_t n = (size_t)1 << k;
  PermArgs a;
  memset(&a, 0, sizeof(a));
  a.m = m;
  a.beta = host_fe(beta);
  a.gamma = host_fe(gamma);
  for (uint32_t j = 0; j < m; j++) {
    if (!d_values[j] || !d_sigmas[j]) return H2MI_EINVAL;
    a.value[j] = (const fe*)d_values[j];
    a.sigma[j] = (const fe*)d_sigmas[j];
    a.beta_delta[j] = host_fe(beta_delta_pows + 4 * j);
  }
  PowTab pw;
  int rc = get_powtab(omega, k, s, &pw);
  if (rc) return rc;
  // scratch (the transforms' shared, stream-ordered buffer): num, P (prefix of den), S (suffix of den), tile
  // totals / offsets, the inverse of the total
  const uint32_t nblocks = ceil_div_u32(n, MS_TILE);
  rc = ensure_tmp(3 * n + 2 * (size_t)nblocks + 2, s);
  if (rc) return rc;
  fe* num = tmp_base();
  fe* P = num + n;
  fe* S = P + n;
  fe* totals = S + n;
  fe* offsets = totals + nblocks;
  fe* inv_total = offsets + nblocks;
  H2_LAUNCH("k_perm_numden", k_perm_numden, ceil_div_u32(n, 256), 256, 0, s, a, n, usable_rows, (const fe*)pw.lo, (const fe*)pw.hi, pw.h, num, P);
  H2_HIP(hipMemcpyAsync(S, P, n * 32, hipMemcpyDeviceToDevice, s));
  rc = mulscan(P, n, 0, totals, offsets, s);
  if (!rc) rc = mulscan(S, n, 1, totals, offsets, s);
  if (rc) return rc;
  H2_LAUNCH("k_fr_inv_one", k_fr_inv_one, 1, 64, 0, s, (const fe*)(P + (n - 1)), inv_total);
  // the ratios overwrite num; their prefix products then give z
  H2_LAUNCH("k_perm_ratio", k_perm_ratio, ceil_div_u32(n, 256), 256, 0, s, (const fe*)num, (const fe*)P, (const fe*)S, (const fe*)inv_total, n, num);
  rc = mulscan(num, n, 0, totals, offsets, s);
  if (rc) return rc;
  H2_LAUNCH("k_perm_write", k_perm_write, ceil_div_u32((uint64_t)usable_rows + 1, 256), 256, 0, s, (const fe*)num, (const fe*)d_start_or_null, usable_rows,
            (fe*)d_z, (fe*)d_last_or_null);
  return release_tmp(s);
}

static int perm_products(const void* const* d_values, const void* const* d_sigmas, uint32_t m, uint32_t chunk_len, uint32_t k, uint32_t usable_rows,
                         const uint64_t beta[4], const uint64_t gamma[4], const uint64_t* beta_delta_pows, const uint64_t omega[4],
                         const uint32_t* d_active, uint32_t n_active, void* const* d_z, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_values || !d_sigmas || !beta || !gamma || !beta_delta_pows || !omega || !d_z || m == 0 || m > H2MI_FLEX_MAX_PERM || chunk_len == 0) return H2MI_EINVAL;
  if (k == 0 || k > H2MI_MAX_LOG_N || usable_rows == 0 || usable_rows >= ((uint64_t)1 << k)) return H2MI_ERANGE;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  const uint32_t sets = (m + chunk_len - 1) / chunk_len;
  if (d_active && (uint64_t)sets * usable_rows >= ((uint64_t)1 << 32)) return H2MI_ERANGE;  // positions are 32-bit
  PermArgs a;
  memset(&a, 0, sizeof(a));
  a.m = m;
  a.beta = host_fe(beta);
  a.gamma = host_fe(gamma);
  for (uint32_t j = 0; j < m; j++) {
    if (!d_values[j] || !d_sigmas[j]) return H2MI_EINVAL;
    a.value[j] = (const fe*)d_values[j];
    a.sigma[j] = (const fe*)d_sigmas[j];
    a.beta_delta[j] = host_fe(beta_delta_pows + 4 * j);
  }
  ZOut zo;
  memset(&zo, 0, sizeof(zo));
  for (uint32_t q = 0; q < sets; q++) {
    if (!d_z[q]) return H2MI_EINVAL;
    zo.z[q] = (fe*)d_z[q];
  }
  const dim3 wgrid(ceil_div_u32((uint64_t)usable_rows + 1, 256), sets);
  if (d_active && n_active == 0) {  // no copy constraint touches a usable row: every product is one
    H2_LAUNCH("k_perm_write_sets", k_perm_write_sets, wgrid, 256, 0, s, (const fe*)nullptr, usable_rows, zo, d_active, 0u);
    return H2MI_OK;
  }
  PowTab pw;
  int rc = get_powtab(omega, k, s, &pw);
  if (rc) return rc;
  if (d_active && n_active <= PERM_SMALL_MAX) {
    rc = ensure_tmp(PERM_SMALL_MAX, s);
    if (rc) return rc;
    H2_LAUNCH("k_perm_sparse_small", k_perm_sparse_small, 1, PERM_SMALL_MAX, 0, s, a, chunk_len, n_active, usable_rows, (const fe*)pw.lo,
              (const fe*)pw.hi, pw.h, d_active, tmp_base());
    H2_LAUNCH("k_perm_write_sets", k_perm_write_sets, wgrid, 256, 0, s, (const fe*)tmp_base(), usable_rows, zo, d_active, n_active);
    return release_tmp(s);
  }
  const size_t total = d_active ? (size_t)n_active : (size_t)sets * usable_rows;  // elements the scans run over
  const uint32_t nblocks = ceil_div_u32(total, MS_TILE);
  rc = ensure_tmp(3 * total + 2 * (size_t)nblocks + 2, s);
  if (rc) return rc;
  fe* num = tmp_base();
  fe* P = num + total;
  fe* S = P + total;
  fe* totals = S + total;
  fe* offsets = totals + nblocks;
  fe* inv_total = offsets + nblocks;
  H2_LAUNCH("k_perm_numden_sets", k_perm_numden_sets, ceil_div_u32(total, 256), 256, 0, s, a, chunk_len, total, usable_rows, (const fe*)pw.lo, (const fe*)pw.hi,
            pw.h, d_active, num, P);
  H2_HIP(hipMemcpyAsync(S, P, total * 32, hipMemcpyDeviceToDevice, s));
  rc = mulscan(P, total, 0, totals, offsets, s);
  if (!rc) rc = mulscan(S, total, 1, totals, offsets, s);
  if (rc) return rc;
  H2_LAUNCH("k_fr_inv_one", k_fr_inv_one, 1, 64, 0, s, (const fe*)(P + (total - 1)), inv_total);
  H2_LAUNCH("k_perm_ratio", k_perm_ratio, ceil_div_u32(total, 256), 256, 0, s, (const fe*)num, (const fe*)P, (const fe*)S, (const fe*)inv_total, total, num);
  rc = mulscan(num, total, 0, totals, offsets, s);
  if (rc) return rc;
  if (d_active) H2_LAUNCH("k_perm_to_mont256", k_perm_to_mont256, ceil_div_u32(n_active, 256), 256, 0, s, num, n_active);
  H2_LAUNCH("k_perm_write_sets", k_perm_write_sets, wgrid, 256, 0, s, (const fe*)num, usable_rows, zo, d_active, n_active);
  return release_tmp(s);
}

int h2mi_plonk_permutation_products_dev(const void* const* d_values, const void* const* d_sigmas, uint32_t m, uint32_t chunk_len, uint32_t k,
                                        uint32_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t* beta_delta_pows,
                                        const uint64_t omega[4], void* const* d_z, h2mi_stream_t stream) {
  return perm_products(d_values, d_sigmas, m, chunk_len, k, usable_rows, beta, gamma, beta_delta_pows, omega, nullptr, 0, d_z, stream);
}

int h2mi_plonk_permutation_products_sparse_dev(const void* const* d_values, const void* const* d_sigmas, uint32_t m, uint32_t chunk_len, uint32_t k,
                                               uint32_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t* beta_delta_pows,
                                               const uint64_t omega[4], const void* d_active, uint32_t n_active, void* const* d_z, h2mi_stream_t stream) {
  if (!d_active) return H2MI_EINVAL;
  return perm_products(d_values, d_sigmas, m, chunk_len, k, usable_rows, beta, gamma, beta_delta_pows, omega, (const uint32_t*)d_active, n_active, d_z,
                       stream);
}

int h2mi_plonk_lookup_product_dev(const void* d_input, const void* d_table, const void* d_permuted_input, const void* d_permuted_table, uint32_t k,
                                  uint32_t usable_rows, const uint64_t beta[4], const uint64_t gamma[4], void* d_z, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_input || !d_table || !d_permuted_input || !d_permuted_table || !beta || !gamma || !d_z) return H2MI_EINVAL;
  if (k == 0 || k > H2MI_MAX_LOG_N || usable_rows == 0 || usable_rows >= ((uint64_t)1 << k)) return H2MI_ERANGE;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  const size_t u = usable_rows;
  const uint32_t uu = (usable_rows + 3u) & ~3u;  // scan length (a multiple of 4; the pad flags are zero)
  const uint32_t nseg = ceil_div_u32(uu, SCAN_SEG_BINS) + 1;
  const size_t words = 2 * ((size_t)uu + 8) + nseg + 8 + u / 4 + 8;  // flag, pos (+ total), segment sums, active positions
  const uint32_t nblocks_dense = ceil_div_u32(u, MS_TILE);
  int rc = ensure_tmp(3 * u + 2 * (size_t)nblocks_dense + 2 + (words * 4 + 31) / 32, s);
  if (rc) return rc;
  uint32_t* flag = reinterpret_cast<uint32_t*>(tmp_base() + 3 * u + 2 * (size_t)nblocks_dense + 2);
  uint32_t* pos = flag + uu + 8;
  uint32_t* segsum = pos + uu + 8;
  uint32_t* active = segsum + nseg + 8;
  const fe *in = (const fe*)d_input, *tab = (const fe*)d_table, *pin = (const fe*)d_permuted_input, *ptab = (const fe*)d_permuted_table;
  // the rows whose ratio can differ from one; the count decides between the sparse and the dense form (one 4-byte read)
  static const bool force_dense = ab_env("H2MI_LOOKUP_DENSE") != nullptr;  // A/B (-DH2MI_AB)
  uint32_t n_act = usable_rows;
  if (!force_dense && usable_rows >= 4096) {
    H2_LAUNCH("k_lookup_flag", k_lookup_flag, ceil_div_u32(uu, 256), 256, 0, s, in, tab, pin, ptab, usable_rows, uu, flag);
    const uint32_t segs = ceil_div_u32(uu, SCAN_SEG_BINS);
    if (segs > 1) H2_LAUNCH("k_scan_segsum", k_scan_segsum<SCAN_SEG_BINS>, segs, 1024, 0, s, (const uint32_t*)flag, uu, segsum);
    H2_LAUNCH("k_scan_seg_lookup", k_scan_seg<SCAN_SEG_BINS>, dim3(segs, 1), 1024, 0, s, (const uint32_t*)flag, pos, (const uint32_t*)nullptr, (uint32_t*)nullptr, uu,
              (const uint32_t*)segsum);
    H2_HIP(hipMemcpyAsync(&n_act, pos + uu, 4, hipMemcpyDeviceToHost, s));
    H2_HIP(hipStreamSynchronize(s));
  }
  const bool sparse = (size_t)n_act * 4 <= u && !force_dense && usable_rows >= 4096;
  const size_t total = sparse ? n_act : u;
  const uint32_t nblocks = ceil_div_u32(std::max<size_t>(total, 1), MS_TILE);
  fe* num = tmp_base();
  fe* P = num + u;
  fe* S = P + u;
  fe* totals = S + u;
  fe* offsets = totals + nblocks_dense;
  fe* inv_total = offsets + nblocks_dense;
  ZOut zo;
  memset(&zo, 0, sizeof(zo));
  zo.z[0] = (fe*)d_z;
  if (sparse && n_act == 0) {  // every ratio is one: z = 1 on rows 0 .. u (an empty position list)
    H2_LAUNCH("k_perm_write_sets", k_perm_write_sets, dim3(ceil_div_u32((uint64_t)usable_rows + 1, 256), 1), 256, 0, s, (const fe*)num, usable_rows, zo,
              (const uint32_t*)active, 0u);
    return release_tmp(s);
  }
  if (sparse) {
    H2_LAUNCH("k_lookup_compact", k_lookup_compact, ceil_div_u32(usable_rows, 256), 256, 0, s, (const uint32_t*)flag, (const uint32_t*)pos, usable_rows, active);
    H2_LAUNCH("k_lookup_numden", k_lookup_numden_sparse, ceil_div_u32(n_act, 256), 256, 0, s, in, tab, pin, ptab, host_fe(beta), host_fe(gamma),
              (const uint32_t*)active, n_act, num, P);
  } else {
    H2_LAUNCH("k_lookup_numden", k_lookup_numden, ceil_div_u32(total, 256), 256, 0, s, in, tab, pin, ptab, host_fe(beta), host_fe(gamma), usable_rows, num, P);
  }
  (void)nblocks;
  H2_HIP(hipMemcpyAsync(S, P, total * 32, hipMemcpyDeviceToDevice, s));
  rc = mulscan(P, total, 0, totals, offsets, s);
  if (!rc) rc = mulscan(S, total, 1, totals, offsets, s);
  if (rc) return rc;
  H2_LAUNCH("k_fr_inv_one", k_fr_inv_one, 1, 64, 0, s, (const fe*)(P + (total - 1)), inv_total);
  H2_LAUNCH("k_perm_ratio", k_perm_ratio, ceil_div_u32(total, 256), 256, 0, s, (const fe*)num, (const fe*)P, (const fe*)S, (const fe*)inv_total, total, num);
  rc = mulscan(num, total, 0, totals, offsets, s);
  if (rc) return rc;
  if (sparse) {
    H2_LAUNCH("k_perm_to_mont256", k_perm_to_mont256, ceil_div_u32(n_act, 256), 256, 0, s, num, n_act);
    H2_LAUNCH("k_perm_write_sets", k_perm_write_sets, dim3(ceil_div_u32((uint64_t)usable_rows + 1, 256), 1), 256, 0, s, (const fe*)num, usable_rows, zo,
              (const uint32_t*)active, n_act);
  } else {
    H2_LAUNCH("k_perm_write_sets", k_perm_write_sets, dim3(ceil_div_u32((uint64_t)usable_rows + 1, 256), 1), 256, 0, s, (const fe*)num, usable_rows, zo,
              (const uint32_t*)nullptr, 0u);
  }
  return release_tmp(s);
}

int h2mi_plonk_evaluate_h_range_dev(const h2mi_range_cosets* c, uint32_t k, uint32_t extended_k, uint32_t blinding_factors, const uint64_t beta[4],
                                    const uint64_t gamma[4], const uint64_t y[4], const uint64_t delta[4], const uint64_t zeta[4],
                                    const uint64_t extended_omega[4], const uint64_t* t_inv, void* d_h_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!c || !beta || !gamma || !y || !delta || !zeta || !extended_omega || !t_inv || !d_h_out) return H2MI_EINVAL;
  if (extended_k < k || extended_k - k > 4 || extended_k > H2MI_MAX_LOG_N) return H2MI_ERANGE;
  if (c->n_perm == 0 || c->n_perm > 4 || c->chunk_len == 0 || c->chunk_len > 3) return H2MI_EINVAL;
  RangeCosets rc_;
  memset(&rc_, 0, sizeof(rc_));
  rc_.a = (const fe*)c->a; rc_.la = (const fe*)c->lookup_advice; rc_.ql = (const fe*)c->lookup_selector; rc_.q = (const fe*)c->q; rc_.table = (const fe*)c->table;
  rc_.lk_input = (const fe*)c->lookup_permuted_input; rc_.lk_table = (const fe*)c->lookup_permuted_table; rc_.lk_z = (const fe*)c->lookup_z;
  rc_.l0 = (const fe*)c->l0; rc_.l_last = (const fe*)c->l_last; rc_.l_active = (const fe*)c->l_active;
  rc_.n_perm = c->n_perm;
  rc_.chunk = c->chunk_len;
  rc_.has_lookup = c->has_lookup ? 1u : 0u;
  if (!rc_.a || !rc_.q || !rc_.l0 || !rc_.l_last || !rc_.l_active) return H2MI_EINVAL;
  if (rc_.has_lookup && ((!rc_.la && !rc_.ql) || !rc_.table || !rc_.lk_input || !rc_.lk_table || !rc_.lk_z)) return H2MI_EINVAL;
  for (uint32_t j = 0; j < c->n_perm; j++) {
    rc_.perm_value[j] = (const fe*)c->perm_value[j];
    rc_.perm_sigma[j] = (const fe*)c->perm_sigma[j];
    if (!rc_.perm_value[j] || !rc_.perm_sigma[j]) return H2MI_EINVAL;
  }
  for (uint32_t q = 0; q < (c->n_perm + c->chunk_len - 1) / c->chunk_len; q++) {
    rc_.perm_z[q] = (const fe*)c->perm_z[q];
    if (!rc_.perm_z[q]) return H2MI_EINVAL;
  }
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  PowTab px;
  int rc = get_powtab(extended_omega, extended_k, s, &px);
  if (rc) return rc;
  const uint32_t rot = 1u << (extended_k - k);
  HConsts hcst;
  fill_common(hcst, host_fe(beta), host_fe(gamma), host_fe(delta), host_fe(zeta), c->n_perm, t_inv, rot);
  {  // term i carries y^(N-1-i) at the level its expression needs (see the kernel's header)
    const uint32_t sets = (c->n_perm + c->chunk_len - 1) / c->chunk_len;
    const uint32_t N = 2 + 2 * sets + (rc_.has_lookup ? 5 : 0);
    const HostY yp(host_fe(y), N);
    auto put = [&](uint32_t term, int level) { hcst.y[term] = h_level(yp.p[N - 1 - term], level); };
    put(0, -3);
    put(1, -2);
    put(2, -3);
    for (uint32_t q = 1; q < sets; q++) put(2 + q, -2);
    for (uint32_t q = 0; q < sets; q++) {
      const uint32_t cols = std::min(c->chunk_len, c->n_perm - q * c->chunk_len);
      put(2 + sets + q, -2 - (int)cols);
    }
    if (rc_.has_lookup) {
      const uint32_t lb = 2 + 2 * sets;
      put(lb, -2);
      put(lb + 1, -3);
      put(lb + 2, -4);
      put(lb + 3, -2);
      put(lb + 4, -3);
    }
  }
  const uint32_t size = 1u << extended_k;
#ifdef H2MI_AB
  static const bool occ4 = ab_env("H2MI_EVALH_OCC") && atoi(ab_env("H2MI_EVALH_OCC")) == 4;
  if (occ4) {
    H2_LAUNCH("k_evaluate_h_range", k_evaluate_h_range_occ4, ceil_div_u32(size, 256), 256, 0, s, rc_, extended_k, k, blinding_factors + 1, hcst,
              (const fe*)px.lo, (const fe*)px.hi, px.h, (fe*)d_h_out);
    return H2MI_OK;
  }
#endif
  H2_LAUNCH("k_evaluate_h_range", k_evaluate_h_range, ceil_div_u32(size, 256), 256, 0, s, rc_, extended_k, k, blinding_factors + 1, hcst,
            (const fe*)px.lo, (const fe*)px.hi, px.h, (fe*)d_h_out);
  return H2MI_OK;
}

int h2mi_plonk_evaluate_h_flex_dev(const h2mi_flex_cosets* c, uint32_t k, uint32_t extended_k, uint32_t blinding_factors, const uint64_t beta[4],
                                   const uint64_t gamma[4], const uint64_t y[4], const uint64_t delta[4], const uint64_t zeta[4],
                                   const uint64_t extended_omega[4], const uint64_t* t_inv, void* d_h_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!c || !beta || !gamma || !y || !delta || !zeta || !extended_omega || !t_inv || !d_h_out) return H2MI_EINVAL;
  if (extended_k < k || extended_k - k > 4 || extended_k > H2MI_MAX_LOG_N) return H2MI_ERANGE;
  if (c->n_gates == 0 || c->n_gates > H2MI_FLEX_MAX_GATES || c->n_perm > H2MI_FLEX_MAX_PERM || c->n_lookups > H2MI_FLEX_MAX_LOOKUPS) return H2MI_EINVAL;
  if (c->n_perm && (c->chunk_len == 0 || c->chunk_len > 3)) return H2MI_EINVAL;
  static_assert(sizeof(FlexCosets) + sizeof(FlexConsts) + 64 <= 4096, "the quotient kernel's arguments travel by value");
  FlexCosets fc;
  memset(&fc, 0, sizeof(fc));
  fc.n_gates = c->n_gates; fc.n_perm = c->n_perm; fc.chunk = c->chunk_len ? c->chunk_len : 1; fc.n_lookups = c->n_lookups;
  for (uint32_t g = 0; g < c->n_gates; g++) {
    fc.gate_a[g] = (const fe*)c->gate_a[g]; fc.gate_q[g] = (const fe*)c->gate_q[g];
    if (!fc.gate_a[g] || !fc.gate_q[g]) return H2MI_EINVAL;
  }
  for (uint32_t j = 0; j < c->n_perm; j++) {
    fc.perm_value[j] = (const fe*)c->perm_value[j]; fc.perm_sigma[j] = (const fe*)c->perm_sigma[j];
    if (!fc.perm_value[j] || !fc.perm_sigma[j]) return H2MI_EINVAL;
  }
  for (uint32_t q = 0; c->n_perm && q < (c->n_perm + fc.chunk - 1) / fc.chunk; q++) {
    fc.perm_z[q] = (const fe*)c->perm_z[q];
    if (!fc.perm_z[q]) return H2MI_EINVAL;
  }
  for (uint32_t l = 0; l < c->n_lookups; l++) {
    fc.lk_in[l] = (const fe*)c->lookup_input[l]; fc.lk_in_b[l] = (const fe*)c->lookup_input_b[l]; fc.lk_table[l] = (const fe*)c->lookup_table[l];
    fc.lk_pin[l] = (const fe*)c->lookup_permuted_input[l]; fc.lk_ptab[l] = (const fe*)c->lookup_permuted_table[l]; fc.lk_z[l] = (const fe*)c->lookup_z[l];
    if (!fc.lk_in[l] || !fc.lk_table[l] || !fc.lk_pin[l] || !fc.lk_ptab[l] || !fc.lk_z[l]) return H2MI_EINVAL;
  }
  fc.l0 = (const fe*)c->l0; fc.l_last = (const fe*)c->l_last; fc.l_active = (const fe*)c->l_active;
  if (!fc.l0 || !fc.l_last || !fc.l_active) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  PowTab px;
  int rc = get_powtab(extended_omega, extended_k, s, &px);
  if (rc) return rc;
  const uint32_t rot = 1u << (extended_k - k);
  FlexConsts hc_;
  memset(&hc_, 0, sizeof(hc_));
  hc_.beta = host_fe(beta); hc_.gamma = host_fe(gamma); hc_.y = host_fe(y); hc_.delta = host_fe(delta); hc_.zeta = host_fe(zeta);
  for (uint32_t i = 0; i < rot; i++) hc_.tinv[i] = host_fe(t_inv + 4 * i);
  const uint32_t size = 1u << extended_k;
  H2_LAUNCH("k_evaluate_h_flex", k_evaluate_h_flex, ceil_div_u32(size, 256), 256, 0, s, fc, extended_k, k, blinding_factors + 1, hc_, (const fe*)px.lo,
            (const fe*)px.hi, px.h, (fe*)d_h_out);
  return H2MI_OK;
}

int h2mi_plonk_evaluate_h_standard_dev(const h2mi_standard_plonk_cosets* c, uint32_t k, uint32_t extended_k, uint32_t blinding_factors,
                                       const uint64_t beta[4], const uint64_t gamma[4], const uint64_t y[4], const uint64_t delta[4],
                                       const uint64_t zeta[4], const uint64_t extended_omega[4], const uint64_t* t_inv, void* d_h_out,
                                       h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!c || !beta || !gamma || !y || !delta || !zeta || !extended_omega || !t_inv || !d_h_out) return H2MI_EINVAL;
  if (extended_k < k || extended_k - k > 4 || extended_k > H2MI_MAX_LOG_N) return H2MI_ERANGE;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  PlonkCosets pc;
  for (int i = 0; i < 3; i++) { pc.advice[i] = (const fe*)c->advice[i]; pc.sigma[i] = (const fe*)c->sigma[i]; pc.z[i] = (const fe*)c->z[i]; }
  for (int i = 0; i < 5; i++) pc.fixed[i] = (const fe*)c->fixed[i];
  pc.l0 = (const fe*)c->l0; pc.l_last = (const fe*)c->l_last; pc.l_active = (const fe*)c->l_active;
  for (int i = 0; i < 3; i++) if (!pc.advice[i] || !pc.sigma[i] || !pc.z[i]) return H2MI_EINVAL;
  for (int i = 0; i < 5; i++) if (!pc.fixed[i]) return H2MI_EINVAL;
  if (!pc.l0 || !pc.l_last || !pc.l_active) return H2MI_EINVAL;
  PowTab px;
  int rc = get_powtab(extended_omega, extended_k, s, &px);
  if (rc) return rc;
  HConsts hcst;
  fill_common(hcst, host_fe(beta), host_fe(gamma), host_fe(delta), host_fe(zeta), 3, t_inv, 1u << (extended_k - k));
  {
    const HostY yp(host_fe(y), 8);  // eight terms: term i carries y^(7-i)
    hcst.y[0] = h_level(yp.p[7], -2);
    hcst.y[1] = h_level(yp.p[7], -3);
    hcst.y[2] = h_level(yp.p[7], -1);
    hcst.y[3] = h_level(yp.p[6], -2);  // term 1
    hcst.y[4] = h_level(yp.p[4], -2);  // term 3
    hcst.y[5] = h_level(yp.p[3], -2);  // term 4
    hcst.y[6] = h_level(yp.p[5], -3);  // term 2
    for (int m = 0; m < 3; m++) hcst.y[7 + m] = h_level(yp.p[2 - m], -3);  // terms 5, 6, 7
  }
  const uint32_t size = 1u << extended_k;
  H2_LAUNCH("k_evaluate_h_standard_plonk", k_evaluate_h_standard_plonk, ceil_div_u32(size, 256), 256, 0, s, pc, extended_k, k, blinding_factors + 1,
            hcst, (const fe*)px.lo, (const fe*)px.hi, px.h, (fe*)d_h_out);
  return H2MI_OK;
}

}  // extern "C"
