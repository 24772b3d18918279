// libh2mi.so — polynomial helpers of the opening argument on device-resident coefficient vectors:
// halo2_proofs::arithmetic::{eval_polynomial, kate_division}, the challenge-weighted linear combinations of ProverSHPLONK
// (poly/kzg/multiopen/shplonk/prover.rs), and the small vector utilities the prover's columns are built with (element-wise
// product, fill, head patch, the seeded random stream, an instance column's coset).  Reached from create_proof
// (reference examples/standard_plonk.rs:41-49, src/scaffold.rs:322-331) through csrc/h2mi_prover.cpp.  Power tables and scratch
// vectors come from h2mi_ntt.hip (h2mi_fr_tables.h).  Split out of h2mi_ntt.hip in round 5; the kernels are unchanged.
#include <algorithm>
#include <vector>

#include "h2mi_fr_tables.h"
#include "scan.cuh"

namespace h2 {



// ---- polynomial helpers of the opening argument: eval_polynomial, kate_division,
// linear combinations.  Bandwidth-leaning vector kernels over HBM-resident coefficient vectors. ----------

// Thread t of T sums the coefficient class i = t (mod T): x^t * P_t(y), y = x^T (coalesced loads); block sums go to `partial`.
// P_t by Horner's rule in y^3, three coefficients per step: acc <- acc y^3 + c2 y^2 + c1 y + c0 as ONE three-product multiplication
// with a shared Montgomery reduction (f29_mul3) — 16 products + 6 reductions per 16 coefficients where the plain rule spent 16 + 16,
// and a dependent chain a third as long.  (Sixteen coefficients at a time as a dot product with y^0 .. y^15 was SLOWER — 24.5 -> 30.5
// us at 2^20: sixteen loads in flight cost 144 registers.)  y, y^2, y^3 come from the host (uniform operands).
struct PolyList {
  const fe* p[24];
};
struct EvalPowers {
  f29 y1, y2, y3;  // Montgomery-2^261
};
__device__ __forceinline__ void eval_poly_body(const fe* poly, size_t n, uint32_t logT, const EvalPowers& yp, const fe* lo, const fe* hi, uint32_t h,
                                               fe* partial) {
  __shared__ fe red[256];
  const uint32_t T = 1u << logT, t = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t m = t < n ? (n - 1 - t) / T + 1 : 0;  // coefficients t, t+T, ... < n
  f29 acc = f29_zero();
  for (size_t g = (m + 2) / 3; g-- > 0;) {
    const size_t j = 3 * g;  // j < m; the top group may run past m: zeros
    const f29 c0 = load_unpack(&poly[t + j * T]);
    const f29 c1 = j + 1 < m ? load_unpack(&poly[t + (j + 1) * T]) : f29_zero();
    const f29 c2 = j + 2 < m ? load_unpack(&poly[t + (j + 2) * T]) : f29_zero();
    acc = f29_normalize(f29_add(f29_mul3<F9>(acc, yp.y3, c2, yp.y2, c1, yp.y1), c0));  // < 1.02 p + p
  }
  f29 term = f29_mul<F9>(acc, pow2tab(lo, hi, h, t));
  fe o;
  f29_pack(f29_reduce_canonical<F9>(term), o.v);
  red[threadIdx.x] = o;
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fe_add<Fr>(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) fe_store(&partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x], red[0]);
}
__global__ void __launch_bounds__(256) k_eval_poly(PolyList polys, size_t n, uint32_t logT, EvalPowers yp, const fe* lo, const fe* hi, uint32_t h, fe* partial) {
  eval_poly_body(polys.p[blockIdx.y], n, logT, yp, lo, hi, h, partial);
}
// the same for polynomials opened at up to EVAL_POINTS different points: polynomial blockIdx.y is evaluated at point grp[blockIdx.y]
constexpr uint32_t EVAL_POINTS = 4;
struct EvalMulti {
  const fe* p[24];
  uint8_t grp[24];
  EvalPowers yp[EVAL_POINTS];
  const fe* lo[EVAL_POINTS];
  const fe* hi[EVAL_POINTS];
};
__global__ void __launch_bounds__(256) k_eval_poly_multi(const EvalMulti em, size_t n, uint32_t logT, uint32_t h, fe* partial) {
  const uint32_t g = em.grp[blockIdx.y];
  eval_poly_body(em.p[blockIdx.y], n, logT, em.yp[g], em.lo[g], em.hi[g], h, partial);
}
// out[y] = sum of the `count` field elements of row y (one block per row)
__global__ void __launch_bounds__(256) k_sum_fe(const fe* in, uint32_t count, fe* out) {
  __shared__ fe red[256];
  in += (size_t)blockIdx.x * count;
  fe acc = fe_zero();
  for (uint32_t i = threadIdx.x; i < count; i += 256) acc = fe_add<Fr>(acc, fe_load(&in[i]));
  red[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fe_add<Fr>(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) fe_store(&out[blockIdx.x], red[0]);
}

// kate_division: q_i = sum_{j > i} a_j b^(j-i-1), in tile-relative form (round 3).  With tiles of 1024 coefficients (tile t =
// [1024 t, 1024 t + 1024), loc = j - 1024 t) and B = b^1024:
//   pass 1  local'_j = sum of a_l b^(l - 1024 t) over l >= j inside the tile (ONE multiplication per coefficient by the tile-relative
//           power b^loc — a 1024-entry table shared by all tiles — and a suffix scan in LDS); T'_t = the tile's total
//   pass 2  O'_t = sum_{t' > t} B^(t' - t) T'_t' = B^-t * SUFFIX_(t' > t)(B^t' T'_t')         (n / 1024 values: one workgroup per root)
//   pass 3  q_i = b^-loc (local'_j + O'_t) for j = i + 1 in tile t                              (ONE multiplication per coefficient)
// The round-2 form scaled by the absolute powers b^j and b^-(i+1): two multiplications per coefficient in each pass (the power
// itself is a product of two table entries) plus the weight's in the several-roots form — five per root where there are now two.
constexpr uint32_t KATE_TILE = 1024;

// out[i] = sum_k scalar_k * poly_k[i]
// Extended-coset form of an instance column WITHOUT transforms (round 3).  The column holds `count` public inputs v_r on rows
// r < count and zeros elsewhere, so its polynomial is sum_r v_r L_r(X) with L_r(X) = L_0(omega^-r X), and on the extended coset
// X_j = g w^j (omega = w^rot): L_r(X_j) = L_0(X_(j - r rot)) — values the proving key already holds as l_0's coset.  out[j] =
// sum_r v_r l0[(j - r rot) mod 2^ext_k]: one pass instead of an iNTT(n) + coset NTT(2^ext_k) (2.8 ms at DEGREE 22).
struct InstanceArgs {
  fe v[16];  // Montgomery-2^256 values times 2^5 (the mixed-domain product's level, see k_kate_finish_multi)
};
__global__ void __launch_bounds__(256) k_instance_coset(const fe* l0, uint32_t ext_k, uint32_t rot, InstanceArgs a, uint32_t count, fe* out) {
  const uint32_t size = 1u << ext_k;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= size) return;
  f29 acc = f29_zero();
#pragma unroll
  for (uint32_t r = 0; r < 16; r++)  // unrolled: a run-time index into the by-value argument would send it through scratch
    if (r < count) {
      acc = f29_add(acc, f29_mul<F9>(load_unpack(&l0[(j + size - r * rot) & (size - 1)]), f29_unpack(a.v[r].v)));
      if ((r & 3u) == 3u) acc = f29_normalize(acc);  // four normalized addends at most between carries (limbs stay below 2^32)
    }
  fe o;
  f29_pack(f29_reduce_loose<F9>(f29_normalize(acc)), o.v);
  fe_store(&out[j], o);
}

// Division by a product of up to four distinct linear factors in ONE round (round 3): for Z(X) = prod (X - r_i) dividing N(X),
// N / Z = sum_i c_i * (N / (X - r_i)), c_i = 1 / prod_{j != i} (r_i - r_j) (partial fractions; every N / (X - r_i) is exact).
// SHPLONK divided a rotation set's numerator by its points one after the other — a chain of up to four dependent three-launch
// divisions (2.0 ms at 2^22 rows, the longest stretch of the opening phase) — where the m divisions are independent: the same
// three launches with blockIdx.y = root, the finish kernel summing the weighted quotients.
constexpr uint32_t KATE_MULTI_MAX = 4;
struct KateRoots {
  const fe* lo[KATE_MULTI_MAX];   // b_r^i tables (split form: pow2tab)
  const fe* hi[KATE_MULTI_MAX];
  const fe* ilo[KATE_MULTI_MAX];  // b_r^-i tables
  const fe* ihi[KATE_MULTI_MAX];
  uint32_t h[KATE_MULTI_MAX], ih[KATE_MULTI_MAX];
  fe c[KATE_MULTI_MAX];           // the partial-fraction weights, Montgomery-2^261 (canonical words)
  uint32_t m;
};
// base^e for e < 1024 (and e below the table's range): a split table with 2^h >= 1024 low entries holds it as one entry
__device__ __forceinline__ f29 kate_small_power(const fe* lo, const fe* hi, uint32_t h, uint32_t e) {
  return h >= 10 ? load_unpack(&lo[e]) : pow2tab(lo, hi, h, e);
}
__device__ __forceinline__ fe fe_shfl_down(const fe& a, uint32_t d) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_down(a.v[i], d);
  return r;
}
// inclusive suffix sums over the 64 lanes of a wavefront (lane l gets the sum of lanes >= l)
__device__ __forceinline__ fe wave_suffix_fe(fe v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    fe o = fe_shfl_down(v, d);
    if (lane + d < 64) v = fe_add<Fr>(v, o);
  }
  return v;
}
__global__ void __launch_bounds__(256) k_kate_local_multi(const fe* a, size_t n, KateRoots R, uint32_t nblocks, fe* local, fe* totals) {
  __shared__ fe tile[KATE_TILE + 8];
  __shared__ fe wtot[4];
  const uint32_t tid = threadIdx.x, r = blockIdx.y, lane = tid & 63u, wave = tid >> 6;
  const size_t base = (size_t)blockIdx.x * KATE_TILE;
  local += (size_t)r * n;
  totals += (size_t)r * nblocks;
  for (uint32_t q = 0; q < 4; q++) {  // coalesced: element base + tid + 256 q
    const uint32_t loc = tid + 256 * q;
    fe o = fe_zero();
    if (base + loc < n) {
      f29 x = f29_mul<F9>(load_unpack(&a[base + loc]), kate_small_power(R.lo[r], R.hi[r], R.h[r], loc));
      f29_pack(f29_reduce_canonical<F9>(x), o.v);
    }
    tile[loc] = o;
  }
  __syncthreads();
  // a thread owns 4 consecutive elements: local suffix, a suffix scan of the thread totals inside the wavefront (shuffles), the four
  // wavefront totals through LDS
  fe e3 = tile[4 * tid + 3], e2 = fe_add<Fr>(tile[4 * tid + 2], e3), e1 = fe_add<Fr>(tile[4 * tid + 1], e2), e0 = fe_add<Fr>(tile[4 * tid], e1);
  const fe incl = wave_suffix_fe(e0);
  if (lane == 0) wtot[wave] = incl;
  fe right = fe_shfl_down(incl, 1);  // everything to the right of this thread's four, inside the wavefront
  if (lane == 63) right = fe_zero();
  __syncthreads();
  for (uint32_t w = wave + 1; w < 4; w++) right = fe_add<Fr>(right, wtot[w]);
  tile[4 * tid] = fe_add<Fr>(e0, right);
  tile[4 * tid + 1] = fe_add<Fr>(e1, right);
  tile[4 * tid + 2] = fe_add<Fr>(e2, right);
  tile[4 * tid + 3] = fe_add<Fr>(e3, right);
  __syncthreads();
  for (uint32_t q = 0; q < 4; q++) {
    const uint32_t loc = tid + 256 * q;
    if (base + loc < n) fe_store(&local[base + loc], tile[loc]);
  }
  if (tid == 0) fe_store(&totals[blockIdx.x], tile[0]);
}
// pass 2, one workgroup of 1024 threads per root: the tile offsets O'_t, and the root's weighted inverse powers ct[loc] = c b^-loc
// (Montgomery-2^261) that pass 3 multiplies by.  A thread owns `per` consecutive tiles.
__global__ void __launch_bounds__(1024) k_kate_offsets_multi(const fe* totals, uint32_t nblocks, size_t n, KateRoots R, fe* offsets, fe* ct) {
  __shared__ fe wtot[16];
  const uint32_t tid = threadIdx.x, r = blockIdx.x, lane = tid & 63u, wave = tid >> 6;
  totals += (size_t)r * nblocks;
  offsets += (size_t)r * nblocks;
  ct += (size_t)r * KATE_TILE;
  if (tid < n) pack_store(&ct[tid], f29_mul<F9>(kate_small_power(R.ilo[r], R.ihi[r], R.ih[r], tid), f29_unpack(R.c[r].v)));
  const uint32_t per = (nblocks + 1023) / 1024;
  const uint32_t t0 = min(tid * per, nblocks), t1 = min(t0 + per, nblocks);
  auto weighted = [&](uint32_t t) {  // B^t T'_t, canonical
    fe o;
    f29_pack(f29_reduce_canonical<F9>(f29_mul<F9>(load_unpack(&totals[t]), pow2tab(R.lo[r], R.hi[r], R.h[r], t * KATE_TILE))), o.v);
    return o;
  };
  fe mine = fe_zero();
  for (uint32_t t = t0; t < t1; t++) mine = fe_add<Fr>(mine, weighted(t));
  const fe incl = wave_suffix_fe(mine);
  if (lane == 0) wtot[wave] = incl;
  fe run = fe_shfl_down(incl, 1);  // the weighted totals of every tile to the right of this thread's
  if (lane == 63) run = fe_zero();
  __syncthreads();
  for (uint32_t w = wave + 1; w < 16; w++) run = fe_add<Fr>(run, wtot[w]);
  for (uint32_t t = t1; t-- > t0;) {
    pack_store(&offsets[t], f29_mul<F9>(f29_unpack(run.v), pow2tab(R.ilo[r], R.ihi[r], R.ih[r], t * KATE_TILE)));
    run = fe_add<Fr>(run, weighted(t));
  }
}
// pass 3: q_i = sum_r (local'_r[j] + O'_r[tile(j)]) * ct_r[loc(j)], j = i + 1: the roots' products share one Montgomery reduction
__global__ void __launch_bounds__(256) k_kate_finish_multi(const fe* local, const fe* offsets, const fe* ct, size_t n, uint32_t m, uint32_t nblocks, fe* q) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 >= n) return;
  const size_t j = i + 1;
  const uint32_t t = (uint32_t)(j / KATE_TILE), loc = (uint32_t)(j % KATE_TILE);
  auto x = [&](uint32_t r) { return f29_unpack(fe_add<Fr>(fe_load(&local[(size_t)r * n + j]), fe_load(&offsets[(size_t)r * nblocks + t])).v); };
  auto w = [&](uint32_t r) { return load_unpack(&ct[(size_t)r * KATE_TILE + loc]); };
  f29 acc;
  if (m == 1) acc = f29_mul<F9>(x(0), w(0));
  else if (m == 2) acc = f29_mul2<F9>(x(0), w(0), x(1), w(1));
  else {
    acc = f29_mul3<F9>(x(0), w(0), x(1), w(1), x(2), w(2));
    if (m == 4) acc = f29_normalize(f29_add(acc, f29_mul<F9>(x(3), w(3))));
  }
  fe o;
  f29_pack(f29_reduce_loose<F9>(acc), o.v);
  fe_store(&q[i], o);
}

constexpr uint32_t LINCOMB_MAX = 24;
struct LincombArgs {
  const fe* poly[LINCOMB_MAX];
  f29 scalar[LINCOMB_MAX];  // Montgomery-2^261 limbs, converted on the host: data (2^256 words as they lie) x scalar stays 2^256
  uint32_t count;
};
// out = sum_k scalar_k poly_k: terms in groups of three sharing ONE Montgomery reduction (f29_mul3: 108 instead of 162
// multiply-adds per term), scalars already in the multiplier's radix (round 3: the kernel converted every scalar per thread
// and term — a second multiplication per term on the vector unit, not the scalar unit as its comment claimed).
__global__ void __launch_bounds__(256) k_lincomb(LincombArgs args, size_t n, fe* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f29 acc = f29_zero();
  uint32_t k = 0;
  for (; k + 3 <= args.count; k += 3) {
    f29 t = f29_mul3<F9>(load_unpack(&args.poly[k][i]), args.scalar[k], load_unpack(&args.poly[k + 1][i]), args.scalar[k + 1],
                         load_unpack(&args.poly[k + 2][i]), args.scalar[k + 2]);
    acc = f29_normalize(f29_add(acc, t));  // each group < 1.02p: at most 8 groups, < 8.2p
  }
  if (args.count - k == 2)
    acc = f29_normalize(f29_add(acc, f29_mul2<F9>(load_unpack(&args.poly[k][i]), args.scalar[k], load_unpack(&args.poly[k + 1][i]), args.scalar[k + 1])));
  else if (args.count - k == 1)
    acc = f29_normalize(f29_add(acc, f29_mul<F9>(load_unpack(&args.poly[k][i]), args.scalar[k])));
  fe o;
  f29_pack(f29_reduce_loose<F9>(acc), o.v);  // < 10p -> canonical without a multiplication
  fe_store(&out[i], o);
}

// out[i] = value
// out[i] = a[i] * b[i] (Montgomery-2^256 in and out): the row values of a product expression, e.g. selector * advice as a
// lookup's input (mixed-domain product: one operand converted, the other taken as it lies in memory)
__global__ void __launch_bounds__(256) k_fr_mul(const fe* a, const fe* b, size_t n, fe* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pack_store(&out[i], f29_mul<F9>(f29_from_mont256<F9>(fe_load(&a[i]).v), load_unpack(&b[i])));
}
__global__ void __launch_bounds__(256) k_fr_fill(fe* out, size_t n, fe value) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) fe_store(&out[i], value);
}
// poly[i] += head[i], i < count <= 16 (the low-degree remainder terms of the opening argument)
struct HeadArgs {
  fe c[16];
};
__global__ void k_fr_add_head(fe* poly, HeadArgs h, uint32_t count) {
  uint32_t i = threadIdx.x;
  if (i < count) fe_store(&poly[i], fe_add<Fr>(fe_load(&poly[i]), h.c[i]));
}
// counter-based SplitMix64 field elements (the seeded stand-in for the prover's `Scalar::random(rng)` sweeps:
// blinding rows, the vanishing argument's random polynomial).  Element i = limbs splitmix64(seed << 32 | 4 i + j),
// j = 0..3, top limb masked to 62 bits, one conditional subtraction of r; the limbs are the Montgomery form.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_fr_random(fe* out, size_t n, uint64_t seed, uint64_t start) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint64_t w = splitmix64((seed << 32) + 4 * (start + i) + j);
    if (j == 3) w &= (1ull << 62) - 1;
    x.v[2 * j] = (uint32_t)w;
    x.v[2 * j + 1] = (uint32_t)(w >> 32);
  }
  fe_store(&out[i], fe_reduce_once<Fr>(x));
}

// ---- the same sweep from a 256-bit key: element i of stream `stream` = Fr::from_u512 of ChaCha20 block i (RFC 7539 block function,
// 64-bit block counter in words 12 - 13, 64-bit stream id in words 14 - 15: the layout of rand_chacha's ChaCha20Rng, whose eight
// next_u64 per Fr::random are exactly one block [RECALL halo2curves: Fr::random = from_u512 of eight next_u64]) --------------------
struct ChaChaKey {
  uint32_t w[8];
};
__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define H2_QR(a, b, c, d)                        \
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);  \
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);  \
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);   \
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7)
__global__ void __launch_bounds__(256) k_fr_random_chacha(fe* out, size_t n, ChaChaKey key, uint64_t stream, uint64_t start) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t ctr = start + i;
  uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6],
                     key.w[7], (uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t x[16];
#pragma unroll
  for (int j = 0; j < 16; j++) x[j] = st[j];
  for (int r = 0; r < 10; r++) {
    H2_QR(0, 4, 8, 12); H2_QR(1, 5, 9, 13); H2_QR(2, 6, 10, 14); H2_QR(3, 7, 11, 15);
    H2_QR(0, 5, 10, 15); H2_QR(1, 6, 11, 12); H2_QR(2, 7, 8, 13); H2_QR(3, 4, 9, 14);
  }
  fe lo, hi;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    lo.v[j] = x[j] + st[j];
    hi.v[j] = x[8 + j] + st[8 + j];
  }
  // from_u512: (lo + hi 2^256) mod r, in Montgomery form: lo R + hi R^2 (Montgomery products accept an unreduced 256-bit operand)
  const fe r2 = fe_const<Fr>(Fr::R2);
  fe_store(&out[i], fe_add<Fr>(fe_mul<Fr>(lo, r2), fe_mul<Fr>(fe_mul<Fr>(hi, r2), r2)));
}
#undef H2_QR

}  // namespace h2

using namespace h2;

extern "C" {

static uint32_t eval_logT(uint32_t log_n) { return log_n > 20 ? 16 : log_n > 12 ? log_n - 4 : 8; }
static EvalPowers eval_powers(const uint64_t point[4], uint32_t logT) {
  EvalPowers yp;  // y = x^T, y^2, y^3 on the host (logT squarings of the same header code)
  yp.y1 = f29_from_mont256<F9>(host_fe(point).v);
  for (uint32_t i = 0; i < logT; i++) yp.y1 = f29_sqr<F9>(yp.y1);
  yp.y2 = f29_sqr<F9>(yp.y1);
  yp.y3 = f29_mul<F9>(yp.y2, yp.y1);
  return yp;
}
// `count` polynomials of n coefficients at one point: one launch (blockIdx.y = polynomial) + one row-sum launch
static int eval_polys(const void* const* d_polys, size_t count, size_t n, const uint64_t point[4], void* d_out, hipStream_t s) {
  uint32_t log_n = 0;
  while (((size_t)1 << log_n) < n) log_n++;
  if (log_n > 30) return H2MI_ERANGE;
  uint32_t logT = eval_logT(log_n);  // T threads, >= 256
  PowTab pt;
  int rc = get_powtab(point, logT, s, &pt);  // x^i, i < T
  if (rc) return rc;
  const uint32_t nblocks = (1u << logT) / 256;
  rc = ensure_tmp((size_t)nblocks * count + 8, s);
  if (rc) return rc;
  PolyList pl;
  memset(&pl, 0, sizeof(pl));
  for (size_t i = 0; i < count; i++) pl.p[i] = (const fe*)d_polys[i];
  const EvalPowers yp = eval_powers(point, logT);
  H2_LAUNCH("k_eval_poly", k_eval_poly, dim3(nblocks, (uint32_t)count), 256, 0, s, pl, n, logT, yp, (const fe*)pt.lo, (const fe*)pt.hi, pt.h, tmp_base());
  H2_LAUNCH("k_sum_fe", k_sum_fe, (uint32_t)count, 256, 0, s, (const fe*)tmp_base(), nblocks, (fe*)d_out);
  return release_tmp(s);
}

// groups of polynomials, group g opened at points[g]: the power tables of all points in one launch, one evaluation launch, one row sum
int h2mi_fr_eval_polys_multi_dev(const void* const* d_polys, const size_t* group_counts, const uint64_t* points, size_t ngroups, size_t n, void* d_out,
                                 h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_polys || !group_counts || !points || !d_out || n == 0 || ngroups == 0) return H2MI_EINVAL;
  size_t total = 0;
  for (size_t g = 0; g < ngroups; g++) {
    if (group_counts[g] == 0) return H2MI_EINVAL;
    total += group_counts[g];
  }
  for (size_t i = 0; i < total; i++)
    if (!d_polys[i]) return H2MI_EINVAL;
  if (ngroups > EVAL_POINTS || total > 24) {  // beyond one launch's descriptor: group by group
    size_t off = 0;
    for (size_t g = 0; g < ngroups; g++) {
      for (size_t c0 = 0; c0 < group_counts[g]; c0 += 24) {  // a launch's descriptor holds 24 polynomials
        const size_t part = std::min<size_t>(24, group_counts[g] - c0);
        int rc = h2mi_fr_eval_polys_dev(d_polys + off, part, n, points + 4 * g, (char*)d_out + 32 * off, stream);
        if (rc) return rc;
        off += part;
      }
    }
    return H2MI_OK;
  }
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  uint32_t log_n = 0;
  while (((size_t)1 << log_n) < n) log_n++;
  if (log_n > 30) return H2MI_ERANGE;
  const uint32_t logT = eval_logT(log_n);
  PowTab pt[EVAL_POINTS];
  int rc = get_powtabs(points, ngroups, logT, s, pt);
  if (rc) return rc;
  const uint32_t nblocks = (1u << logT) / 256;
  rc = ensure_tmp((size_t)nblocks * total + 8, s);
  if (rc) return rc;
  EvalMulti em;
  memset(&em, 0, sizeof(em));
  size_t i = 0;
  for (size_t g = 0; g < ngroups; g++) {
    em.yp[g] = eval_powers(points + 4 * g, logT);
    em.lo[g] = pt[g].lo;
    em.hi[g] = pt[g].hi;
    for (size_t j = 0; j < group_counts[g]; j++, i++) {
      em.p[i] = (const fe*)d_polys[i];
      em.grp[i] = (uint8_t)g;
    }
  }
  H2_LAUNCH("k_eval_poly", k_eval_poly_multi, dim3(nblocks, (uint32_t)total), 256, 0, s, em, n, logT, pt[0].h, tmp_base());
  H2_LAUNCH("k_sum_fe", k_sum_fe, (uint32_t)total, 256, 0, s, (const fe*)tmp_base(), nblocks, (fe*)d_out);
  return release_tmp(s);
}
// builds (or refreshes) the cached power tables of `count` bases at the size the division / evaluation helpers use for n coefficients,
// missing ones in one launch: call it with a rotation set's roots and their inverses before the divisions that use them
int h2mi_fr_powtab_prefetch_dev(const uint64_t* bases, size_t count, size_t n, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!bases || count == 0 || n == 0 || count > 32) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  uint32_t log_n = 0;
  while (((size_t)1 << log_n) < n) log_n++;
  if (log_n > 30) return H2MI_ERANGE;
  std::vector<PowTab> out(count);
  return get_powtabs(bases, count, log_n, pick_stream(stream), out.data());
}

int h2mi_fr_eval_poly_dev(const void* d_poly, size_t n, const uint64_t point[4], void* d_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_poly || !point || !d_out || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  return eval_polys(&d_poly, 1, n, point, d_out, pick_stream(stream));
}

int h2mi_fr_eval_polys_dev(const void* const* d_polys, size_t count, size_t n, const uint64_t point[4], void* d_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_polys || !point || !d_out || n == 0 || count == 0 || count > 24) return H2MI_EINVAL;
  for (size_t i = 0; i < count; i++)
    if (!d_polys[i]) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  return eval_polys(d_polys, count, n, point, d_out, pick_stream(stream));
}

static int kate_multi(const void* d_poly, size_t n, const struct KateRoots& R, void* d_out, hipStream_t s) {
  const uint32_t nblocks = (uint32_t)((n + KATE_TILE - 1) / KATE_TILE), m = R.m;
  int rc = ensure_tmp(m * (n + 2 * (size_t)nblocks + KATE_TILE) + 16, s);
  if (rc) return rc;
  fe* local = tmp_base();
  fe* totals = tmp_base() + m * n;
  fe* offsets = totals + m * nblocks;
  fe* ct = offsets + m * nblocks;
  H2_LAUNCH("k_kate_local", k_kate_local_multi, dim3(nblocks, m), 256, 0, s, (const fe*)d_poly, n, R, nblocks, local, totals);
  H2_LAUNCH("k_kate_offsets", k_kate_offsets_multi, m, 1024, 0, s, (const fe*)totals, nblocks, n, R, offsets, ct);
  H2_LAUNCH("k_kate_finish", k_kate_finish_multi, ceil_div_u32(n - 1, 256), 256, 0, s, (const fe*)local, (const fe*)offsets, (const fe*)ct, n, m, nblocks,
            (fe*)d_out);
  return release_tmp(s);
}

int h2mi_fr_kate_division_dev(const void* d_poly, size_t n, const uint64_t b[4], const uint64_t b_inv[4], void* d_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_poly || !b || !b_inv || !d_out || n < 2) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  uint32_t log_n = 0;
  while (((size_t)1 << log_n) < n) log_n++;
  if (log_n > 30) return H2MI_ERANGE;
  uint64_t both[8];
  memcpy(both, b, 32);
  memcpy(both + 4, b_inv, 32);
  PowTab tabs[2];
  int rc = get_powtabs(both, 2, log_n, s, tabs);  // exponents i + 1 <= n - 1 < 2^log_n; both tables in one launch
  if (rc) return rc;
  const PowTab &pb = tabs[0], &pi = tabs[1];
  KateRoots R;
  memset(&R, 0, sizeof(R));
  R.m = 1;
  R.lo[0] = pb.lo; R.hi[0] = pb.hi; R.h[0] = pb.h;
  R.ilo[0] = pi.lo; R.ihi[0] = pi.hi; R.ih[0] = pi.h;
  f29_pack(f29_reduce_canonical<F9>(f29_const<F9>(F9::ONE)), R.c[0].v);  // weight one, Montgomery-2^261
  return kate_multi(d_poly, n, R, d_out, s);
}

int h2mi_plonk_instance_coset_dev(const void* d_l0_coset, uint32_t k, uint32_t extended_k, const uint64_t* values, size_t count, void* d_out,
                                  h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_l0_coset || !d_out || (count && !values) || count > 16) return H2MI_EINVAL;
  if (k == 0 || extended_k < k || extended_k > H2MI_MAX_LOG_N || count > ((size_t)1 << k)) return H2MI_ERANGE;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  InstanceArgs a;
  memset(&a, 0, sizeof(a));
  for (size_t r = 0; r < count; r++) a.v[r] = h_level(host_fe(values + 4 * r), -1);
  const uint32_t size = 1u << extended_k;
  H2_LAUNCH("k_instance_coset", k_instance_coset, ceil_div_u32(size, 256), 256, 0, s, (const fe*)d_l0_coset, extended_k, 1u << (extended_k - k), a,
            (uint32_t)count, (fe*)d_out);
  return H2MI_OK;
}

int h2mi_fr_kate_division_multi_dev(const void* d_poly, size_t n, const uint64_t* roots, const uint64_t* roots_inv, const uint64_t* weights,
                                    size_t m, void* d_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_poly || !roots || !roots_inv || !weights || !d_out || n < 2 || m == 0 || m > KATE_MULTI_MAX) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  uint32_t log_n = 0;
  while (((size_t)1 << log_n) < n) log_n++;
  if (log_n > 30) return H2MI_ERANGE;
  KateRoots R;
  memset(&R, 0, sizeof(R));
  R.m = (uint32_t)m;
  uint64_t all[2 * KATE_MULTI_MAX * 4];  // roots, then inverses: the missing tables of both in one launch
  memcpy(all, roots, 32 * m);
  memcpy(all + 4 * m, roots_inv, 32 * m);
  PowTab tabs[2 * KATE_MULTI_MAX];
  int rc = get_powtabs(all, 2 * m, log_n, s, tabs);
  if (rc) return rc;
  for (size_t r = 0; r < m; r++) {
    const PowTab &pb = tabs[r], &pi = tabs[m + r];
    R.lo[r] = pb.lo; R.hi[r] = pb.hi; R.h[r] = pb.h;
    R.ilo[r] = pi.lo; R.ihi[r] = pi.hi; R.ih[r] = pi.h;
    // the weight enters a mixed-domain product (see k_kate_finish_multi): c 2^256 -> c 2^261, i.e. five doublings
    fe c = host_fe(weights + 4 * r);
    R.c[r] = h_level(c, -1);
  }
  return kate_multi(d_poly, n, R, d_out, s);
}

int h2mi_fr_lincomb_dev(const void* const* d_polys, const uint64_t* scalars, size_t count, size_t n, void* d_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_polys || !scalars || !d_out || n == 0 || count == 0 || count > LINCOMB_MAX) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  LincombArgs args;
  memset(&args, 0, sizeof(args));
  args.count = (uint32_t)count;
  for (size_t k = 0; k < count; k++) {
    if (!d_polys[k]) return H2MI_EINVAL;
    args.poly[k] = (const fe*)d_polys[k];
    const fe sc = host_fe(scalars + 4 * k);
    args.scalar[k] = f29_from_mont256<F9>(sc.v);  // host-side: the same header code
  }
  H2_LAUNCH("k_lincomb", k_lincomb, ceil_div_u32(n, 256), 256, 0, s, args, n, (fe*)d_out);
  return H2MI_OK;
}


int h2mi_fr_mul_dev(const void* d_a, const void* d_b, size_t n, void* d_out, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_a || !d_b || !d_out || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  H2_LAUNCH("k_fr_mul", k_fr_mul, ceil_div_u32(n, 256), 256, 0, s, (const fe*)d_a, (const fe*)d_b, n, (fe*)d_out);
  return H2MI_OK;
}

int h2mi_fr_fill_dev(void* d_out, size_t n, const uint64_t value[4], h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_out || !value || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  H2_LAUNCH("k_fr_fill", k_fr_fill, ceil_div_u32(n, 256), 256, 0, s, (fe*)d_out, n, host_fe(value));
  return H2MI_OK;
}

int h2mi_fr_add_head_dev(void* d_poly, const uint64_t* head, size_t count, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_poly || !head || count == 0 || count > 16) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  HeadArgs h;
  memset(&h, 0, sizeof(h));
  for (size_t i = 0; i < count; i++) h.c[i] = host_fe(head + 4 * i);
  H2_LAUNCH("k_fr_add_head", k_fr_add_head, 1, 64, 0, s, (fe*)d_poly, h, (uint32_t)count);
  return H2MI_OK;
}

int h2mi_fr_random_dev(void* d_out, size_t n, uint64_t seed, uint64_t start, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_out || n == 0 || seed >> 32) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  H2_LAUNCH("k_fr_random", k_fr_random, ceil_div_u32(n, 256), 256, 0, s, (fe*)d_out, n, seed, start);
  return H2MI_OK;
}

int h2mi_fr_random_chacha_dev(void* d_out, size_t n, const uint8_t key[32], uint64_t stream_id, uint64_t start, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_out || n == 0 || !key) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  ChaChaKey k;
  memcpy(k.w, key, 32);  // little-endian words, as RFC 7539 reads the key
  H2_LAUNCH("k_fr_random_chacha", k_fr_random_chacha, ceil_div_u32(n, 256), 256, 0, s, (fe*)d_out, n, k, stream_id, start);
  return H2MI_OK;
}

}  // extern "C"
