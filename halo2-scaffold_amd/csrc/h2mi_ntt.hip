// libh2mi.so — radix-2 NTT over the BN254 scalar field on gfx950.
//
// Replaces halo2_proofs::arithmetic::best_fft (G = bn256::Fr) and the scaling sweeps of
// poly::EvaluationDomain::{ifft, coeff_to_extended, extended_to_coeff} — SURVEY.md 8a rows a3/a4,
// reached from the reference through create_proof / keygen_pk (examples/standard_plonk.rs:34,41-49).
//
// Algorithm (four-step / Stockham-style autosort, 1..3 passes).  n = N1*N2*N3, N_p = 2^m_p <= 1024:
//   pass p < last : inside every segment of length SEG (n, n/N1, ...) run the size-N_p DFT down each
//                   column (stride S = SEG/N_p), multiply element (k, j) by w_seg^(j*k), store in place;
//   last pass     : size-N_P DFT of every contiguous row, scattered to the digit-reversed output slot.
// Every size-2^m DFT runs in LDS: the tile (2^m x C columns, nine 29-bit limb planes) and the 2^(m-1)
// local twiddles are staged in LDS once, the tile written bit-reversed; m radix-2 DIT stages exchange
// through LDS.  HBM sees each element once per pass (read + write, C*32 B segments).
// The coset pre-scale (a[i] *= g^i) is fused into the first pass's load and the n^-1 post-scale into
// the last pass's store, so EvaluationDomain's extra sweeps over memory disappear.
// Twiddles: w^e for the inter-pass factors comes from two small tables (e = hi*2^h + lo, one extra
// field mul) instead of an n-entry table, so all twiddle data stays L2/LDS resident.
#include <algorithm>
#include <map>

#include "h2mi_fr_tables.h"
#include "scan.cuh"

namespace h2 {

// Arithmetic: the lazy 29-bit-limb layer (f29.cuh).  Data stays in the ABI's Montgomery-2^256 form
// (only unpacked to 9 limbs); every twiddle / scale table is kept in Montgomery-2^261 form, so
// f29_mul(data, twiddle) = data * twiddle * 2^-261 lands back in the data's own domain.
// Butterflies are decimation-in-time: t = v*w is freshly reduced by the multiplication and u +- t grow
// by at most 2p per stage (< 25p after 10 stages, capacity 2^261 = 169p), so stages need no modular
// correction at all — one carry normalisation per output.

// out[i] = (base^(2^log_stride))^i for i < count = 2^bits, written as canonical Montgomery-2^261 words.  Per-challenge
// tables (evaluation points, their inverses) are built on a prover's critical path, so the chain is kept short: the
// lazy 29-bit-limb layer, log_stride squarings, then square-and-multiply over the `bits` exponent bits that can be set
// (the 32-bit-limb version walked all 32 bits: 36 us per launch, 28 launches per proof).
__global__ void __launch_bounds__(256) k_pow_table(fe* out, uint32_t count, fe base, uint32_t log_stride, uint32_t bits) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  f29 b = f29_from_mont256<F9>(base.v);  // Mont261; products of Mont261 values stay Mont261
  for (uint32_t s = 0; s < log_stride; s++) b = f29_sqr<F9>(b);
  f29 r = f29_const<F9>(F9::ONE);
  for (int bit = (int)bits - 1; bit >= 0; bit--) {
    r = f29_sqr<F9>(r);
    if ((i >> bit) & 1u) r = f29_mul<F9>(r, b);
  }
  fe o;
  f29_pack(f29_reduce_canonical<F9>(f29_mul<F9>(r, f29_const<F9>(F9::ONE))), o.v);  // below 2p, then canonical
  fe_store(&out[i], o);
}
// lo (i < 2^h) and hi ((base^(2^h))^i, i < 2^(log_n - h)) halves of a two-level table in one launch
__global__ void __launch_bounds__(256) k_pow_table2(fe* lo, fe* hi, uint32_t h, uint32_t hi_bits, fe base) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool is_hi = blockIdx.y != 0;
  const uint32_t count = 1u << (is_hi ? hi_bits : h);
  if (i >= count) return;
  f29 b = f29_from_mont256<F9>(base.v);
  if (is_hi)
    for (uint32_t s = 0; s < h; s++) b = f29_sqr<F9>(b);
  f29 r = f29_const<F9>(F9::ONE);
  for (int bit = (int)(is_hi ? hi_bits : h) - 1; bit >= 0; bit--) {
    r = f29_sqr<F9>(r);
    if ((i >> bit) & 1u) r = f29_mul<F9>(r, b);
  }
  fe o;
  f29_pack(f29_reduce_canonical<F9>(f29_mul<F9>(r, f29_const<F9>(F9::ONE))), o.v);
  fe_store(&(is_hi ? hi : lo)[i], o);
}

// up to POW_BATCH two-level tables of one geometry in one launch (blockIdx.z = table): a proof's evaluation points and the division
// roots' powers are known together, and one launch per table was a dozen 10-us launches per proof
constexpr uint32_t POW_BATCH = 8;
struct PowBatch {
  fe* lo[POW_BATCH];
  fe* hi[POW_BATCH];
  fe base[POW_BATCH];
};
__global__ void __launch_bounds__(256) k_pow_table2_b(const PowBatch pb, uint32_t h, uint32_t hi_bits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.z;
  const bool is_hi = blockIdx.y != 0;
  const uint32_t count = 1u << (is_hi ? hi_bits : h);
  if (i >= count) return;
  f29 b = f29_from_mont256<F9>(pb.base[t].v);
  if (is_hi)
    for (uint32_t s = 0; s < h; s++) b = f29_sqr<F9>(b);
  f29 r = f29_const<F9>(F9::ONE);
  for (int bit = (int)(is_hi ? hi_bits : h) - 1; bit >= 0; bit--) {
    r = f29_sqr<F9>(r);
    if ((i >> bit) & 1u) r = f29_mul<F9>(r, b);
  }
  fe o;
  f29_pack(f29_reduce_canonical<F9>(f29_mul<F9>(r, f29_const<F9>(F9::ONE))), o.v);
  fe_store(&(is_hi ? pb.hi[t] : pb.lo[t])[i], o);
}

struct PassParams {
  const fe* in;
  fe* out;
  uint32_t log_n, log_seg, m, logC;
  const fe* loc;  // local twiddles w_loc^e, e < 2^(m-1)            (Mont261)
  const fe* tlo;  // omega^i, i < 2^h                                (Mont261)
  const fe* thi;  // omega^(i << h)
  uint32_t h;
  const fe* plo;  // pre-scale tables (first pass only) or null      (Mont261)
  const fe* phi;
  uint32_t ph;
  uint32_t tfull, pfull;  // the twiddle / pre-scale table holds every power: fetch instead of hi * lo
  const fe* wmat;         // non-final pass: the pass's inter-pass twiddles in the order its tiles read them, or null (see k_ntt_wmat_build)
  int has_post;
  f29 post;       // Montgomery-2^261 limbs (converted on the host: every thread of the last pass multiplies by it)
  uint32_t logN1, logN2;  // last pass: digit-reversal geometry
  uint32_t remap;         // XCD-aware block remap on/off
  uint32_t nofuse;        // A/B (-DH2MI_AB, H2MI_NTT_NO_FUSE): the first / last round through LDS like the others
  size_t in_len;          // elements of `in` that exist; indices beyond read as zero (first pass of a
                          // zero-extending transform: coeff_to_extended without materialising the padding)
};

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t m) { return m ? (__brev(x) >> (32 - m)) : 0; }

// blocks b and b+8 share an XCD (and its L2): give each XCD a contiguous run of tiles so that tiles
// sharing 128-B lines (adjacent columns) hit the same L2.  Bijective when nblocks % 8 == 0.
__device__ __forceinline__ uint32_t tile_of_block(uint32_t b, uint32_t nb, uint32_t remap) {
  if (!remap || (nb & 7u)) return b;
  return (b & 7u) * (nb >> 3) + (b >> 3);
}

// LDS image: structure-of-arrays, limb plane l of element i at lds[l * stride + sw(i)].  The swizzle sw XORs
// element-index bits 6..9 into the bank bits (0..5): a wavefront's accesses in a butterfly round s vary index bits
// {0..s-1} and {s+2..7} (rounds 0, 2, 4 would otherwise hit 16 of the 64 banks: 4-way conflicts, measured 25 M
// conflict cycles per 2^20 pass against 4 M of LDS issue), and the bit-reversed first write varies bits 4..9 (16-way).
// With bits (6,7) replicated into (0,1), (2,3), (4,5) and bits (8,9) added to (0,1) every one of these patterns maps its
// 64 lanes to 64 distinct banks.
__device__ __forceinline__ uint32_t lds_sw(uint32_t i) { return i ^ (((i >> 6) & 3u) * 21u) ^ ((i >> 8) & 3u); }
__device__ __forceinline__ f29 lds_get(const uint32_t* lds, uint32_t stride, uint32_t i) {
  f29 r;
  const uint32_t j = lds_sw(i);
#pragma unroll
  for (int l = 0; l < 9; l++) r.v[l] = lds[l * stride + j];
  return r;
}
__device__ __forceinline__ void lds_put(uint32_t* lds, uint32_t stride, uint32_t i, const f29& a) {
  const uint32_t j = lds_sw(i);
#pragma unroll
  for (int l = 0; l < 9; l++) lds[l * stride + j] = a.v[l];
}

// m DIT stages over C tiles of 2^m elements (bit-reversed order in, natural order out), two stages per
// LDS round trip: a thread takes the four elements {i, i+h, i+2h, i+3h} (h = 2^s), does the two
// butterflies of stage s and the two of stage s+1 in registers, and normalises only the four results
// (limb bounds: a = x +- t < 1.5 * 2^30 may feed the next multiplication un-normalised, y < 2.5 * 2^30).
// Half the barriers, LDS traffic and carry normalisations of a radix-2 schedule; same multiplications.
// local twiddles w_loc^e, e < 2^(m-1), staged at position bitrev(e, m-1): stage s then reads the CONTIGUOUS block
// [0, 2^s) — entry bitrev(pos, s) for butterfly position pos — instead of a stride of 2^(m-1-s) packed elements
// (32 B each: 4-way bank conflicts at the late stages, where every lane needs its own twiddle)
// The table is kept as eight 4-byte word planes (word l of entry e at tw[l * 2^(m-1) + e]): consecutive lanes read
// consecutive words, where 32-byte packed entries made every 16-byte read a two-way conflict.
// Only the first TW_STAGED = 128 staged entries live in LDS (4 KB): they serve every round up to stages (6, 7).  The last round of
// a 2^9 / 2^10 DFT (stages 8, 9: entries up to 511) fetches its twiddles from the global table, which is 8 / 16 KB and stays in the
// L1 / L2 (round 3).  Why: a 2^10 tile with 16 KB of staged twiddles takes 52 KB of LDS — three workgroups per CU, so the 1024 tiles
// of a 2^20 pass ran as 768 + 256: a second, one-third-full round.  With 4 KB it is 40 KB: four per CU, all 1024 tiles resident at once.
constexpr uint32_t TW_STAGED = 128;
__device__ __forceinline__ uint32_t tw_staged_count(uint32_t m) { return min(1u << (m - 1), TW_STAGED); }
__device__ __forceinline__ void stage_twiddles(uint32_t* tw, const fe* loc, uint32_t m) {
  if (m == 0) return;
  const uint32_t cnt = tw_staged_count(m);
  for (uint32_t j = threadIdx.x; j < cnt; j += blockDim.x) {
    const fe x = fe_load(&loc[bitrev(j, m - 1)]);
#pragma unroll
    for (int l = 0; l < 8; l++) tw[l * cnt + j] = x.v[l];
  }
}
__device__ __forceinline__ f29 tw_get(const uint32_t* tw, uint32_t cnt, uint32_t j) {
  uint32_t w[8];
#pragma unroll
  for (int l = 0; l < 8; l++) w[l] = tw[l * cnt + j];
  return f29_unpack(w);
}
// staged index j of a round that starts at stage s: LDS while the round's entries (j < 2^(s+1)) are all staged, else the global table
__device__ __forceinline__ f29 tw_fetch(const uint32_t* tw, uint32_t cnt, const fe* loc, uint32_t m, uint32_t s, uint32_t j) {
  if ((2u << s) <= cnt) return tw_get(tw, cnt, j);
  return load_unpack(&loc[bitrev(j, m - 1)]);
}
__device__ __forceinline__ void local_ntt(uint32_t* lds, uint32_t dstride, const uint32_t* tw, const fe* loc, uint32_t m, uint32_t logC) {
  const uint32_t T = blockDim.x, tid = threadIdx.x;
  if (m == 0) return;
  const uint32_t tcnt = tw_staged_count(m);
  uint32_t s = 0;
  if (m >= 2) {
    const uint32_t nq = 1u << (m - 2 + logC);
    // One quad per thread and 64 quads per wavefront: a wavefront's quads of round s stay inside its own aligned block
    // of 256 elements while 2^(s+2) <= 256, so consecutive rounds up to s = 6 exchange data only between lanes of one
    // wavefront.  LDS instructions of a wavefront execute in order, so those rounds need no workgroup barrier: three
    // barriers per 2^10 tile instead of six.
    const bool wave_local = nq == T && (T & 63u) == 0;
    for (; s + 1 < m; s += 2) {
      const uint32_t h = 1u << s;
      for (uint32_t q = tid; q < nq; q += T) {
        uint32_t c = q >> (m - 2);
        uint32_t r = q & ((1u << (m - 2)) - 1);
        uint32_t pos = r & (h - 1);
        uint32_t grp = r >> s;
        uint32_t i = (c << m) | (grp << (s + 2)) | pos;
        f29 x0 = lds_get(lds, dstride, i), x1 = lds_get(lds, dstride, i + h);
        f29 x2 = lds_get(lds, dstride, i + 2 * h), x3 = lds_get(lds, dstride, i + 3 * h);
        f29 t1 = x1, t3 = x3;
        const uint32_t pb = bitrev(pos, s);  // the twiddle table is staged in bit-reversed order (see stage_twiddles)
        if (s != 0) {
          f29 wa = tw_fetch(tw, tcnt, loc, m, s, pb);
          t1 = f29_mul<F9>(x1, wa);
          t3 = f29_mul<F9>(x3, wa);
        }
        f29 a0 = f29_add(x0, t1), a1 = f29_sub(x0, t1, F9::K2);
        f29 a2 = f29_add(x2, t3), a3 = f29_sub(x2, t3, F9::K2);
        f29 u3 = f29_mul<F9>(a3, tw_fetch(tw, tcnt, loc, m, s, 2 * pb + 1));
        if (s == 0) {  // pos = 0: the twiddle of (a0, a2) is omega^0 — no multiplication; a2 = x2 + x3 < 4p, limbs < 2^30
          lds_put(lds, dstride, i, f29_normalize(f29_add(a0, a2)));
          lds_put(lds, dstride, i + 2 * h, f29_normalize(f29_sub(a0, a2, F9::KW4)));
        } else {
          f29 u2 = f29_mul<F9>(a2, tw_fetch(tw, tcnt, loc, m, s, 2 * pb));
          lds_put(lds, dstride, i, f29_normalize(f29_add(a0, u2)));
          lds_put(lds, dstride, i + 2 * h, f29_normalize(f29_sub(a0, u2, F9::K2)));
        }
        lds_put(lds, dstride, i + h, f29_normalize(f29_add(a1, u3)));
        lds_put(lds, dstride, i + 3 * h, f29_normalize(f29_sub(a1, u3, F9::K2)));
      }
      if (wave_local && s + 3 < m && s + 4 <= 8) __builtin_amdgcn_wave_barrier();  // next round is a wave-local radix-4 round
      else __syncthreads();
    }
  }
  if (s < m) {  // odd m: one closing radix-2 stage (s = m - 1; for m = 1 it is the multiplication-free stage 0)
    const uint32_t nbf = 1u << (m - 1 + logC), half = 1u << s;
    for (uint32_t b = tid; b < nbf; b += T) {
      uint32_t c = b >> (m - 1);
      uint32_t i_ = b & ((1u << (m - 1)) - 1);
      uint32_t pos = i_ & (half - 1);
      uint32_t grp = i_ >> s;
      uint32_t i = (c << m) | (grp << (s + 1)) | pos;
      f29 u = lds_get(lds, dstride, i), v = lds_get(lds, dstride, i + half);
      f29 t = v;
      if (s != 0) t = f29_mul<F9>(v, tw_fetch(tw, tcnt, loc, m, s ? s - 1 : 0, bitrev(pos, s)));
      lds_put(lds, dstride, i, f29_normalize(f29_add(u, t)));
      lds_put(lds, dstride, i + half, f29_normalize(f29_sub(u, t, F9::K2)));
    }
    __syncthreads();
  }
}

// ---- fused ends (round 4) -------------------------------------------------------------------------------------------------
// With one quad per thread (T = tile / 4) the four elements a thread LOADS — e0 + k 2^(m-2), k < 4, of one DFT — are exactly the
// quad of the first radix-4 round (bit reversal puts them at four consecutive LDS positions), and the four outputs of the LAST
// round — positions pos + j 2^(m-2) — are exactly what a thread of the store loop writes.  So the first round runs on the loaded
// registers and the last round's results go straight to the store: two of a 2^10 tile's six LDS round trips (initial put, five
// get / put rounds, final get) and one barrier disappear; the arithmetic, and with it every result bit, is that of local_ntt.
// m >= 4 with compile-time geometry only (every pass of the default splits); an odd m ends with the closing radix-2 stage instead.
// one radix-4 round on registers: inputs x0..x3 from positions i, i+h, i+2h, i+3h; outputs for the same positions
template <bool FIRST>
__device__ __forceinline__ void ntt_r4(const f29& x0, const f29& x1, const f29& x2, const f29& x3, const f29& wa, const f29& w2, const f29& w3,
                                       f29& r0, f29& r1, f29& r2, f29& r3) {
  f29 t1 = x1, t3 = x3;
  if (!FIRST) {
    t1 = f29_mul<F9>(x1, wa);
    t3 = f29_mul<F9>(x3, wa);
  }
  const f29 a0 = f29_add(x0, t1), a1 = f29_sub(x0, t1, F9::K2);
  const f29 a2 = f29_add(x2, t3), a3 = f29_sub(x2, t3, F9::K2);
  const f29 u3 = f29_mul<F9>(a3, w3);
  if (FIRST) {  // the twiddle of (a0, a2) is omega^0
    r0 = f29_normalize(f29_add(a0, a2));
    r2 = f29_normalize(f29_sub(a0, a2, F9::KW4));
  } else {
    const f29 u2 = f29_mul<F9>(a2, w2);
    r0 = f29_normalize(f29_add(a0, u2));
    r2 = f29_normalize(f29_sub(a0, u2, F9::K2));
  }
  r1 = f29_normalize(f29_add(a1, u3));
  r3 = f29_normalize(f29_sub(a1, u3, F9::K2));
}
// first round (stages 0, 1) on the loaded elements e_k = element e0 + k 2^(m-2) of DFT c, results to their LDS positions
// (named values, not an array: an f29[4] passed by reference stayed in scratch memory — 304 B per thread, passes 8 - 20 % slower)
__device__ __forceinline__ void ntt_first_round_put(uint32_t* lds, uint32_t dstride, const fe* loc, uint32_t m, uint32_t c, uint32_t e0, const f29& e_0,
                                                    const f29& e_1, const f29& e_2, const f29& e_3) {
  const uint32_t i = (c << m) | (bitrev(e0, m - 2) << 2);
  // bit reversal: element k sits at position i + bitrev2(k): (x0, x1, x2, x3) of the round = elements 0, 2, 1, 3
  const f29 w3 = load_unpack(&loc[1u << (m - 2)]);  // staged entry 1 = loc[bitrev(1, m - 1)]
  f29 r0, r1, r2, r3;
  ntt_r4<true>(e_0, e_2, e_1, e_3, w3, w3, w3, r0, r1, r2, r3);
  lds_put(lds, dstride, i, r0);
  lds_put(lds, dstride, i + 1, r1);
  lds_put(lds, dstride, i + 2, r2);
  lds_put(lds, dstride, i + 3, r3);
}
// rounds s = 2 .. m - 4 through LDS, as local_ntt runs them; ends with a workgroup barrier (the last round regroups the threads)
__device__ __forceinline__ void ntt_middle_rounds(uint32_t* lds, uint32_t dstride, const uint32_t* tw, const fe* loc, uint32_t m, uint32_t logC) {
  const uint32_t tid = threadIdx.x, tcnt = tw_staged_count(m);
  for (uint32_t s = 2; s + 2 < m; s += 2) {
    const uint32_t h = 1u << s;
    const uint32_t c = tid >> (m - 2), r = tid & ((1u << (m - 2)) - 1);
    const uint32_t pos = r & (h - 1), grp = r >> s;
    const uint32_t i = (c << m) | (grp << (s + 2)) | pos;
    const f29 x0 = lds_get(lds, dstride, i), x1 = lds_get(lds, dstride, i + h), x2 = lds_get(lds, dstride, i + 2 * h), x3 = lds_get(lds, dstride, i + 3 * h);
    const uint32_t pb = bitrev(pos, s);
    f29 r0, r1, r2, r3;
    ntt_r4<false>(x0, x1, x2, x3, tw_fetch(tw, tcnt, loc, m, s, pb), tw_fetch(tw, tcnt, loc, m, s, 2 * pb), tw_fetch(tw, tcnt, loc, m, s, 2 * pb + 1), r0, r1,
                  r2, r3);
    lds_put(lds, dstride, i, r0);
    lds_put(lds, dstride, i + h, r1);
    lds_put(lds, dstride, i + 2 * h, r2);
    lds_put(lds, dstride, i + 3 * h, r3);
    // wave-local while the next round's quads stay inside a wavefront's own 256 elements — and the next round is not the last
    if (s + 4 < m && s + 4 <= 8) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
  }
}
// last round (stages m-2, m-1) for the quad (c, pos): y[j] = the DFT's output k = pos + j 2^(m-2)
__device__ __forceinline__ void ntt_last_round_get(const uint32_t* lds, uint32_t dstride, const uint32_t* tw, const fe* loc, uint32_t m, uint32_t c,
                                                   uint32_t pos, f29& y0, f29& y1, f29& y2, f29& y3) {
  const uint32_t s = m - 2, h = 1u << s, tcnt = tw_staged_count(m);
  const uint32_t i = (c << m) | pos;
  const f29 x0 = lds_get(lds, dstride, i), x1 = lds_get(lds, dstride, i + h), x2 = lds_get(lds, dstride, i + 2 * h), x3 = lds_get(lds, dstride, i + 3 * h);
  const uint32_t pb = bitrev(pos, s);
  ntt_r4<false>(x0, x1, x2, x3, tw_fetch(tw, tcnt, loc, m, s, pb), tw_fetch(tw, tcnt, loc, m, s, 2 * pb), tw_fetch(tw, tcnt, loc, m, s, 2 * pb + 1), y0, y1, y2,
                y3);
}
// odd m: the closing radix-2 stage s = m - 1 for the thread's two butterflies (pos, pos + 2Q) and (pos + Q, pos + 3Q), Q = 2^(m-2):
// the same four outputs k = pos + j Q
__device__ __forceinline__ void ntt_last_stage_get_odd(const uint32_t* lds, uint32_t dstride, const uint32_t* tw, const fe* loc, uint32_t m, uint32_t c,
                                                       uint32_t pos, f29& y0, f29& y1, f29& y2, f29& y3) {
  const uint32_t s = m - 1, half = 1u << s, Q = 1u << (m - 2), tcnt = tw_staged_count(m);
  const uint32_t i = (c << m) | pos;
  const f29 u0 = lds_get(lds, dstride, i), v0 = lds_get(lds, dstride, i + half);
  const f29 u1 = lds_get(lds, dstride, i + Q), v1 = lds_get(lds, dstride, i + Q + half);
  const f29 t0 = f29_mul<F9>(v0, tw_fetch(tw, tcnt, loc, m, s - 1, bitrev(pos, s)));
  const f29 t1 = f29_mul<F9>(v1, tw_fetch(tw, tcnt, loc, m, s - 1, bitrev(pos + Q, s)));
  y0 = f29_normalize(f29_add(u0, t0));
  y2 = f29_normalize(f29_sub(u0, t0, F9::K2));
  y1 = f29_normalize(f29_add(u1, t1));
  y3 = f29_normalize(f29_sub(u1, t1, F9::K2));
}
template <uint32_t M>
__device__ __forceinline__ void ntt_last_get(const uint32_t* lds, uint32_t dstride, const uint32_t* tw, const fe* loc, uint32_t c, uint32_t pos, f29& y0,
                                             f29& y1, f29& y2, f29& y3) {
  if (M & 1u) ntt_last_stage_get_odd(lds, dstride, tw, loc, M, c, pos, y0, y1, y2, y3);
  else ntt_last_round_get(lds, dstride, tw, loc, M, c, pos, y0, y1, y2, y3);
}
constexpr bool ntt_fused_geometry(uint32_t DS, uint32_t M) { return DS != 0 && M >= 4; }


extern __shared__ uint32_t h2_smem[];

// Compile-time geometry (round 3; tools/ntt_isa_budget.py): DS = elements per tile (limb-plane stride of the LDS image), M =
// log2 of the DFT size.  With both known the nine plane offsets l * DS * 4 of every LDS get / put and the eight of every staged-
// twiddle get become immediate offsets of the ds instructions instead of nine / eight v_add each — 16 of the 32 VALU
// instructions of a get + put pair, ~6 % of a pass — and the round's strides and shifts become constants.  DS = M = 0: the
// run-time form (small or non-default tiles).
// non-final pass: column DFTs inside segments, in-place layout
template <uint32_t DS, uint32_t M>
__global__ void __launch_bounds__(512) k_ntt_pass_col(PassParams p) {
  const uint32_t m = M ? M : p.m, logC = (DS && M) ? (uint32_t)(__builtin_ctz(DS ? DS : 1u) - M) : p.logC, C = 1u << logC;
  const uint32_t dstride = DS ? DS : (C << m);
  uint32_t* lds = h2_smem;
  uint32_t* tw = lds + 9 * dstride;  // 8 word planes of packed twiddles, at most TW_STAGED entries: a 2^10 tile is 40 KiB = 4 blocks/CU
  const uint32_t T = blockDim.x, tid = threadIdx.x;
  const uint32_t logS = p.log_seg - m;
  const uint32_t tile = tile_of_block(blockIdx.x, gridDim.x, p.remap);
  const uint32_t tiles_per_seg_log = logS - logC;
  const uint32_t seg = tile >> tiles_per_seg_log;
  const uint32_t jl0 = (tile & ((1u << tiles_per_seg_log) - 1)) << logC;
  const size_t base = (size_t)seg << p.log_seg;
  const uint32_t sh = p.log_n - p.log_seg;

  if constexpr (ntt_fused_geometry(DS, M)) if (T * 4 == DS && !p.nofuse) {  // one quad per thread: first round on the loaded registers, last round into the store
    const uint32_t c = tid & (C - 1), e0 = tid >> logC, Q = 1u << (m - 2);
    auto ld = [&](uint32_t k) {
      const size_t idx = base + ((size_t)(e0 + k * Q) << logS) + jl0 + c;
      f29 x = f29_zero();
      if (idx < p.in_len) {
        x = load_unpack(&p.in[idx]);
        if (p.plo) x = f29_mul<F9>(x, powtab(p.plo, p.phi, p.ph, p.pfull, (uint32_t)idx));
      }
      return x;
    };
    {
      const f29 x0 = ld(0), x1 = ld(1), x2 = ld(2), x3 = ld(3);
      ntt_first_round_put(lds, dstride, p.loc, m, c, e0, x0, x1, x2, x3);
    }
    stage_twiddles(tw, p.loc, m);
    __syncthreads();
    ntt_middle_rounds(lds, dstride, tw, p.loc, m, logC);
    f29 y0, y1, y2, y3;
    ntt_last_get<M>(lds, dstride, tw, p.loc, c, e0, y0, y1, y2, y3);  // pos = e0: this thread's outputs k = e0 + j Q of DFT c
    auto st = [&](uint32_t j, const f29& y) {
      const uint32_t k = e0 + j * Q, o = (k << logC) | c;
      f29 v;
      if (p.wmat) {
        v = f29_mul<F9>(y, load_unpack(&p.wmat[((size_t)(jl0 >> logC) << (m + logC)) + o]));
      } else {
        v = f29_mul<F9>(y, powtab(p.tlo, p.thi, p.h, p.tfull, ((jl0 + c) * k) << sh));
      }
      fe o_;
      f29_pack(v, o_.v);
      fe_store(&p.out[base + ((size_t)k << logS) + jl0 + c], o_);
    };
    st(0, y0);
    st(1, y1);
    st(2, y2);
    st(3, y3);
    return;
  }

  for (uint32_t o = tid; o < (C << m); o += T) {
    uint32_t c = o & (C - 1), e = o >> logC;
    size_t idx = base + ((size_t)e << logS) + jl0 + c;
    // zero-extended input (coeff_to_extended: three quarters of a 4n coset transform's input are padding): a padded element needs
    // neither its load nor its coset power (two table loads and two multiplications above 2^22) — round 4: coset 2^24 from 2^22
    f29 x = f29_zero();
    if (idx < p.in_len) {
      x = load_unpack(&p.in[idx]);
      if (p.plo) x = f29_mul<F9>(x, powtab(p.plo, p.phi, p.ph, p.pfull, (uint32_t)idx));
    }
    lds_put(lds, dstride, (c << m) | bitrev(e, m), x);
  }
  stage_twiddles(tw, p.loc, m);
  __syncthreads();
  local_ntt(lds, dstride, tw, p.loc, m, logC);
  for (uint32_t o = tid; o < (C << m); o += T) {
    uint32_t c = o & (C - 1), k = o >> logC;
    f29 x = lds_get(lds, dstride, (c << m) | k);
    if (p.wmat) {  // tile-ordered copy of the twiddles: entry o of this tile, whichever segment the tile sits in
      x = f29_mul<F9>(x, load_unpack(&p.wmat[((size_t)(jl0 >> logC) << (m + logC)) + o]));
    } else {
      uint32_t ex = ((jl0 + c) * k) << sh;  // < n
      x = f29_mul<F9>(x, powtab(p.tlo, p.thi, p.h, p.tfull, ex));
    }
    // the product is normalized and below 1.2 p (< 2^255): stored as it is, without the canonical reduction — the next pass
    // reads it as a loosely reduced input (its lazy rounds then stay below 34 p of the 169 p capacity; the LAST pass alone
    // returns canonical values)
    fe o_;
    f29_pack(x, o_.v);
    fe_store(&p.out[base + ((size_t)k << logS) + jl0 + c], o_);
  }
}

// final pass: row DFTs, digit-reversed scatter
template <uint32_t DS, uint32_t M>
__global__ void __launch_bounds__(512) k_ntt_pass_row(PassParams p) {
  const uint32_t m = M ? M : p.m, logC = (DS && M) ? (uint32_t)(__builtin_ctz(DS ? DS : 1u) - M) : p.logC, C = 1u << logC;
  const uint32_t dstride = DS ? DS : (C << m);
  uint32_t* lds = h2_smem;
  uint32_t* tw = lds + 9 * dstride;
  const uint32_t T = blockDim.x, tid = threadIdx.x;
  const uint32_t tile = tile_of_block(blockIdx.x, gridDim.x, p.remap);
  const uint32_t k2 = tile & ((1u << p.logN2) - 1);
  const uint32_t k1_0 = (tile >> p.logN2) << logC;

  if constexpr (ntt_fused_geometry(DS, M)) if (T * 4 == DS && !p.nofuse) {  // see k_ntt_pass_col
    const uint32_t Q = 1u << (m - 2);
    {
      const uint32_t c = tid >> (m - 2), e0 = tid & (Q - 1);  // loads: a wavefront reads 64 consecutive elements of one row
      const size_t rho = ((size_t)(k1_0 + c) << p.logN2) + k2;
      auto ld = [&](uint32_t k) {
        const size_t idx = (rho << m) + e0 + k * Q;
        f29 x = f29_zero();
        if (idx < p.in_len) {
          x = load_unpack(&p.in[idx]);
          if (p.plo) x = f29_mul<F9>(x, powtab(p.plo, p.phi, p.ph, p.pfull, (uint32_t)idx));
        }
        return x;
      };
      const f29 x0 = ld(0), x1 = ld(1), x2 = ld(2), x3 = ld(3);
      ntt_first_round_put(lds, dstride, p.loc, m, c, e0, x0, x1, x2, x3);
    }
    stage_twiddles(tw, p.loc, m);
    __syncthreads();
    ntt_middle_rounds(lds, dstride, tw, p.loc, m, logC);
    const uint32_t c = tid & (C - 1), pos = tid >> logC;  // stores: the C rows' outputs k are adjacent in memory
    f29 y0, y1, y2, y3;
    ntt_last_get<M>(lds, dstride, tw, p.loc, c, pos, y0, y1, y2, y3);
    const f29 fin = p.post;
    auto st = [&](uint32_t j, const f29& y) {
      const uint32_t k = pos + j * Q;
      const size_t oidx = (size_t)(k1_0 + c) + ((size_t)k2 << p.logN1) + ((size_t)k << (p.logN1 + p.logN2));
      if (p.has_post) {
        pack_store(&p.out[oidx], f29_mul<F9>(y, fin));
      } else {
        fe o_;
        f29_pack(f29_reduce_loose<F9>(y), o_.v);
        fe_store(&p.out[oidx], o_);
      }
    };
    st(0, y0);
    st(1, y1);
    st(2, y2);
    st(3, y3);
    return;
  }

  for (uint32_t o = tid; o < (C << m); o += T) {
    uint32_t c = o >> m, e = o & ((1u << m) - 1);
    size_t rho = ((size_t)(k1_0 + c) << p.logN2) + k2;
    size_t idx = (rho << m) + e;
    // zero-extended input (coeff_to_extended: three quarters of a 4n coset transform's input are padding): a padded element needs
    // neither its load nor its coset power (two table loads and two multiplications above 2^22) — round 4: coset 2^24 from 2^22
    f29 x = f29_zero();
    if (idx < p.in_len) {
      x = load_unpack(&p.in[idx]);
      if (p.plo) x = f29_mul<F9>(x, powtab(p.plo, p.phi, p.ph, p.pfull, (uint32_t)idx));
    }
    lds_put(lds, dstride, (c << m) | bitrev(e, m), x);
  }
  stage_twiddles(tw, p.loc, m);
  __syncthreads();
  local_ntt(lds, dstride, tw, p.loc, m, logC);
  // the caller's post-scale also brings the lazily accumulated value back below 2p
  const f29 fin = p.post;
  for (uint32_t o = tid; o < (C << m); o += T) {
    uint32_t c = o & (C - 1), k = o >> logC;
    f29 x = lds_get(lds, dstride, (c << m) | k);
    size_t oidx = (size_t)(k1_0 + c) + ((size_t)k2 << p.logN1) + ((size_t)k << (p.logN1 + p.logN2));
    if (p.has_post) {
      pack_store(&p.out[oidx], f29_mul<F9>(x, fin));
    } else {  // m/2 lazy rounds leave a value below (2 + 6 * ceil(m/2)) p <= 32p: reduce it directly
      fe o_;
      f29_pack(f29_reduce_loose<F9>(x), o_.v);
      fe_store(&p.out[oidx], o_);
    }
  }
}

// Inter-pass twiddles in TILE ORDER (round 4).  A column-pass tile multiplies its element (column j = jl0 + c, row k) by
// omega^((j k) << sh).  Fetched from the full power table omega^i that is a gather with stride j * 32 B: every 32-byte entry
// sits in a 128-byte line of its own, and the pass fetched 130 MB where it needs 32 MB of data and 32 MB of twiddles
// (profiles/r03_traffic.json: 129.7 MB per k_ntt_pass_col<1024,10> launch; profiles/r04_ntt_traffic.txt has both forms).  The
// matrix W[tile][o] = omega^((j k) << sh), o = (k << logC) + c, j = tile * C + c, holds the same 2^log_seg values in the order
// the tiles' threads read them: consecutive lanes read consecutive 32-byte entries.  Built once per plan from the full table.
// `lo` / `hi` / `h`: the plan's power table — the full table (h = log_n: one entry per power, copied) or, above 2^22, the two-level
// table (an entry = the product of two: the multiplication the pass kernel then no longer does per element and pass).
__global__ void __launch_bounds__(256) k_ntt_wmat_build(const fe* lo, const fe* hi, uint32_t h, uint32_t full, fe* W, uint32_t log_seg, uint32_t m,
                                                         uint32_t logC, uint32_t sh) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >> log_seg) return;
  const uint32_t o = (uint32_t)idx & ((1u << (m + logC)) - 1), tile = (uint32_t)(idx >> (m + logC));
  const uint32_t c = o & ((1u << logC) - 1), k = o >> logC, j = (tile << logC) + c;
  const uint32_t e = (j * k) << sh;  // < n
  if (full) {
    fe_store(&W[idx], fe_load(&lo[e]));
  } else {
    fe v;
    f29_pack(f29_reduce_canonical<F9>(pow2tab(lo, hi, h, e)), v.v);
    fe_store(&W[idx], v);
  }
}

// a[i] = a[i] * base^i (* post)
__global__ void __launch_bounds__(256) k_scale_powers(fe* a, size_t n, const fe* lo, const fe* hi, uint32_t h, int has_post, f29 post261) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f29 x = load_unpack(&a[i]);
  x = f29_mul<F9>(x, pow2tab(lo, hi, h, (uint32_t)i));
  if (has_post) x = f29_mul<F9>(x, post261);  // converted to the multiplier's radix on the host, once
  pack_store(&a[i], x);
}
// out[i] = base^i in the ABI's Montgomery-2^256 form
__global__ void __launch_bounds__(256) k_powers(fe* out, size_t n, const fe* lo, const fe* hi, uint32_t h) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe o;
  f29_to_mont256<F9>(pow2tab(lo, hi, h, (uint32_t)i), o.v);
  fe_store(&out[i], o);
}

// ---- host side: plans and table caches -----------------------------------------------------------
struct Key {
  uint64_t w[4];
  uint32_t log_n;
  bool full = false;
  bool operator<(const Key& o) const {
    if (log_n != o.log_n) return log_n < o.log_n;
    if (full != o.full) return full < o.full;
    return memcmp(w, o.w, 32) < 0;
  }
};
struct Plan {
  int P = 0;
  uint32_t m[3] = {0, 0, 0};
  PowTab tw;
  fe* loc[3] = {nullptr, nullptr, nullptr};
  fe* wmat[3] = {nullptr, nullptr, nullptr};  // tile-ordered inter-pass twiddles of the non-final passes (k_ntt_wmat_build)
  uint32_t wlogC[3] = {0, 0, 0};              // the tile geometry each was built for
  size_t wmat_bytes = 0;                      // what the matrices hold of the cache budget (g_wmat_bytes)
  Built built;
  uint64_t last_use = 0;
};

static std::map<Key, PowTab> g_powtabs;
static std::map<Key, Plan> g_plans;
// scratch vectors for the NTT passes and the polynomial helpers: one per stream that issues them (the library's own
// stream, plus a caller's side stream that runs transforms beside it — see h2mi_stream_create), so that calls on
// different streams do not serialise on a shared buffer.  Calls are serialised by the library mutex: ensure_tmp() points
// g_tmp at the scratch of the calling stream for the duration of the call; release_tmp() records the event that orders
// a later user of the same scratch on another stream (only when more streams than scratches are in play).
struct Scratch {
  fe* p = nullptr;
  size_t elems = 0;
  hipEvent_t event = nullptr;
  hipStream_t stream = nullptr;
  bool used = false;
  uint64_t last = 0;
};
constexpr int N_SCRATCH = 4;
static Scratch g_scratch[N_SCRATCH];
static Scratch* g_cur = nullptr;
static uint64_t g_scratch_clock = 0;
static fe* g_tmp = nullptr;

// `full`: one table entry per power (32 B x 2^log_n) so that a kernel fetches base^e instead of multiplying
// two table entries: for the bases that live as long as a domain (omega, the coset generator), up to 2^22.
//
// Cache policy.  Tables are keyed by (base, size); per-domain bases (omega, zeta) are hit by every proof,
// per-challenge bases (evaluation points, their inverses) only by one.  The cache is bounded; when it is full
// the least recently used entries are dropped, never an entry the running call has already been handed
// (`g_epoch` counts ABI calls; an entry touched in this epoch is pinned), and only after the whole device is
// idle, since tables may be in use on any stream.
constexpr uint32_t FULL_TABLE_MAX_LOG = 22;
constexpr uint32_t NTT_TILE_LOG = 10;  // elements a pass stages per workgroup (2^10: four tiles of 40 KB per CU)
constexpr uint32_t WMAT_MAX_LOG = 24;  // largest transform whose inter-pass twiddles are kept as tile-ordered matrices
constexpr size_t POWTAB_MAX_ENTRIES = 64, POWTAB_KEEP_ENTRIES = 32;
constexpr size_t PLAN_KEEP = 24;  // transform plans (one per (omega, size): a prover uses four to six) kept through an eviction
constexpr size_t POWTAB_MAX_BYTES = (size_t)3 << 30, POWTAB_KEEP_BYTES = (size_t)3 << 29;
static size_t g_powtab_bytes = 0;
// the plans' tile-ordered twiddle matrices (32 MB at 2^20, 512 MB at 2^24) count against a budget of their own: a process that
// transforms at many sizes would otherwise pile them up unseen by the evictions (round-4 ADVICE)
constexpr size_t WMAT_MAX_BYTES = (size_t)8 << 30, WMAT_KEEP_BYTES = (size_t)4 << 30;
static size_t g_wmat_bytes = 0;
static uint64_t g_epoch = 1, g_evictions = 0;
void tables_new_call() { g_epoch++; }
fe* tmp_base() { return g_tmp; }

// Power tables are ONE allocation (lo, then hi) and evicted ones are kept in a small pool keyed by size instead of going back to the
// runtime (round 4): a proof builds about a dozen per-challenge tables (evaluation points, their inverses: 16 - 64 KB each), and two
// hipMalloc + two hipFree per table were host time on every proof's critical path.  In a prover loop the sizes repeat, so after the first
// eviction no table build allocates at all.
constexpr size_t TABLE_POOL_MAX_BYTES = (size_t)256 << 20;
static std::multimap<size_t, fe*> g_table_pool;
static size_t g_table_pool_bytes = 0;
static fe* table_alloc(size_t bytes) {
  auto it = g_table_pool.find(bytes);
  if (it != g_table_pool.end()) {
    fe* p = it->second;
    g_table_pool.erase(it);
    g_table_pool_bytes -= bytes;
    return p;
  }
  fe* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  return p;
}
// the caller has made sure nothing reads the table any more (evictions run behind a device synchronisation)
static void table_release(fe* p, size_t bytes) {
  if (!p) return;
  if (g_table_pool_bytes + bytes <= TABLE_POOL_MAX_BYTES) {
    g_table_pool.insert({bytes, p});
    g_table_pool_bytes += bytes;
  } else {
    H2_IGNORE(hipFree(p));
  }
}
static void table_pool_clear() {
  for (auto& kv : g_table_pool) H2_IGNORE(hipFree(kv.second));
  g_table_pool.clear();
  g_table_pool_bytes = 0;
}

static void free_plan(Plan& pl) {
  for (int i = 0; i < 3; i++) {
    fe* p = pl.loc[i];
    if (!p) continue;
    for (int j = i + 1; j < 3; j++)  // passes of equal size share one table
      if (pl.loc[j] == p) pl.loc[j] = nullptr;
    H2_IGNORE(hipFree(p));
    pl.loc[i] = nullptr;
  }
  for (int i = 0; i < 3; i++) {
    if (pl.wmat[i]) H2_IGNORE(hipFree(pl.wmat[i]));
    pl.wmat[i] = nullptr;
  }
  g_wmat_bytes -= std::min(g_wmat_bytes, pl.wmat_bytes);
  pl.wmat_bytes = 0;
  pl.built.destroy();
}

static int evict_tables() {
  H2_HIP(hipDeviceSynchronize());  // every stream: a table may be read by kernels the caller queued elsewhere
  // What fills the cache in a prover loop are the per-challenge tables (evaluation points, their inverses: a dozen per proof, used
  // once); the per-domain plans and the omega / coset tables they reference are hit by every proof.  So: unreferenced tables go
  // first, oldest first, and plans are dropped only when that was not enough.  (Until round 4 every plan not used by the running
  // call was freed on every eviction — harmless while a plan was two small twiddle tables, but a plan now owns its tile-ordered
  // twiddle matrices, 32 - 512 MB: a proof loop rebuilt them every five or six proofs, k_ntt_wmat_build inside the steady state.)
  auto drop_unreferenced = [&]() {
    std::vector<std::pair<uint64_t, Key>> order;
    for (auto& kv : g_powtabs) {
      if (kv.second.last_use >= g_epoch) continue;  // handed out during this call
      bool referenced = false;
      for (auto& pk : g_plans) referenced = referenced || pk.second.tw.lo == kv.second.lo;
      if (!referenced) order.push_back({kv.second.last_use, kv.first});
    }
    std::sort(order.begin(), order.end(), [](const std::pair<uint64_t, Key>& a, const std::pair<uint64_t, Key>& b) { return a.first < b.first; });
    for (auto& e : order) {
      if (g_powtabs.size() <= POWTAB_KEEP_ENTRIES && g_powtab_bytes <= POWTAB_KEEP_BYTES) break;
      auto it = g_powtabs.find(e.second);
      table_release(it->second.lo, it->second.bytes);  // hi lives in the same allocation
      it->second.built.destroy();
      g_powtab_bytes -= it->second.bytes;
      g_powtabs.erase(it);
    }
  };
  drop_unreferenced();
  if (g_powtabs.size() > POWTAB_KEEP_ENTRIES || g_powtab_bytes > POWTAB_KEEP_BYTES || g_plans.size() > PLAN_KEEP || g_wmat_bytes > WMAT_KEEP_BYTES) {
    // still over: the plans of domains no longer in use (oldest first), then the tables they held
    std::vector<std::pair<uint64_t, Key>> plans;
    for (auto& kv : g_plans)
      if (kv.second.last_use < g_epoch) plans.push_back({kv.second.last_use, kv.first});
    std::sort(plans.begin(), plans.end(), [](const std::pair<uint64_t, Key>& a, const std::pair<uint64_t, Key>& b) { return a.first < b.first; });
    for (auto& e : plans) {
      auto it = g_plans.find(e.second);
      free_plan(it->second);
      g_plans.erase(it);
      drop_unreferenced();
      if (g_powtabs.size() <= POWTAB_KEEP_ENTRIES && g_powtab_bytes <= POWTAB_KEEP_BYTES && g_plans.size() <= PLAN_KEEP && g_wmat_bytes <= WMAT_KEEP_BYTES) break;
    }
  }
  g_evictions++;
  return H2MI_OK;
}

int get_powtab(const uint64_t base[4], uint32_t log_n, hipStream_t s, PowTab* out, bool full) {
  full = full && log_n <= FULL_TABLE_MAX_LOG && !ab_env("H2MI_NTT_NO_FULL_TABLES");
  Key k;
  memcpy(k.w, base, 32);
  k.log_n = log_n;
  k.full = full;
  auto it = g_powtabs.find(k);
  if (it != g_powtabs.end()) {
    it->second.last_use = g_epoch;
    H2_HIP(it->second.built.use(s));
    *out = it->second;
    return H2MI_OK;
  }
  static const size_t max_entries = getenv("H2MI_POWTAB_MAX") ? (size_t)atoi(getenv("H2MI_POWTAB_MAX")) : POWTAB_MAX_ENTRIES;
  if (g_powtabs.size() >= max_entries || g_powtab_bytes > POWTAB_MAX_BYTES) {
    int rc = evict_tables();
    if (rc) return rc;
  }
  PowTab t;
  t.full = full;
  t.h = full ? log_n : (log_n + 1) / 2;
  uint32_t nlo = 1u << t.h, nhi = 1u << (log_n - t.h);
  t.bytes = ((size_t)nlo + nhi) * 32;
  t.lo = table_alloc(t.bytes);
  if (!t.lo) return H2MI_ENOMEM;
  t.hi = t.lo + nlo;
  fe b = host_fe(base);
  H2_LAUNCH("k_pow_table", k_pow_table2, dim3(ceil_div_u32(std::max(nlo, nhi), 256), 2), 256, 0, s, t.lo, t.hi, t.h, log_n - t.h, b);
  H2_HIP(t.built.mark(s));
  t.last_use = g_epoch;
  g_powtab_bytes += t.bytes;
  g_powtabs[k] = t;
  *out = t;
  return H2MI_OK;
}

// the tables of `m` bases at one size: cache hits as get_powtab, the missing ones allocated and built by ONE launch per POW_BATCH
int get_powtabs(const uint64_t* bases /* m x 4 */, size_t m, uint32_t log_n, hipStream_t s, PowTab* out) {
  std::vector<size_t> missing;
  for (size_t i = 0; i < m; i++) {
    Key k;
    memcpy(k.w, bases + 4 * i, 32);
    k.log_n = log_n;
    k.full = false;
    auto it = g_powtabs.find(k);
    if (it != g_powtabs.end()) {
      it->second.last_use = g_epoch;
      H2_HIP(it->second.built.use(s));
      out[i] = it->second;
      continue;
    }
    bool dup = false;  // the same base twice in one request: built once, copied below
    for (size_t j : missing) dup = dup || memcmp(bases + 4 * j, bases + 4 * i, 32) == 0;
    if (!dup) missing.push_back(i);
  }
  static const size_t max_entries = getenv("H2MI_POWTAB_MAX") ? (size_t)atoi(getenv("H2MI_POWTAB_MAX")) : POWTAB_MAX_ENTRIES;
  if (!missing.empty() && (g_powtabs.size() + missing.size() > max_entries || g_powtab_bytes > POWTAB_MAX_BYTES)) {
    int rc = evict_tables();  // entries handed out above carry this call's epoch: never evicted
    if (rc) return rc;
  }
  const uint32_t h = (log_n + 1) / 2, nlo = 1u << h, nhi = 1u << (log_n - h);
  for (size_t b0 = 0; b0 < missing.size(); b0 += POW_BATCH) {
    const uint32_t cnt = (uint32_t)std::min<size_t>(POW_BATCH, missing.size() - b0);
    PowBatch pb;
    PowTab tabs[POW_BATCH];
    for (uint32_t j = 0; j < cnt; j++) {
      PowTab& t = tabs[j];
      t.full = false;
      t.h = h;
      t.bytes = ((size_t)nlo + nhi) * 32;
      t.lo = table_alloc(t.bytes);
      if (!t.lo) {
        for (uint32_t q = 0; q < j; q++) table_release(tabs[q].lo, tabs[q].bytes);
        return H2MI_ENOMEM;
      }
      t.hi = t.lo + nlo;
      pb.lo[j] = t.lo;
      pb.hi[j] = t.hi;
      pb.base[j] = host_fe(bases + 4 * missing[b0 + j]);
    }
    for (uint32_t j = cnt; j < POW_BATCH; j++) { pb.lo[j] = pb.lo[0]; pb.hi[j] = pb.hi[0]; pb.base[j] = pb.base[0]; }
    H2_LAUNCH("k_pow_table", k_pow_table2_b, dim3(ceil_div_u32(std::max(nlo, nhi), 256), 2, cnt), 256, 0, s, pb, h, log_n - h);
    for (uint32_t j = 0; j < cnt; j++) {
      PowTab& t = tabs[j];
      H2_HIP(t.built.mark(s));
      t.last_use = g_epoch;
      g_powtab_bytes += t.bytes;
      Key k;
      memcpy(k.w, bases + 4 * missing[b0 + j], 32);
      k.log_n = log_n;
      k.full = false;
      g_powtabs[k] = t;
    }
  }
  for (size_t i = 0; i < m; i++) {  // the freshly built ones (and duplicates of them)
    Key k;
    memcpy(k.w, bases + 4 * i, 32);
    k.log_n = log_n;
    k.full = false;
    auto it = g_powtabs.find(k);
    if (it == g_powtabs.end()) return H2MI_EHIP;
    H2_HIP(it->second.built.use(s));
    out[i] = it->second;
  }
  return H2MI_OK;
}

static void choose_split(uint32_t log_n, Plan* pl) {
  static const uint32_t MAXM = ab_env("H2MI_NTT_MAXM") ? (uint32_t)atoi(ab_env("H2MI_NTT_MAXM")) : 10;  // tuning knob (7 .. 10)
  if (const char* ev = ab_env("H2MI_NTT_SPLIT")) {  // tuning knob: "8,8,4" — used for the size whose log_n the parts add up to
    uint32_t a = 0, b = 0, c = 0;
    const int got = sscanf(ev, "%u,%u,%u", &a, &b, &c);
    if (got >= 2 && a + b + c == log_n && a >= 1 && b >= 1 && a <= 10 && b <= 10 && c <= 10) {
      pl->P = c ? 3 : 2;
      pl->m[0] = a; pl->m[1] = b; pl->m[2] = c;
      return;
    }
  }
  if (log_n <= MAXM) {
    pl->P = 1;
    pl->m[0] = log_n;
  } else if (log_n <= 2 * MAXM) {
    pl->P = 2;
    pl->m[0] = (log_n + 1) / 2;
    pl->m[1] = log_n - pl->m[0];
  } else {
    // three passes: 2^8-point DFTs first where the size allows (round 4, re-swept with the tile-ordered twiddles,
    // profiles/r04_ntt_sweep.txt: 2^21 (8,8,5) 269 / 264 us plain / coset against 277 / 270 for (7,7,7); 2^22 (8,8,6) 520 / 498 against
    // 524 / 502 for (8,7,7); 2^24 (8,8,8) as before; every split within +- 3 %); the last pass keeps at least 2^4 points
    pl->P = 3;
    pl->m[0] = std::max<uint32_t>(8, (log_n + 2) / 3);
    const uint32_t rem = log_n - pl->m[0];
    pl->m[1] = std::max<uint32_t>(std::min<uint32_t>(8, rem - 4), (rem + 1) / 2);
    pl->m[2] = rem - pl->m[1];
  }
}

static int get_plan(const uint64_t omega[4], uint32_t log_n, hipStream_t s, Plan* out) {
  Key k;
  memcpy(k.w, omega, 32);
  k.log_n = log_n;
  auto it = g_plans.find(k);
  if (it != g_plans.end()) {
    Plan& pl = it->second;
    pl.last_use = g_epoch;
    H2_HIP(pl.built.use(s));
    PowTab tw;  // refresh the twiddle table's pin and stream dependency
    int rc = get_powtab(omega, log_n, s, &tw, /*full=*/pl.P > 1);
    if (rc) return rc;
    *out = pl;
    return H2MI_OK;
  }
  Plan pl;
  choose_split(log_n, &pl);
  int rc = get_powtab(omega, log_n, s, &pl.tw, /*full=*/pl.P > 1);  // single-pass sizes have no inter-pass twiddles
  if (rc) return rc;
  fe w = host_fe(omega);
  for (int p = 0; p < pl.P; p++) {
    uint32_t m = pl.m[p];
    if (m == 0) continue;
    // reuse a table of the same m
    for (int q = 0; q < p; q++)
      if (pl.m[q] == m) pl.loc[p] = pl.loc[q];
    if (pl.loc[p]) continue;
    uint32_t cnt = 1u << (m - 1);
    if (hipMalloc(&pl.loc[p], (size_t)cnt * 32) != hipSuccess) {
      pl.loc[p] = nullptr;
      free_plan(pl);
      return H2MI_ENOMEM;
    }
    // w_loc = omega^(n / 2^m): order 2^m
    H2_LAUNCH("k_pow_table", k_pow_table, ceil_div_u32(cnt, 256), 256, 0, s, pl.loc[p], cnt, w, log_n - m, m - 1);
  }
  // tile-ordered twiddle matrices of the non-final passes (default tile: 2^10 elements): from the full table up to 2^22, from the
  // two-level table up to 2^24 (512 MB per 2^24 plan of 288 GB: the pass kernels then fetch the inter-pass twiddle instead of
  // multiplying two table entries per element — 11.75 -> 9.75 multiplications per element of a 2^24 transform)
  if (pl.P > 1 && log_n <= WMAT_MAX_LOG && !ab_env("H2MI_NTT_NO_WMAT")) {
    uint32_t log_seg = log_n;
    for (int p = 0; p + 1 < pl.P; p++) {
      const uint32_t m = pl.m[p], logS = log_seg - m;
      const uint32_t logC = std::min(m >= NTT_TILE_LOG ? 0u : NTT_TILE_LOG - m, logS);
      // the matrices are an accelerator, not a requirement: over budget or out of memory, older plans go first, and if that does
      // not help the pass keeps the two-level gather (ntt_dev reads wmat = nullptr as exactly that)
      const size_t wbytes = ((size_t)1 << log_seg) * 32;
      if (g_wmat_bytes + wbytes > WMAT_MAX_BYTES) {
        int rce = evict_tables();
        if (rce) { free_plan(pl); return rce; }
      }
      if (hipMalloc(&pl.wmat[p], wbytes) != hipSuccess) {
        (void)hipGetLastError();
        pl.wmat[p] = nullptr;
        int rce = evict_tables();
        if (rce) { free_plan(pl); return rce; }
        if (hipMalloc(&pl.wmat[p], wbytes) != hipSuccess) {
          (void)hipGetLastError();
          pl.wmat[p] = nullptr;
          log_seg -= m;
          continue;
        }
      }
      pl.wmat_bytes += wbytes;
      g_wmat_bytes += wbytes;
      pl.wlogC[p] = logC;
      H2_LAUNCH("k_ntt_wmat_build", k_ntt_wmat_build, ceil_div_u32((size_t)1 << log_seg, 256), 256, 0, s, (const fe*)pl.tw.lo, (const fe*)pl.tw.hi, pl.tw.h,
                pl.tw.full ? 1u : 0u, pl.wmat[p], log_seg, m, logC, log_n - log_seg);
      log_seg -= m;
    }
  }
  H2_HIP(pl.built.mark(s));
  pl.last_use = g_epoch;
  g_plans[k] = pl;
  *out = pl;
  return H2MI_OK;
}

// h2mi_shutdown: plans, power tables and scratch vectors live on the device that is being torn down; a later h2mi_init
// (possibly of another device) starts from empty caches
void ntt_teardown() {
  for (auto& kv : g_plans) free_plan(kv.second);
  g_plans.clear();
  for (auto& kv : g_powtabs) {
    H2_IGNORE(hipFree(kv.second.lo));  // hi lives in the same allocation
    kv.second.built.destroy();
  }
  g_powtabs.clear();
  g_powtab_bytes = 0;
  table_pool_clear();
  for (Scratch& c : g_scratch) {
    if (c.p) H2_IGNORE(hipFree(c.p));
    if (c.event) H2_IGNORE(hipEventDestroy(c.event));
    c = Scratch();
  }
  g_cur = nullptr;
  g_tmp = nullptr;
}

int ensure_tmp(size_t elems, hipStream_t s) {
  Scratch* e = nullptr;
  for (Scratch& c : g_scratch)
    if (c.used && c.stream == s) { e = &c; break; }
  if (!e)
    for (Scratch& c : g_scratch)
      if (!c.used) { e = &c; break; }
  if (!e) {  // more streams than scratches: take over the least recently used one, behind its last user
    e = &g_scratch[0];
    for (Scratch& c : g_scratch)
      if (c.last < e->last) e = &c;
    H2_HIP(hipStreamWaitEvent(s, e->event, 0));
  }
  if (e->elems < elems) {
    if (e->p) {
      H2_HIP(hipDeviceSynchronize());
      H2_HIP(hipFree(e->p));
      e->p = nullptr;
      e->elems = 0;
    }
    hipError_t err = hipMalloc(&e->p, elems * 32);
    if (err == hipErrorOutOfMemory) return H2MI_ENOMEM;
    H2_HIP(err);
    e->elems = elems;
  }
  e->last = ++g_scratch_clock;
  g_cur = e;
  g_tmp = e->p;
  return H2MI_OK;
}
// call after the last kernel that touches g_tmp has been queued on `s`
int release_tmp(hipStream_t s) {
  Scratch* e = g_cur;
  if (!e) return H2MI_EHIP;
  if (!e->event) H2_HIP(hipEventCreateWithFlags(&e->event, hipEventDisableTiming));
  H2_HIP(hipEventRecord(e->event, s));
  e->stream = s;
  e->used = true;
  return H2MI_OK;
}

static uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* v = ab_env(name);  // tile geometry experiments: -DH2MI_AB builds only
  return v ? (uint32_t)atoi(v) : dflt;
}

// d_src == d_a: in place.  Otherwise the first pass reads d_src (src_len elements, zero beyond) and the last
// pass writes d_a; d_src is left untouched.
static int ntt_dev(fe* d_a, uint32_t log_n, const uint64_t omega[4], const uint64_t* pre, const uint64_t* post, hipStream_t s,
                   const fe* d_src = nullptr, size_t src_len = 0) {
  if (log_n > H2MI_MAX_LOG_N) return H2MI_ERANGE;
  if (!d_src) {
    d_src = d_a;
    src_len = (size_t)1 << log_n;
  }
  Plan pl;
  int rc = get_plan(omega, log_n, s, &pl);
  if (rc) return rc;
  PowTab pt;
  if (pre) {
    rc = get_powtab(pre, log_n, s, &pt, /*full=*/true);
    if (rc) return rc;
  }
  const size_t n = (size_t)1 << log_n;
  if (pl.P > 1) {
    rc = ensure_tmp(n, s);
    if (rc) return rc;
  }
  const uint32_t tile_elems_log = env_u32("H2MI_NTT_TILE_LOG", NTT_TILE_LOG);  // elements staged per block
  const uint32_t remap = env_u32("H2MI_NTT_XCD_REMAP", 1);
  uint32_t nthreads = env_u32("H2MI_NTT_THREADS", 256);
  if (nthreads != 64 && nthreads != 128 && nthreads != 256 && nthreads != 512) nthreads = 256;
  static bool attr_set = false;
  if (!attr_set) {  // tiles above 64 KiB of LDS need the opt-in
#define H2_NTT_ATTR(DS, M)                                                                                                                         \
  H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass_col<DS, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));       \
  H2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ntt_pass_row<DS, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    H2_NTT_ATTR(0, 0);
    H2_NTT_ATTR(1024, 4);
    H2_NTT_ATTR(1024, 5);
    H2_NTT_ATTR(1024, 6);
    H2_NTT_ATTR(1024, 7);
    H2_NTT_ATTR(1024, 8);
    H2_NTT_ATTR(1024, 9);
    H2_NTT_ATTR(1024, 10);
#undef H2_NTT_ATTR
    attr_set = true;
  }
  static const bool fixed_geometry = !ab_env("H2MI_NTT_RUNTIME_GEOMETRY");  // A/B knob (-DH2MI_AB): the run-time kernels everywhere
// the compile-time form exists for 1024-element tiles and DFT sizes 2^7 .. 2^10 (every pass of every transform >= 2^14)
#define H2_NTT_LAUNCH(NAME, KERNEL)                                                                                       \
  do {                                                                                                                    \
    const bool fixed_ = fixed_geometry && pp.m + pp.logC == 10;                                                           \
    if (fixed_ && pp.m == 10) H2_LAUNCH(NAME, (KERNEL<1024, 10>), nblocks, nthreads, shmem, s, pp);                        \
    else if (fixed_ && pp.m == 9) H2_LAUNCH(NAME, (KERNEL<1024, 9>), nblocks, nthreads, shmem, s, pp);                     \
    else if (fixed_ && pp.m == 8) H2_LAUNCH(NAME, (KERNEL<1024, 8>), nblocks, nthreads, shmem, s, pp);                     \
    else if (fixed_ && pp.m == 7) H2_LAUNCH(NAME, (KERNEL<1024, 7>), nblocks, nthreads, shmem, s, pp);                     \
    else if (fixed_ && pp.m == 6) H2_LAUNCH(NAME, (KERNEL<1024, 6>), nblocks, nthreads, shmem, s, pp);                     \
    else if (fixed_ && pp.m == 5) H2_LAUNCH(NAME, (KERNEL<1024, 5>), nblocks, nthreads, shmem, s, pp);                     \
    else if (fixed_ && pp.m == 4) H2_LAUNCH(NAME, (KERNEL<1024, 4>), nblocks, nthreads, shmem, s, pp);                     \
    else H2_LAUNCH(NAME, (KERNEL<0, 0>), nblocks, nthreads, shmem, s, pp);                                                 \
  } while (0)
  // buffer schedule: P=1: a->a ; P=2: a->tmp, tmp->a ; P=3: a->tmp, tmp->tmp, tmp->a
  uint32_t log_seg = log_n;
  for (int p = 0; p < pl.P; p++) {
    PassParams pp;
    memset(&pp, 0, sizeof(pp));
    const bool last = (p == pl.P - 1);
    pp.in = (p == 0) ? d_src : g_tmp;
    pp.in_len = (p == 0) ? src_len : n;
    pp.out = last ? d_a : g_tmp;
    pp.log_n = log_n;
    pp.log_seg = log_seg;
    pp.m = pl.m[p];
    pp.loc = pl.loc[p];
    pp.tlo = pl.tw.lo;
    pp.thi = pl.tw.hi;
    pp.h = pl.tw.h;
    pp.tfull = pl.tw.full;
    pp.remap = remap;
    // below 2^18 a pass is a few dozen tiles and the regrouping barrier before the last round costs more than the LDS round trips
    // it saves (2^16: 36.0 -> 36.8 us fused; 2^20: 134.7 -> 129.5, 2^24: 2000 -> 1958: profiles/r04_ntt_fused_rounds.txt)
    pp.nofuse = (log_n < 18 || ab_env("H2MI_NTT_NO_FUSE")) ? 1 : 0;
    if (p == 0 && pre) {
      pp.plo = pt.lo;
      pp.phi = pt.hi;
      pp.ph = pt.h;
      pp.pfull = pt.full;
    }
    uint32_t logC = pp.m >= tile_elems_log ? 0 : tile_elems_log - pp.m;
    if (!last) {
      uint32_t logS = log_seg - pp.m;
      if (logC > logS) logC = logS;
      pp.logC = logC;
      pp.wmat = (pl.wmat[p] && pl.wlogC[p] == logC) ? pl.wmat[p] : nullptr;  // (a tile-geometry experiment falls back to the gather)
      uint32_t nblocks = (uint32_t)(n >> (pp.m + logC));
      size_t shmem = ((size_t)1 << (pp.m + logC)) * 36 + std::min<size_t>((size_t)1 << (pp.m - 1), TW_STAGED) * 32;
      H2_NTT_LAUNCH("k_ntt_pass_col", k_ntt_pass_col);
    } else {
      pp.has_post = post ? 1 : 0;
      pp.post = post ? f29_from_mont256<F9>(host_fe(post).v) : f29_zero();
      if (pl.P == 1) {
        pp.logN1 = 0;
        pp.logN2 = 0;
      } else if (pl.P == 2) {
        pp.logN1 = pl.m[0];
        pp.logN2 = 0;
      } else {
        pp.logN1 = pl.m[0];
        pp.logN2 = pl.m[1];
      }
      if (logC > pp.logN1) logC = pp.logN1;
      pp.logC = logC;
      uint32_t nblocks = (uint32_t)(n >> (pp.m + logC));
      size_t shmem = ((size_t)1 << (pp.m + logC)) * 36 + (pp.m ? std::min<size_t>((size_t)1 << (pp.m - 1), TW_STAGED) : 1) * 32;
      H2_NTT_LAUNCH("k_ntt_pass_row", k_ntt_pass_row);
    }
    log_seg -= pp.m;
  }
  if (pl.P > 1) return release_tmp(s);
  return H2MI_OK;
}

}  // namespace h2

using namespace h2;

extern "C" {

int h2mi_ntt_bn254_fr_dev(void* d_a, uint32_t log_n, const uint64_t omega[4], const uint64_t* pre, const uint64_t* post,
                          h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_a || !omega) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  return ntt_dev((fe*)d_a, log_n, omega, pre, post, pick_stream(stream));
}

int h2mi_ntt_bn254_fr_oop_dev(const void* d_src, size_t src_len, void* d_dst, uint32_t log_n, const uint64_t omega[4], const uint64_t* pre,
                              const uint64_t* post, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_src || !d_dst || !omega || d_src == d_dst) return H2MI_EINVAL;
  if (log_n > H2MI_MAX_LOG_N || src_len > ((size_t)1 << log_n)) return H2MI_ERANGE;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  return ntt_dev((fe*)d_dst, log_n, omega, pre, post, pick_stream(stream), (const fe*)d_src, src_len);
}

int h2mi_ntt_ext_bn254_fr(uint64_t* a, uint32_t log_n, const uint64_t omega[4], const uint64_t* pre, const uint64_t* post) {
  H2_REQUIRE_INIT();
  if (!a || !omega) return H2MI_EINVAL;
  if (log_n > H2MI_MAX_LOG_N) return H2MI_ERANGE;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = ctx().stream;
  const size_t bytes = ((size_t)1 << log_n) * 32;
  // device staging kept between calls (grow-only, like the ping-pong scratch): EvaluationDomain calls this
  // dozens of times per proof with two sizes
  static fe* stage = nullptr;
  static size_t stage_bytes = 0;
  if (stage_bytes < bytes) {
    if (stage) {
      H2_HIP(hipStreamSynchronize(s));
      H2_IGNORE(hipFree(stage));
      stage = nullptr;
      stage_bytes = 0;
    }
    hipError_t e = hipMalloc(&stage, bytes);
    if (e == hipErrorOutOfMemory) return H2MI_ENOMEM;
    H2_HIP(e);
    stage_bytes = bytes;
  }
  fe* d = stage;
  int rc = H2MI_OK;
  if (hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, s) != hipSuccess) rc = H2MI_EHIP;
  if (!rc) rc = ntt_dev(d, log_n, omega, pre, post, s);
  if (!rc && hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) rc = H2MI_EHIP;
  if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = H2MI_EHIP;
  return rc;
}

int h2mi_ntt_bn254_fr(uint64_t* a, const uint64_t omega[4], uint32_t log_n) {
  return h2mi_ntt_ext_bn254_fr(a, log_n, omega, nullptr, nullptr);
}

int h2mi_fr_scale_powers_dev(void* d_a, size_t n, const uint64_t base[4], const uint64_t* post, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_a || !base || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  uint32_t log_n = 0;
  while (((size_t)1 << log_n) < n) log_n++;
  if (log_n > 31) return H2MI_ERANGE;
  PowTab pt;
  int rc = get_powtab(base, log_n, s, &pt);
  if (rc) return rc;
  f29 p = post ? f29_from_mont256<F9>(host_fe(post).v) : f29_zero();
  H2_LAUNCH("k_scale_powers", k_scale_powers, ceil_div_u32(n, 256), 256, 0, s, (fe*)d_a, n, (const fe*)pt.lo, (const fe*)pt.hi, pt.h,
            post ? 1 : 0, p);
  return H2MI_OK;
}

int h2mi_fr_powers_dev(void* d_out, size_t n, const uint64_t base[4], h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_out || !base || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  uint32_t log_n = 0;
  while (((size_t)1 << log_n) < n) log_n++;
  if (log_n > 31) return H2MI_ERANGE;
  PowTab pt;
  int rc = get_powtab(base, log_n, s, &pt);
  if (rc) return rc;
  H2_LAUNCH("k_powers", k_powers, ceil_div_u32(n, 256), 256, 0, s, (fe*)d_out, n, (const fe*)pt.lo, (const fe*)pt.hi, pt.h);
  return H2MI_OK;
}


}  // extern "C"
