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

#include "f29.cuh"
#include "fp.cuh"
#include "h2mi_internal.h"
#include "scan.cuh"

namespace h2 {

using Fr = FrP;
using F9 = Fr29;

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
__device__ __forceinline__ f29 load_unpack(const fe* p) {
  fe x = fe_load(p);
  return f29_unpack(x.v);
}
__device__ __forceinline__ void pack_store(fe* p, const f29& a_lt2p) {
  fe o;
  f29_pack(f29_reduce_canonical<F9>(a_lt2p), o.v);
  fe_store(p, o);
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

__device__ __forceinline__ f29 pow2tab(const fe* lo, const fe* hi, uint32_t h, uint32_t e) {  // Mont261
  return f29_mul<F9>(load_unpack(&hi[e >> h]), load_unpack(&lo[e & ((1u << h) - 1)]));
}
__device__ __forceinline__ f29 powtab(const fe* lo, const fe* hi, uint32_t h, uint32_t full, uint32_t e) {
  return full ? load_unpack(&lo[e]) : pow2tab(lo, hi, h, e);
}

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
struct PowTab {  // base^i split as hi/lo for i < 2^log_n; full: lo holds every power (h = log_n, hi = {1})
  fe* lo = nullptr;
  fe* hi = nullptr;
  uint32_t h = 0;
  bool full = false;
  size_t bytes = 0;
  Built built;            // build kernels' completion: consumers on other streams wait for it
  uint64_t last_use = 0;  // call epoch of the last user (entries of the running call are never evicted)
};
struct Plan {
  int P = 0;
  uint32_t m[3] = {0, 0, 0};
  PowTab tw;
  fe* loc[3] = {nullptr, nullptr, nullptr};
  fe* wmat[3] = {nullptr, nullptr, nullptr};  // tile-ordered inter-pass twiddles of the non-final passes (k_ntt_wmat_build)
  uint32_t wlogC[3] = {0, 0, 0};              // the tile geometry each was built for
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

struct PermArgs {
  const fe* value[8];
  const fe* sigma[8];
  fe beta_delta[8];  // beta * delta^(column index), Mont256
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
  fe* z[8];
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
  const fe* gate_a[4];
  const fe* gate_q[4];
  const fe* perm_value[8];
  const fe* perm_sigma[8];
  const fe* perm_z[8];
  const fe* lk_in[2];
  const fe* lk_in_b[2];  // optional second factor of the input expression (selector * advice)
  const fe* lk_table[2];
  const fe* lk_pin[2];
  const fe* lk_ptab[2];
  const fe* lk_z[2];
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

static fe host_fe(const uint64_t w[4]) {
  fe r;
  memcpy(r.v, w, 32);
  return r;
}

// host-side Fr arithmetic for the handful of per-call constants of evaluate_h (HConsts): Montgomery-2^256 words in,
// canonical words out, through the same f29 layer the device uses (plain C++ there)
static fe h_canon(const f29& x_lt2p) {
  fe o;
  f29_pack(f29_reduce_canonical<F9>(x_lt2p), o.v);
  return o;
}
static fe h_mul256(const fe& a, const fe& b) {  // (a 2^256, b 2^256) -> a b 2^256
  return h_canon(f29_mul<F9>(f29_from_mont256<F9>(a.v), f29_unpack(b.v)));
}
static fe h_level(const fe& a, int level) {  // a 2^256 -> a 2^256 2^(-5 level)
  fe r = a;
  const fe up = h_canon(f29_mul<F9>(f29_const<F9>(F9::ONE), f29_const<F9>(F9::ONE)));   // 2^261: times it = * 2^5 in the 2^256 domain
  // 2^-5 in the 2^256 domain = the memory word 2^251 = mul(2^256, 2^256) in f29 (divides by 2^261)
  const fe down = h_canon(f29_mul<F9>(f29_const<F9>(F9::TO256), f29_const<F9>(F9::TO256)));
  for (int i = 0; i < (level < 0 ? -level : level); i++) r = h_mul256(r, level < 0 ? up : down);
  return r;
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
static uint64_t g_epoch = 1, g_evictions = 0;
struct CallScope {  // one per ABI call that uses cached tables
  CallScope() { g_epoch++; }
};

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
  if (g_powtabs.size() > POWTAB_KEEP_ENTRIES || g_powtab_bytes > POWTAB_KEEP_BYTES || g_plans.size() > PLAN_KEEP) {
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
      if (g_powtabs.size() <= POWTAB_KEEP_ENTRIES && g_powtab_bytes <= POWTAB_KEEP_BYTES && g_plans.size() <= PLAN_KEEP) break;
    }
  }
  g_evictions++;
  return H2MI_OK;
}

static int get_powtab(const uint64_t base[4], uint32_t log_n, hipStream_t s, PowTab* out, bool full = false) {
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
static int get_powtabs(const uint64_t* bases /* m x 4 */, size_t m, uint32_t log_n, hipStream_t s, PowTab* out) {
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
      if (hipMalloc(&pl.wmat[p], ((size_t)1 << log_seg) * 32) != hipSuccess) {
        pl.wmat[p] = nullptr;
        free_plan(pl);
        return H2MI_ENOMEM;
      }
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

static int ensure_tmp(size_t elems, hipStream_t s) {
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
static int release_tmp(hipStream_t s) {
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
  H2_LAUNCH("k_eval_poly", k_eval_poly, dim3(nblocks, (uint32_t)count), 256, 0, s, pl, n, logT, yp, (const fe*)pt.lo, (const fe*)pt.hi, pt.h, g_tmp);
  H2_LAUNCH("k_sum_fe", k_sum_fe, (uint32_t)count, 256, 0, s, (const fe*)g_tmp, nblocks, (fe*)d_out);
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
  H2_LAUNCH("k_eval_poly", k_eval_poly_multi, dim3(nblocks, (uint32_t)total), 256, 0, s, em, n, logT, pt[0].h, g_tmp);
  H2_LAUNCH("k_sum_fe", k_sum_fe, (uint32_t)total, 256, 0, s, (const fe*)g_tmp, nblocks, (fe*)d_out);
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
  fe* local = g_tmp;
  fe* totals = g_tmp + m * n;
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
  if (!d_values || !d_sigmas || !beta || !gamma || !beta_delta_pows || !omega || !d_z || m == 0 || m > 8) return H2MI_EINVAL;
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
  fe* num = g_tmp;
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
  if (!d_values || !d_sigmas || !beta || !gamma || !beta_delta_pows || !omega || !d_z || m == 0 || m > 8 || chunk_len == 0) return H2MI_EINVAL;
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
              (const fe*)pw.hi, pw.h, d_active, g_tmp);
    H2_LAUNCH("k_perm_write_sets", k_perm_write_sets, wgrid, 256, 0, s, (const fe*)g_tmp, usable_rows, zo, d_active, n_active);
    return release_tmp(s);
  }
  const size_t total = d_active ? (size_t)n_active : (size_t)sets * usable_rows;  // elements the scans run over
  const uint32_t nblocks = ceil_div_u32(total, MS_TILE);
  rc = ensure_tmp(3 * total + 2 * (size_t)nblocks + 2, s);
  if (rc) return rc;
  fe* num = g_tmp;
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
  uint32_t* flag = reinterpret_cast<uint32_t*>(g_tmp + 3 * u + 2 * (size_t)nblocks_dense + 2);
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
  fe* num = g_tmp;
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
  if (c->n_gates == 0 || c->n_gates > 4 || c->n_perm > 8 || c->n_lookups > 2) return H2MI_EINVAL;
  if (c->n_perm && (c->chunk_len == 0 || c->chunk_len > 3)) return H2MI_EINVAL;
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
