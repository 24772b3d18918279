// Lazy 29-bit-limb Montgomery arithmetic for the BN254 fields on gfx950 — the fast inner representation.
//
// Why: gfx950 multiplies 32x32 -> 64 with v_mad_u64_u32 (half rate) and has no carry-in on it, so a
// 32-bit-limb Montgomery product needs one carry capture per partial product (136 mad + 136 addc).
// With 9 limbs of 29 bits every column sum of <= 18 partial products (< 2^58 each) fits a 64-bit
// accumulator: 162 mads and NO carry handling, and additions/subtractions are limb-wise (no carry
// chains, no conditional subtracts).  Measured: ~1.5x the multiplication throughput of fp.cuh.
//
// Representation: x = sum v[i] * 2^(29 i), i < 9 (261 bits).  Montgomery radix is 2^261.
// "normalized" = limbs 0..7 < 2^29 (limb 8 holds the rest); "lazy" = limbs may exceed 29 bits after
// limb-wise add/sub.  Values are kept only loosely reduced ([0, ~8p)); p / 2^261 = 0.0059, so
// mul(A, B) < (1 + 0.0059 * a * b) p for A < a p, B < b p — inputs up to ~8p give outputs < 1.4p.
// Multiplication contract: limbs(a) < 1.9 * 2^30 and limbs(b) < 2^29 (b normalized), or both < 2^29.
// This header is plain C++ (no intrinsics) so the same code is unit-tested on the host
// (tests/test_f29_host.py) before it runs on the GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define H2_HD __host__ __device__ __forceinline__
#else
#define H2_HD inline
#endif

namespace h2 {

#include "f29_consts.inc"

constexpr uint32_t M29 = (1u << 29) - 1;

struct f29 {
  uint32_t v[9];
};

template <class F>
H2_HD f29 f29_const(const uint32_t (&c)[9]) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = c[i];
  return r;
}
H2_HD f29 f29_zero() {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = 0;
  return r;
}
H2_HD bool f29_limbs_zero(const f29& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= a.v[i];
  return o == 0;
}

// 8 x 32-bit words (a 256-bit integer) -> 9 x 29-bit limbs, normalized
H2_HD f29 f29_unpack(const uint32_t w[8]) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, j = bit >> 5, s = bit & 31;
    uint32_t lo = w[j] >> s;
    if (s > 3 && j + 1 < 8) lo |= w[j + 1] << (32 - s);  // limb straddles two words
    r.v[i] = (i < 8) ? (lo & M29) : lo;                   // limb 8 = bits 232..255
  }
  return r;
}
// normalized limbs with value < 2^256 -> 8 x 32-bit words
H2_HD void f29_pack(const f29& a, uint32_t w[8]) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int bit = 32 * j, i = bit / 29, s = bit - 29 * i;  // word j starts inside limb i at bit s
    uint32_t x = a.v[i] >> s;                                 // 29 - s bits
    x |= a.v[i + 1] << (29 - s);                              // next limb
    if (29 - s + 29 < 32 && i + 2 < 9) x |= a.v[i + 2] << (58 - s);
    w[j] = x;
  }
}

// carry-propagate: afterwards limbs 0..7 < 2^29 (limb 8 absorbs the top)
H2_HD f29 f29_normalize(const f29& a) {
  f29 r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t x = a.v[i] + c;  // a.v[i] < 2^32 - 8 is required (all call sites stay < 2^31 + 2^30)
    r.v[i] = x & M29;
    c = x >> 29;
  }
  r.v[8] = a.v[8] + c;
  return r;
}

H2_HD f29 f29_add(const f29& a, const f29& b) {  // lazy
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// a - b + K, K = k*p in biased-limb form (F::K2.. / F::KW4..): never underflows while limbs(b) stay
// below the bias and value(b) < k*p - 2^232
H2_HD f29 f29_sub(const f29& a, const f29& b, const uint32_t (&K)[9]) {  // lazy
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + K[i] - b.v[i];
  return r;
}
H2_HD f29 f29_dbl(const f29& a) {  // lazy
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] << 1;
  return r;
}

// acc += x * y as ONE v_mad_u64_u32 whose 64-bit addend is the running column accumulator.  Written
// as inline asm for the first product of every column: left to itself hipcc starts each column's chain
// from zero and merges the carried-in accumulator with an extra v_lshl_add_u64 (17 per multiplication).
H2_HD void f29_mac_first(uint64_t& acc, uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
#else
  acc += (uint64_t)x * y;
#endif
}

// Montgomery product a * b / 2^261 mod p (loosely reduced), output normalized.
template <class F>
H2_HD f29 f29_mul(const f29& a, const f29& b) {
  uint32_t m[9];
  f29 t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    f29_mac_first(acc, a.v[0], b.v[k]);
#pragma unroll
    for (int j = 1; j <= k; j++) acc += (uint64_t)a.v[j] * b.v[k - j];
#pragma unroll
    for (int j = 0; j < k; j++) acc += (uint64_t)m[j] * F::P[k - j];
    m[k] = ((uint32_t)acc * F::INV) & M29;
    acc += (uint64_t)m[k] * F::P[0];  // low 29 bits become 0
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
    f29_mac_first(acc, a.v[k - 8], b.v[8]);
#pragma unroll
    for (int j = k - 7; j < 9; j++) acc += (uint64_t)a.v[j] * b.v[k - j];
#pragma unroll
    for (int j = k - 8; j < 9; j++) acc += (uint64_t)m[j] * F::P[k - j];
    t.v[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  t.v[8] = (uint32_t)acc;
  return t;
}
// (a * b + c * d) / 2^261 mod p with ONE Montgomery reduction (81 multiply-adds saved against two products and a
// limb-wise sum), output normalized, value < (a b + c d) / 2^261 + p.  Contract: b, d normalized; limbs(a) < 1.5 * 2^30,
// limbs(c) < 2^30: a column then holds at most 9 * 1.5 * 2^59 + 9 * 2^59 + 9 * 2^58 = 27 * 2^59 < 2^64.
template <class F>
H2_HD f29 f29_mul2(const f29& a, const f29& b, const f29& c, const f29& d) {
  uint32_t m[9];
  f29 t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    f29_mac_first(acc, a.v[0], b.v[k]);
#pragma unroll
    for (int j = 1; j <= k; j++) acc += (uint64_t)a.v[j] * b.v[k - j];
#pragma unroll
    for (int j = 0; j <= k; j++) acc += (uint64_t)c.v[j] * d.v[k - j];
#pragma unroll
    for (int j = 0; j < k; j++) acc += (uint64_t)m[j] * F::P[k - j];
    m[k] = ((uint32_t)acc * F::INV) & M29;
    acc += (uint64_t)m[k] * F::P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
    f29_mac_first(acc, a.v[k - 8], b.v[8]);
#pragma unroll
    for (int j = k - 7; j < 9; j++) acc += (uint64_t)a.v[j] * b.v[k - j];
#pragma unroll
    for (int j = k - 8; j < 9; j++) acc += (uint64_t)c.v[j] * d.v[k - j];
#pragma unroll
    for (int j = k - 8; j < 9; j++) acc += (uint64_t)m[j] * F::P[k - j];
    t.v[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  t.v[8] = (uint32_t)acc;
  return t;
}
// (a b + c d + e f) / 2^261 with one reduction — the dot products of the polynomial helpers (three terms of a linear combination).
// Contract: all six operands normalized (limbs < 2^29): a column holds at most 27 * 2^58 + 9 * 2^58 = 36 * 2^58 < 2^64.
template <class F>
H2_HD f29 f29_mul3(const f29& a, const f29& b, const f29& c, const f29& d, const f29& e, const f29& f) {
  uint32_t m[9];
  f29 t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    f29_mac_first(acc, a.v[0], b.v[k]);
#pragma unroll
    for (int j = 1; j <= k; j++) acc += (uint64_t)a.v[j] * b.v[k - j];
#pragma unroll
    for (int j = 0; j <= k; j++) acc += (uint64_t)c.v[j] * d.v[k - j];
#pragma unroll
    for (int j = 0; j <= k; j++) acc += (uint64_t)e.v[j] * f.v[k - j];
#pragma unroll
    for (int j = 0; j < k; j++) acc += (uint64_t)m[j] * F::P[k - j];
    m[k] = ((uint32_t)acc * F::INV) & M29;
    acc += (uint64_t)m[k] * F::P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
    f29_mac_first(acc, a.v[k - 8], b.v[8]);
#pragma unroll
    for (int j = k - 7; j < 9; j++) acc += (uint64_t)a.v[j] * b.v[k - j];
#pragma unroll
    for (int j = k - 8; j < 9; j++) acc += (uint64_t)c.v[j] * d.v[k - j];
#pragma unroll
    for (int j = k - 8; j < 9; j++) acc += (uint64_t)e.v[j] * f.v[k - j];
#pragma unroll
    for (int j = k - 8; j < 9; j++) acc += (uint64_t)m[j] * F::P[k - j];
    t.v[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  t.v[8] = (uint32_t)acc;
  return t;
}
// a^2 / 2^261: the 36 cross products are taken once against 2a (limbs < 2^30) instead of twice,
// 45 + 81 multiply-adds instead of 162.  Requires a normalized (limbs < 2^29).
template <class F>
H2_HD f29 f29_sqr(const f29& a) {
  uint32_t m[9], a2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) a2[i] = a.v[i] << 1;
  f29 t;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    if (k == 0) f29_mac_first(acc, a.v[0], a.v[0]);
    else f29_mac_first(acc, a.v[0], a2[k]);
#pragma unroll
    for (int j = 1; 2 * j < k; j++) acc += (uint64_t)a.v[j] * a2[k - j];
    if ((k & 1) == 0 && k > 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
    for (int j = 0; j < k; j++) acc += (uint64_t)m[j] * F::P[k - j];
    m[k] = ((uint32_t)acc * F::INV) & M29;
    acc += (uint64_t)m[k] * F::P[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; k++) {
    f29_mac_first(acc, m[k - 8], F::P[8]);
#pragma unroll
    for (int j = k - 8; 2 * j < k; j++) acc += (uint64_t)a.v[j] * a2[k - j];
    if ((k & 1) == 0) acc += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
    for (int j = k - 7; j < 9; j++) acc += (uint64_t)m[j] * F::P[k - j];
    t.v[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  t.v[8] = (uint32_t)acc;
  return t;
}

// x (normalized, value < 2p) -> canonical [0, p)
template <class F>
H2_HD f29 f29_reduce_canonical(const f29& a) {
  // d = a - p with borrow propagation over 29-bit limbs
  f29 d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t x = a.v[i] - F::P[i] - borrow;
    borrow = (x >> 31) & 1u;  // limbs are far below 2^31, so a wrapped result has its top bit set
    d.v[i] = (i < 8) ? (x & M29) : x;
  }
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = borrow ? a.v[i] : d.v[i];
  return r;
}
// x (normalized, value < 64p) -> canonical [0, p) without a multiplication: q = floor(x / p) is estimated
// from the top limb (never above, at most two below), x - q*p is formed with 9 small products and a signed
// carry chain, two conditional subtractions finish.  ~115 simple instructions against ~300 for the
// multiplication by the Montgomery one that otherwise brings a lazily accumulated value below 2p.
template <class F>
H2_HD f29 f29_reduce_loose(const f29& a) {
  constexpr uint32_t M = (uint32_t)(((uint64_t)1 << 32) / (F::P[8] + 1));  // floor(2^32 / (P8 + 1))
  const uint32_t q = (uint32_t)(((uint64_t)a.v[8] * M) >> 32);           // <= floor(x / p), >= floor(x / p) - 2
  f29 r;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int64_t d = (int64_t)a.v[i] - (int64_t)((uint64_t)q * F::P[i]) + c;
    if (i < 8) {
      r.v[i] = (uint32_t)d & M29;
      c = d >> 29;  // arithmetic shift: floor division, so the low 29 bits above are the matching remainder
    } else {
      r.v[i] = (uint32_t)d;  // 0 <= x - q*p < 3p: non-negative and small
    }
  }
  return f29_reduce_canonical<F>(f29_reduce_canonical<F>(r));
}

// is x == 0 mod p, for normalized x with value < 2p
template <class F>
H2_HD bool f29_is_zero_mod(const f29& a) {
  if (a.v[0] != 0 && a.v[0] != F::P[0]) return false;  // all but 2^-28 of the values: two compares instead of 27 or / xor
  uint32_t z = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    z |= a.v[i];
    e |= a.v[i] ^ F::P[i];
  }
  return z == 0 || e == 0;
}

// a^(p-2) (Fermat inversion) for a normalized input of any loose value; inv(0) = 0.  Output < 1.1p.
template <class F>
H2_HD f29 f29_inv(const f29& a) {
  f29 r = f29_const<F>(F::ONE);
  for (int i = 8; i >= 0; i--) {
    // exponent p - 2 in 29-bit limbs: only limb 0 changes (P[0] is odd and > 2)
    const uint32_t e = F::P[i] - (i == 0 ? 2u : 0u);
    const int top = (i == 8) ? 21 : 28;  // p has 254 bits: limb 8 holds 22 of them
    for (int b = top; b >= 0; b--) {
      r = f29_sqr<F>(r);
      if ((e >> b) & 1u) r = f29_mul<F>(r, a);
    }
  }
  return r;
}

// Mont256 words (the ABI / memory format) <-> internal Mont261 limbs
template <class F>
H2_HD f29 f29_from_mont256(const uint32_t w[8]) {
  return f29_mul<F>(f29_unpack(w), f29_const<F>(F::TO261));
}
template <class F>
H2_HD void f29_to_mont256(const f29& a_norm, uint32_t w[8]) {  // a normalized, value < ~100p
  f29 c = f29_reduce_canonical<F>(f29_mul<F>(a_norm, f29_const<F>(F::TO256)));
  f29_pack(c, w);
}

}  // namespace h2
