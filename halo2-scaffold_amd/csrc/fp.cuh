// BN254 prime-field arithmetic for gfx950 (CDNA4), 8 x 32-bit limbs, Montgomery form R = 2^256.
//
// Replaces (on the device) halo2curves::bn256::{Fq, Fr} — reference call site src/scaffold.rs:14
// (`halo2curves::bn256::{Bn256, Fr, G1Affine}`), SURVEY.md 8a row a6.  The in-memory layout is the
// crate's: [u64; 4] little-endian limbs, Montgomery form, always fully reduced (< modulus); on a
// little-endian machine that is bit-identical to the [u32; 8] used here.
//
// gfx950 has 32-bit integer multipliers only: the inner product is v_mad_u64_u32 (32x32+64 -> 64).
// No MFMA: this is carry-chain integer work, not a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "inv_divsteps.cuh"

namespace h2 {

struct alignas(16) fe {
  uint32_t v[8];
};

// ---- moduli -------------------------------------------------------------------------------------
struct FqP {  // base field q
  static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                      0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};  // R mod q
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};  // R^2 mod q
  static constexpr uint32_t INV = 0xe4866389u;  // -q^-1 mod 2^32
};
struct FrP {  // scalar field r
  static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                      0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
  static constexpr uint32_t INV = 0xefffffffu;
};

// ---- helpers ------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ fe fe_const(const uint32_t (&c)[8]) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c[i];
  return r;
}
template <class F>
__device__ __forceinline__ fe fe_one() {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = F::ONE[i];
  return r;
}
__device__ __forceinline__ fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r;
}
__device__ __forceinline__ bool fe_is_zero(const fe& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i];
  return o == 0;
}
__device__ __forceinline__ bool fe_eq(const fe& a, const fe& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}
__device__ __forceinline__ fe fe_load(const void* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 lo = q[0], hi = q[1];
  fe r;
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}
__device__ __forceinline__ void fe_store(void* p, const fe& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
__device__ __forceinline__ fe fe_select(bool c, const fe& a, const fe& b) {  // c ? a : b
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

// raw 256-bit add / sub with carry / borrow out
__device__ __forceinline__ fe raw_add(const fe& a, const fe& b, uint32_t& carry) {
  fe r;
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  carry = c;
  return r;
}
__device__ __forceinline__ fe raw_sub(const fe& a, const fe& b, uint32_t& borrow) {
  fe r;
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(a.v[i], b.v[i], c, &c);
  borrow = c;
  return r;
}
template <class F>
__device__ __forceinline__ fe raw_sub_mod(const fe& a, uint32_t& borrow) {  // a - MOD
  fe r;
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(a.v[i], F::MOD[i], c, &c);
  borrow = c;
  return r;
}
// a in [0, 2*MOD) -> [0, MOD)
template <class F>
__device__ __forceinline__ fe fe_reduce_once(const fe& a) {
  uint32_t bo;
  fe d = raw_sub_mod<F>(a, bo);
  return fe_select(bo != 0, a, d);
}

template <class F>
__device__ __forceinline__ fe fe_add(const fe& a, const fe& b) {
  uint32_t c;
  fe s = raw_add(a, b, c);  // a + b < 2^255: no carry out of 256 bits
  return fe_reduce_once<F>(s);
}
template <class F>
__device__ __forceinline__ fe fe_dbl(const fe& a) {
  fe s;
#pragma unroll
  for (int i = 7; i > 0; i--) s.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 31);
  s.v[0] = a.v[0] << 1;
  return fe_reduce_once<F>(s);
}
template <class F>
__device__ __forceinline__ fe fe_sub(const fe& a, const fe& b) {
  uint32_t bo;
  fe d = raw_sub(a, b, bo);
  // add MOD back when the subtraction borrowed
  fe r;
  unsigned c = 0;
  const uint32_t mask = bo ? 0xffffffffu : 0u;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(d.v[i], F::MOD[i] & mask, c, &c);
  return r;
}
template <class F>
__device__ __forceinline__ fe fe_neg(const fe& a) {
  fe m = fe_const<F>(F::MOD);
  uint32_t bo;
  fe d = raw_sub(m, a, bo);
  return fe_select(fe_is_zero(a), a, d);
}

// ---- Montgomery multiplication ------------------------------------------------------------------
// Product-scanning (column-wise) Montgomery: column k sums a[j]*b[k-j] and m[j]*MOD[k-j] into a
// 64-bit accumulator (v_mad_u64_u32 with the running sum as its 64-bit addend) plus a carry word
// (v_addc_co_u32 on the mad's carry-out).  MOD < 2^254, so the loop result is < 2*MOD and one
// conditional subtract finishes.  Each column's chain is ONE asm statement (fp_mac.inc).
#include "fp_mac.inc"

template <int N>
__device__ __forceinline__ void mac_col_vv(uint64_t& acc, uint32_t& ov, const uint32_t* x, const uint32_t* y) {
  // sum_{i<N} x[i] * y[-i]   (y walks downwards)
  if constexpr (N == 1) mac_vv_1(acc, ov, x[0], y[0]);
  if constexpr (N == 2) mac_vv_2(acc, ov, x[0], y[0], x[1], y[-1]);
  if constexpr (N == 3) mac_vv_3(acc, ov, x[0], y[0], x[1], y[-1], x[2], y[-2]);
  if constexpr (N == 4) mac_vv_4(acc, ov, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3]);
  if constexpr (N == 5) mac_vv_5(acc, ov, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4]);
  if constexpr (N == 6) mac_vv_6(acc, ov, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5]);
  if constexpr (N == 7) mac_vv_7(acc, ov, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6]);
  if constexpr (N == 8)
    mac_vv_8(acc, ov, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6], x[7], y[-7]);
}
// sum_{i<N} m[J0+i] * MOD[K-J0-i]
template <class F, int N, int J0, int K>
__device__ __forceinline__ void mac_col_mod(uint64_t& acc, uint32_t& ov, const uint32_t* m) {
  if constexpr (N == 1) mac_vs_1(acc, ov, m[J0], F::MOD[K - J0]);
  if constexpr (N == 2) mac_vs_2(acc, ov, m[J0], F::MOD[K - J0], m[J0 + 1], F::MOD[K - J0 - 1]);
  if constexpr (N == 3) mac_vs_3(acc, ov, m[J0], F::MOD[K - J0], m[J0 + 1], F::MOD[K - J0 - 1], m[J0 + 2], F::MOD[K - J0 - 2]);
  if constexpr (N == 4)
    mac_vs_4(acc, ov, m[J0], F::MOD[K - J0], m[J0 + 1], F::MOD[K - J0 - 1], m[J0 + 2], F::MOD[K - J0 - 2], m[J0 + 3], F::MOD[K - J0 - 3]);
  if constexpr (N == 5)
    mac_vs_5(acc, ov, m[J0], F::MOD[K - J0], m[J0 + 1], F::MOD[K - J0 - 1], m[J0 + 2], F::MOD[K - J0 - 2], m[J0 + 3], F::MOD[K - J0 - 3],
             m[J0 + 4], F::MOD[K - J0 - 4]);
  if constexpr (N == 6)
    mac_vs_6(acc, ov, m[J0], F::MOD[K - J0], m[J0 + 1], F::MOD[K - J0 - 1], m[J0 + 2], F::MOD[K - J0 - 2], m[J0 + 3], F::MOD[K - J0 - 3],
             m[J0 + 4], F::MOD[K - J0 - 4], m[J0 + 5], F::MOD[K - J0 - 5]);
  if constexpr (N == 7)
    mac_vs_7(acc, ov, m[J0], F::MOD[K - J0], m[J0 + 1], F::MOD[K - J0 - 1], m[J0 + 2], F::MOD[K - J0 - 2], m[J0 + 3], F::MOD[K - J0 - 3],
             m[J0 + 4], F::MOD[K - J0 - 4], m[J0 + 5], F::MOD[K - J0 - 5], m[J0 + 6], F::MOD[K - J0 - 6]);
  if constexpr (N == 8)
    mac_vs_8(acc, ov, m[J0], F::MOD[K - J0], m[J0 + 1], F::MOD[K - J0 - 1], m[J0 + 2], F::MOD[K - J0 - 2], m[J0 + 3], F::MOD[K - J0 - 3],
             m[J0 + 4], F::MOD[K - J0 - 4], m[J0 + 5], F::MOD[K - J0 - 5], m[J0 + 6], F::MOD[K - J0 - 6], m[J0 + 7], F::MOD[K - J0 - 7]);
}

template <class F, int K>
__device__ __forceinline__ void mont_col_lo(uint64_t& acc, uint32_t& ov, const fe& a, const fe& b, uint32_t* m) {
  // column K < 8: a[0..K]*b[K..0], m[0..K-1]*MOD[K..1], then m[K] and m[K]*MOD[0]
  mac_col_vv<K + 1>(acc, ov, &a.v[0], &b.v[K]);
  if constexpr (K > 0) mac_col_mod<F, K, 0, K>(acc, ov, m);
  m[K] = (uint32_t)acc * F::INV;
  mac_col_mod<F, 1, K, K>(acc, ov, m);  // low word becomes 0
  acc = (acc >> 32) | ((uint64_t)ov << 32);
  ov = 0;
}
template <class F, int K>
__device__ __forceinline__ void mont_col_hi(uint64_t& acc, uint32_t& ov, const fe& a, const fe& b, const uint32_t* m, uint32_t* t) {
  // column K in 8..14: a[K-7..7]*b[7..K-7], m[K-7..7]*MOD[7..K-7]
  mac_col_vv<15 - K>(acc, ov, &a.v[K - 7], &b.v[7]);
  mac_col_mod<F, 15 - K, K - 7, K>(acc, ov, m);
  t[K - 8] = (uint32_t)acc;
  acc = (acc >> 32) | ((uint64_t)ov << 32);
  ov = 0;
}

template <class F>
__device__ __forceinline__ fe fe_mul(const fe& a, const fe& b) {
  uint32_t m[8];
  uint32_t t[8];
  uint64_t acc = 0;
  uint32_t ov = 0;
  mont_col_lo<F, 0>(acc, ov, a, b, m);
  mont_col_lo<F, 1>(acc, ov, a, b, m);
  mont_col_lo<F, 2>(acc, ov, a, b, m);
  mont_col_lo<F, 3>(acc, ov, a, b, m);
  mont_col_lo<F, 4>(acc, ov, a, b, m);
  mont_col_lo<F, 5>(acc, ov, a, b, m);
  mont_col_lo<F, 6>(acc, ov, a, b, m);
  mont_col_lo<F, 7>(acc, ov, a, b, m);
  mont_col_hi<F, 8>(acc, ov, a, b, m, t);
  mont_col_hi<F, 9>(acc, ov, a, b, m, t);
  mont_col_hi<F, 10>(acc, ov, a, b, m, t);
  mont_col_hi<F, 11>(acc, ov, a, b, m, t);
  mont_col_hi<F, 12>(acc, ov, a, b, m, t);
  mont_col_hi<F, 13>(acc, ov, a, b, m, t);
  mont_col_hi<F, 14>(acc, ov, a, b, m, t);
  t[7] = (uint32_t)acc;  // column 15 has no products
  fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  return fe_reduce_once<F>(r);
}

template <class F>
__device__ __forceinline__ fe fe_sqr(const fe& a) {
  return fe_mul<F>(a, a);
}

template <class F>
__device__ __forceinline__ fe fe_from_mont(const fe& a) {  // a * R^-1 : Montgomery -> canonical
  fe one = fe_zero();
  one.v[0] = 1;
  return fe_mul<F>(a, one);
}
template <class F>
__device__ __forceinline__ fe fe_to_mont(const fe& a) {  // canonical -> Montgomery
  return fe_mul<F>(a, fe_const<F>(F::R2));
}

// a^(MOD-2) by square-and-multiply over the fixed exponent (Fermat inversion); inv(0) = 0.
template <class F>
__device__ __noinline__ fe fe_inv(const fe& a) {
  fe r = fe_one<F>();
  // exponent = MOD - 2 (MOD is odd and its low word is > 2, so only limb 0 changes)
  for (int i = 7; i >= 0; i--) {
    uint32_t e = F::MOD[i] - (i == 0 ? 2u : 0u);
    for (int b = 31; b >= 0; b--) {
      r = fe_sqr<F>(r);
      if ((e >> b) & 1u) r = fe_mul<F>(r, a);
    }
  }
  return r;
}

// a^-1 by the binary extended Euclidean algorithm: ~2 * 254 shift / subtract rounds of 256-bit integer work
// instead of ~380 dependent multiplications — for the places where ONE inversion sits on a latency-critical
// path (a lone lane: ~50 us against ~500 us for fe_inv).  Montgomery in and out; inv(0) = 0.  Not constant time.
template <class F>
__device__ __noinline__ fe fe_inv_gcd(const fe& a_mont) {
  if (fe_is_zero(a_mont)) return a_mont;
  const fe p = fe_const<F>(F::MOD);
  auto is_one = [](const fe& x) {
    uint32_t o = x.v[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 8; i++) o |= x.v[i];
    return o == 0;
  };
  auto shr1 = [](fe& x, uint32_t top) {  // x = (top : x) >> 1
#pragma unroll
    for (int i = 0; i < 7; i++) x.v[i] = (x.v[i] >> 1) | (x.v[i + 1] << 31);
    x.v[7] = (x.v[7] >> 1) | (top << 31);
  };
  auto halve_mod = [&](fe& x) {  // x / 2 mod p for x < p
    uint32_t carry = 0;
    if (x.v[0] & 1u) x = raw_add(x, p, carry);
    shr1(x, carry);
  };
  fe u = a_mont, v = p, x1 = fe_zero(), x2 = fe_zero();
  x1.v[0] = 1;
  while (!is_one(u) && !is_one(v)) {
    while (!(u.v[0] & 1u)) { shr1(u, 0); halve_mod(x1); }
    while (!(v.v[0] & 1u)) { shr1(v, 0); halve_mod(x2); }
    uint32_t borrow;
    fe d = raw_sub(u, v, borrow);
    if (!borrow) {  // u >= v
      u = d;
      x1 = fe_sub<F>(x1, x2);
    } else {
      v = raw_sub(v, u, borrow);
      x2 = fe_sub<F>(x2, x1);
    }
  }
  fe r = is_one(u) ? x1 : x2;  // (a R)^-1 as a plain integer: a^-1 R^-1
  const fe r2 = fe_const<F>(F::R2);
  return fe_mul<F>(fe_mul<F>(r, r2), r2);  // * R^2 -> a^-1 R
}

// The same inverse by division steps (inv_divsteps.cuh): a third of the instructions, no data-dependent branches inside a batch.
// For the single inversions a prover's critical path waits for (the grand products' denominators); fe_inv_gcd stays as its
// fallback and cross-check (h2mi_dbg_field_op: op 4 = Fermat, op 9 = this, op 10 = fe_inv_gcd).
template <class F>
__device__ __noinline__ fe fe_inv_ds(const fe& a_mont) {
  if (fe_is_zero(a_mont)) return a_mont;
  uint32_t mod[8];
#pragma unroll
  for (int i = 0; i < 8; i++) mod[i] = F::MOD[i];
  fe r;
  if (!inv_divsteps_256(a_mont.v, mod, F::INV, r.v)) return fe_inv_gcd<F>(a_mont);
  const fe r2 = fe_const<F>(F::R2);
  return fe_mul<F>(fe_mul<F>(r, r2), r2);  // (a R)^-1 = a^-1 R^-1 as a plain integer; * R^2 twice -> a^-1 R
}

}  // namespace h2
