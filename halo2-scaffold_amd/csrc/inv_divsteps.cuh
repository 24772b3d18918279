// Modular inversion by Bernstein - Yang division steps ("Fast constant-time gcd computation and modular inversion", 2019), 32-bit form:
// batches of 30 steps on the low words of f and g give a 2 x 2 transition matrix (entries of magnitude <= 2^30) that is applied to the
// full f, g (nine signed 30-bit limbs, exactly divisible by 2^30) and to the cofactors d, e modulo p (kept in [0, p): a negative
// matrix entry takes p - operand, one Montgomery step clears the low 30 bits).  Branch-free inside a batch; ~20 batches for a 254-bit
// modulus: ~15 k simple instructions against ~50 k for the shift / subtract Euclid of fe_inv_gcd, which is what a LONE wavefront on a
// prover's critical path pays for (the sparse grand product's one inversion: 110 us -> see DESIGN.md 4.5).
// Plain C++ (no intrinsics): the same code is checked on the host against pow(x, -1, p) (tests/test_host.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define H2_INV_HD __host__ __device__ inline
#else
#define H2_INV_HD inline
#endif

namespace h2 {

// out = a^-1 mod p for plain 256-bit integers (8 x 32-bit words), 0 < a < p, p odd and below 2^255; pinv_neg = -p^-1 mod 2^32.
// Returns false when the steps did not end in f = +-1 (gcd(a, p) != 1 or a = 0): out is then untouched.
H2_INV_HD bool inv_divsteps_256(const uint32_t a[8], const uint32_t p[8], uint32_t pinv_neg, uint32_t out[8]) {
  constexpr uint32_t M30 = (1u << 30) - 1;
  int32_t f[9], g[9];
  {
    // 8 x 32 -> 9 x 30
    #pragma unroll
    for (int i = 0; i < 9; i++) {
      const int bit = 30 * i, j = bit >> 5, s = bit & 31;
      uint32_t lo = p[j] >> s, lg = a[j] >> s;
      if (s > 2 && j + 1 < 8) {
        lo |= p[j + 1] << (32 - s);
        lg |= a[j + 1] << (32 - s);
      }
      f[i] = (int32_t)(i < 8 ? (lo & M30) : lo);
      g[i] = (int32_t)(i < 8 ? (lg & M30) : lg);
    }
  }
  uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e[8] = {1, 0, 0, 0, 0, 0, 0, 0};  // f = d a, g = e a (mod p), in [0, p)
  int32_t eta = -1;                                                            // eta = -delta
  #pragma unroll 1
  for (int batch = 0; batch < 26; batch++) {
    uint32_t gz = 0;
    #pragma unroll
    for (int i = 0; i < 9; i++) gz |= (uint32_t)g[i];
    if (gz == 0) break;
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t fl = (uint32_t)f[0] | ((uint32_t)f[1] << 30), gl = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
    #pragma unroll 1
    for (int i = 0; i < 30; i++) {
      uint32_t c1 = (uint32_t)(eta >> 31), c2 = 0u - (gl & 1u);
      const uint32_t x = (fl ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
      gl += x & c2; q += y & c2; r += z & c2;
      c1 &= c2;
      eta = (int32_t)(((uint32_t)eta ^ c1) - (c1 + 1u));
      fl += gl & c1; u += q & c1; v += r & c1;
      gl >>= 1; u <<= 1; v <<= 1;
    }
    const int32_t su = (int32_t)u, sv = (int32_t)v, sq = (int32_t)q, sr = (int32_t)r;
    // f, g <- (u f + v g, q f + r g) / 2^30
    {
      int64_t cf = (int64_t)su * f[0] + (int64_t)sv * g[0], cg = (int64_t)sq * f[0] + (int64_t)sr * g[0];
      cf >>= 30; cg >>= 30;
      #pragma unroll
      for (int i = 1; i < 9; i++) {
        cf += (int64_t)su * f[i] + (int64_t)sv * g[i];
        cg += (int64_t)sq * f[i] + (int64_t)sr * g[i];
        f[i - 1] = (int32_t)((uint32_t)cf & M30); cf >>= 30;
        g[i - 1] = (int32_t)((uint32_t)cg & M30); cg >>= 30;
      }
      f[8] = (int32_t)cf;
      g[8] = (int32_t)cg;
    }
    // d, e <- (u d + v e, q d + r e) / 2^30 mod p
    uint32_t nd[8], ne[8];
    #pragma unroll
    for (int which = 0; which < 2; which++) {
      const int32_t cu = which ? sq : su, cv = which ? sr : sv;
      uint32_t xx[8], yy[8];
      #pragma unroll
      for (int t = 0; t < 2; t++) {
        const bool neg = (t ? cv : cu) < 0;
        const uint32_t* src = t ? e : d;
        uint32_t* dst = t ? yy : xx;
        uint32_t nz = 0;
        #pragma unroll
        for (int i = 0; i < 8; i++) nz |= src[i];
        if (!neg || nz == 0) {
          #pragma unroll
          for (int i = 0; i < 8; i++) dst[i] = src[i];
        } else {  // p - src
          uint32_t bo = 0;
          #pragma unroll
          for (int i = 0; i < 8; i++) {
            const uint64_t df = (uint64_t)p[i] - src[i] - bo;
            dst[i] = (uint32_t)df;
            bo = (uint32_t)(df >> 63);
          }
        }
      }
      const uint32_t mu = (uint32_t)(cu < 0 ? -cu : cu), mv = (uint32_t)(cv < 0 ? -cv : cv);
      uint32_t S[9];
      uint64_t c = 0;
      #pragma unroll
      for (int i = 0; i < 8; i++) {  // mu xx + mv yy: two products below 2^62 and a carry below 2^33
        const uint64_t lo = (uint64_t)mu * xx[i], hi = (uint64_t)mv * yy[i];
        c += (lo & 0xffffffffu) + (hi & 0xffffffffu);
        S[i] = (uint32_t)c;
        c = (c >> 32) + (lo >> 32) + (hi >> 32);
      }
      S[8] = (uint32_t)c;
      const uint32_t m = (S[0] * pinv_neg) & M30;
      uint32_t T[9];
      c = 0;
      #pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint64_t pr = (uint64_t)m * p[i];
        c += (uint64_t)S[i] + (pr & 0xffffffffu);
        T[i] = (uint32_t)c;
        c = (c >> 32) + (pr >> 32);
      }
      c += S[8];
      T[8] = (uint32_t)c;
      uint32_t rr[8];
      #pragma unroll
      for (int i = 0; i < 8; i++) rr[i] = (T[i] >> 30) | (T[i + 1] << 2);  // / 2^30: below 3 p < 2^256
      #pragma unroll
      for (int round = 0; round < 2; round++) {
        uint32_t diff[8], bo = 0;
        #pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint64_t df = (uint64_t)rr[i] - p[i] - bo;
          diff[i] = (uint32_t)df;
          bo = (uint32_t)(df >> 63);
        }
        if (!bo)
          #pragma unroll
          for (int i = 0; i < 8; i++) rr[i] = diff[i];
      }
      uint32_t* dst = which ? ne : nd;
      #pragma unroll
      for (int i = 0; i < 8; i++) dst[i] = rr[i];
    }
    #pragma unroll
    for (int i = 0; i < 8; i++) {
      d[i] = nd[i];
      e[i] = ne[i];
    }
  }
  uint32_t gz = 0;
  #pragma unroll
  for (int i = 0; i < 9; i++) gz |= (uint32_t)g[i];
  if (gz != 0) return false;
  bool plus = f[0] == 1, minus = f[0] == (int32_t)M30;
  #pragma unroll
  for (int i = 1; i < 8; i++) {
    plus = plus && f[i] == 0;
    minus = minus && f[i] == (int32_t)M30;
  }
  plus = plus && f[8] == 0;
  minus = minus && f[8] == -1;
  if (!plus && !minus) return false;
  if (plus) {
    #pragma unroll
    for (int i = 0; i < 8; i++) out[i] = d[i];
  } else {
    uint32_t nz = 0, bo = 0;
    #pragma unroll
    for (int i = 0; i < 8; i++) nz |= d[i];
    #pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t df = (uint64_t)p[i] - d[i] - bo;
      out[i] = nz ? (uint32_t)df : 0u;
      bo = (uint32_t)(df >> 63);
    }
  }
  return true;
}

}  // namespace h2
