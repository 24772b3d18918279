// h2mi.hpp — C++17 host layer over the C ABI (h2mi.h), mirroring the names the reference reaches through
// create_proof() so that host code and tests read like the reference's own:
//
//   h2mi::arithmetic::best_multiexp / best_fft / eval_polynomial / kate_division
//        halo2_proofs::arithmetic (un-vendored dependency of the reference, Cargo.toml:13)
//   h2mi::poly::EvaluationDomain            halo2_proofs::poly::EvaluationDomain
//   h2mi::poly::kzg::ParamsKZG              halo2_proofs::poly::kzg::commitment::ParamsKZG
//        (reference examples/standard_plonk.rs:29 `ParamsKZG::<Bn256>::setup(k, OsRng)`)
//
// Rust is absent from this image, so this header stands where the Rust shim of INTEGRATION.md would:
// same argument meaning, same failure behaviour (the crate's asserts / `.expect(..)` panics become
// h2mi::Error exceptions).  Bulk arithmetic never happens here — only the handful of per-domain scalars
// EvaluationDomain::new computes on the CPU (omega, inverses), with a small Montgomery Fr.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <array>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "h2mi.h"

namespace h2mi {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what + ": " + h2mi_strerror(c)), code(c) {}
};
inline void check(int rc, const char* what) {
  if (rc != H2MI_OK) throw Error(rc, what);
}
// one process drives one GPU: LOCAL_RANK selects it unless a device is named
inline void init(int device = -1) {
  if (device < 0) {
    const char* lr = std::getenv("LOCAL_RANK");
    device = lr ? std::atoi(lr) : 0;
  }
  check(h2mi_init(device), "h2mi_init");
}

// ---- halo2curves::bn256 value types (layouts of SURVEY.md 8a-0) ------------------------------------
struct Fr {
  uint64_t l[4];  // Montgomery form, R = 2^256, fully reduced
  bool operator==(const Fr& o) const { return std::memcmp(l, o.l, 32) == 0; }
};
struct G1Affine {
  uint64_t x[4], y[4];  // (0, 0) = identity
};
struct G1 {
  uint64_t x[4], y[4], z[4];  // Jacobian; z = 0 identity
  bool is_identity() const { return (z[0] | z[1] | z[2] | z[3]) == 0; }
};

// ---- host-side scalar-field helpers (a few scalars per domain; never bulk data) --------------------
namespace detail {
// x with a x = 1 (mod p) for an odd modulus p < 2^255 and 0 < a < p, plain integers (4 x 64-bit limbs), by Bernstein - Yang division
// steps ("Fast constant-time gcd computation and modular inversion", 2019) in batches of 62: a batch looks at the low words of f and g
// only and yields a 2 x 2 transition matrix that is then applied to the full f, g (exactly divisible by 2^62) and to the cofactors d, e
// (modulo p).  ~1.5 us against ~18 us for the Fermat exponentiation a^(p-2): the provers' hosts invert a dozen single elements per
// proof on the transcript's critical path (batch_normalize of a phase's points, the rotation sets' Lagrange denominators) — at 2^16
// rows and below that was ~0.2 ms of a proof.  Returns false if the steps did not converge (they always do for gcd(a, p) = 1).
inline bool inv_mod_odd(const uint64_t a[4], const uint64_t p[4], uint64_t out[4]) {
  typedef unsigned __int128 u128;
  typedef __int128 i128;
  constexpr int64_t M62 = (int64_t)((1ULL << 62) - 1);
  auto to62 = [&](const uint64_t x[4], int64_t y[5]) {
    y[0] = (int64_t)(x[0] & (uint64_t)M62);
    y[1] = (int64_t)(((x[0] >> 62) | (x[1] << 2)) & (uint64_t)M62);
    y[2] = (int64_t)(((x[1] >> 60) | (x[2] << 4)) & (uint64_t)M62);
    y[3] = (int64_t)(((x[2] >> 58) | (x[3] << 6)) & (uint64_t)M62);
    y[4] = (int64_t)(x[3] >> 56);
  };
  uint64_t pinv = p[0];  // p^-1 mod 2^64 by Newton's iteration (p odd: correct to 3 bits, doubling each round)
  for (int i = 0; i < 6; i++) pinv *= 2 - p[0] * pinv;
  int64_t f[5], g[5];
  to62(p, f);
  to62(a, g);
  uint64_t d[4] = {0, 0, 0, 0}, e[4] = {1, 0, 0, 0};  // f = d a, g = e a (mod p), both kept in [0, p)
  int64_t eta = -1;                                    // eta = -delta
  auto mul_add_shift = [&](int64_t cu, const uint64_t* x, int64_t cv, const uint64_t* y, uint64_t* o) {
    // o = (cu x + cv y) / 2^62 mod p for |cu|, |cv| <= 2^62 and x, y in [0, p): a negative coefficient takes p - operand, the sum
    // S < 2^63 p gets the multiple m p that clears its low 62 bits (one Montgomery step), (S + m p) / 2^62 < 3 p
    uint64_t xx[4], yy[4];
    auto cond_neg = [&](bool neg, const uint64_t* s, uint64_t* t) {
      if (!neg || (s[0] | s[1] | s[2] | s[3]) == 0) { std::memcpy(t, s, 32); return; }
      u128 bo = 0;
      for (int i = 0; i < 4; i++) {
        u128 df = (u128)p[i] - s[i] - (uint64_t)bo;
        t[i] = (uint64_t)df;
        bo = (df >> 64) & 1;
      }
    };
    cond_neg(cu < 0, x, xx);
    cond_neg(cv < 0, y, yy);
    const uint64_t mu = (uint64_t)(cu < 0 ? -cu : cu), mv = (uint64_t)(cv < 0 ? -cv : cv);
    uint64_t S[5];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      u128 lo = (u128)mu * xx[i], hi = (u128)mv * yy[i];
      c += (uint64_t)lo;
      c += (uint64_t)hi;
      S[i] = (uint64_t)c;
      c = (c >> 64) + (lo >> 64) + (hi >> 64);
    }
    S[4] = (uint64_t)c;
    const uint64_t m = (0 - S[0] * pinv) & (uint64_t)M62;
    c = 0;
    uint64_t T[5];
    for (int i = 0; i < 4; i++) {
      u128 pr = (u128)m * p[i];
      c += (u128)S[i] + (uint64_t)pr;
      T[i] = (uint64_t)c;
      c = (c >> 64) + (pr >> 64);
    }
    c += S[4];
    T[4] = (uint64_t)c;
    uint64_t r[4];
    for (int i = 0; i < 4; i++) r[i] = (T[i] >> 62) | (T[i + 1] << 2);  // / 2^62 (the value is below 3 p < 2^256)
    for (int round = 0; round < 2; round++) {
      bool ge = true;
      for (int i = 3; i >= 0; i--) {
        if (r[i] > p[i]) break;
        if (r[i] < p[i]) { ge = false; break; }
      }
      if (!ge) break;
      u128 bo = 0;
      for (int i = 0; i < 4; i++) {
        u128 df = (u128)r[i] - p[i] - (uint64_t)bo;
        r[i] = (uint64_t)df;
        bo = (df >> 64) & 1;
      }
    }
    std::memcpy(o, r, 32);
  };
  for (int batch = 0; batch < 13; batch++) {
    if ((g[0] | g[1] | g[2] | g[3] | g[4]) == 0) break;
    // 62 division steps on the low words; (u, v; q, r) scaled by 2^62
    uint64_t u = 1, v = 0, q = 0, r = 1, fl = (uint64_t)f[0] | ((uint64_t)f[1] << 62), gl = (uint64_t)g[0] | ((uint64_t)g[1] << 62);
    for (int i = 0; i < 62; i++) {
      uint64_t c1 = (uint64_t)(eta >> 63), c2 = 0 - (gl & 1);
      const uint64_t x = (fl ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
      gl += x & c2; q += y & c2; r += z & c2;
      c1 &= c2;
      eta = (int64_t)(((uint64_t)eta ^ c1) - (c1 + 1));
      fl += gl & c1; u += q & c1; v += r & c1;
      gl >>= 1; u <<= 1; v <<= 1;
    }
    const int64_t su = (int64_t)u, sv = (int64_t)v, sq = (int64_t)q, sr = (int64_t)r;
    // f, g <- (u f + v g, q f + r g) / 2^62 over signed 62-bit limbs
    i128 cf = (i128)su * f[0] + (i128)sv * g[0], cg = (i128)sq * f[0] + (i128)sr * g[0];
    cf >>= 62; cg >>= 62;
    for (int i = 1; i < 5; i++) {
      cf += (i128)su * f[i] + (i128)sv * g[i];
      cg += (i128)sq * f[i] + (i128)sr * g[i];
      f[i - 1] = (int64_t)cf & M62; cf >>= 62;
      g[i - 1] = (int64_t)cg & M62; cg >>= 62;
    }
    f[4] = (int64_t)cf;
    g[4] = (int64_t)cg;
    uint64_t nd[4], ne[4];
    mul_add_shift(su, d, sv, e, nd);
    mul_add_shift(sq, d, sr, e, ne);
    std::memcpy(d, nd, 32);
    std::memcpy(e, ne, 32);
  }
  if ((g[0] | g[1] | g[2] | g[3] | g[4]) != 0) return false;
  // f = +1 or -1 (gcd 1): the inverse is d or p - d
  const bool plus = f[0] == 1 && (f[1] | f[2] | f[3] | f[4]) == 0;
  const bool minus = f[0] == M62 && f[1] == M62 && f[2] == M62 && f[3] == M62 && f[4] == -1;
  if (!plus && !minus) return false;
  if (plus) {
    std::memcpy(out, d, 32);
  } else {
    u128 bo = 0;
    for (int i = 0; i < 4; i++) {
      u128 df = (u128)p[i] - d[i] - (uint64_t)bo;
      out[i] = (uint64_t)df;
      bo = (df >> 64) & 1;
    }
  }
  return true;
}
}  // namespace detail
namespace fr {
typedef unsigned __int128 u128;
constexpr uint64_t MODULUS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
constexpr uint64_t INV = 0xc2e1f593efffffffULL;
constexpr Fr ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
constexpr Fr R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
constexpr Fr R3 = {{0x5e94d8e1b4bf0040ULL, 0x2a489cbe1cfbb6b8ULL, 0x893cc664a19fcfedULL, 0x0cf8594b7fcc657cULL}};  // 2^768 mod r
constexpr uint32_t S = 28;

inline Fr mul(const Fr& a, const Fr& b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a.l[j] * b.l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * INV;
    c = (u128)m * MODULUS[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * MODULUS[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  bool ge = t[4] != 0;
  if (!ge) {
    ge = true;
    for (int i = 3; i >= 0; i--) {
      if (t[i] > MODULUS[i]) break;
      if (t[i] < MODULUS[i]) { ge = false; break; }
    }
  }
  if (ge) {
    u128 bo = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)t[i] - MODULUS[i] - (uint64_t)bo;
      t[i] = (uint64_t)d;
      bo = (d >> 64) & 1;
    }
  }
  Fr r;
  std::memcpy(r.l, t, 32);
  return r;
}
inline Fr add(const Fr& a, const Fr& b) {
  u128 c = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; i++) {
    c += (u128)a.l[i] + b.l[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  bool ge = true;
  for (int i = 3; i >= 0; i--) {
    if (t[i] > MODULUS[i]) break;
    if (t[i] < MODULUS[i]) { ge = false; break; }
  }
  if (ge) {
    u128 bo = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)t[i] - MODULUS[i] - (uint64_t)bo;
      t[i] = (uint64_t)d;
      bo = (d >> 64) & 1;
    }
  }
  Fr r;
  std::memcpy(r.l, t, 32);
  return r;
}
inline Fr sub(const Fr& a, const Fr& b) {
  u128 bo = 0;
  Fr d;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a.l[i] - b.l[i] - (uint64_t)bo;
    d.l[i] = (uint64_t)t;
    bo = (t >> 64) & 1;
  }
  if (bo) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)d.l[i] + MODULUS[i];
      d.l[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return d;
}
inline Fr neg(const Fr& a) { return sub(Fr{{0, 0, 0, 0}}, a); }
inline Fr from_u64(uint64_t v) {
  Fr a = {{v, 0, 0, 0}};
  return mul(a, R2);
}
inline Fr pow(const Fr& a, const uint64_t e[4]) {
  int top = 255;  // a small exponent (omega^i, delta^j) must not cost 256 squarings
  while (top >= 0 && !((e[top >> 6] >> (top & 63)) & 1)) top--;
  Fr r = ONE;
  for (int i = top; i >= 0; i--) {
    r = mul(r, r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a);
  }
  return r;
}
// Montgomery's trick: the inverses of k nonzero elements for one inversion and 3 (k - 1) multiplications
inline std::vector<Fr> batch_invert(const std::vector<Fr>& v);
inline Fr pow_u64(const Fr& a, uint64_t e) {
  uint64_t ee[4] = {e, 0, 0, 0};
  return pow(a, ee);
}
inline Fr invert_fermat(const Fr& a) {  // a^(r-2): the definition, kept as the cross-check of invert()
  uint64_t e[4] = {MODULUS[0] - 2, MODULUS[1], MODULUS[2], MODULUS[3]};
  return pow(a, e);
}
inline Fr invert(const Fr& a) {  // the crate returns CtOption, callers here never pass zero (0 -> 0, as a^(r-2) gives)
  if ((a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0) return a;
  Fr t;  // a = A R: the integer inverse is A^-1 R^-1, and (A^-1 R^-1)(R^3) R^-1 = A^-1 R
  if (!detail::inv_mod_odd(a.l, MODULUS, t.l)) return invert_fermat(a);
  return mul(t, R3);
}
inline std::vector<Fr> batch_invert(const std::vector<Fr>& v) {
  std::vector<Fr> out(v.size());
  if (v.empty()) return out;
  Fr acc = ONE;
  for (size_t i = 0; i < v.size(); i++) {
    out[i] = acc;
    acc = mul(acc, v[i]);
  }
  Fr inv = invert(acc);
  for (size_t i = v.size(); i-- > 0;) {
    out[i] = mul(out[i], inv);
    inv = mul(inv, v[i]);
  }
  return out;
}
// ROOT_OF_UNITY = 7^((r-1)/2^28); ZETA = 7^((r-1)/3)
inline Fr root_of_unity() {
  // (r - 1) >> 28
  uint64_t e[4];
  uint64_t m1[4] = {MODULUS[0] - 1, MODULUS[1], MODULUS[2], MODULUS[3]};
  for (int i = 0; i < 4; i++) e[i] = (m1[i] >> S) | (i < 3 ? m1[i + 1] << (64 - S) : 0);
  return pow(from_u64(7), e);
}
inline Fr zeta() {
  // (r - 1) / 3 by long division on 64-bit limbs
  uint64_t m1[4] = {MODULUS[0] - 1, MODULUS[1], MODULUS[2], MODULUS[3]}, e[4];
  u128 rem = 0;
  for (int i = 3; i >= 0; i--) {
    u128 cur = (rem << 64) | m1[i];
    e[i] = (uint64_t)(cur / 3);
    rem = cur % 3;
  }
  return pow(from_u64(7), e);
}
inline Fr omega_for(uint32_t k) {  // EvaluationDomain::new: square ROOT_OF_UNITY (S - k) times
  if (k > S) throw Error(H2MI_ERANGE, "omega_for");
  Fr w = root_of_unity();
  for (uint32_t i = k; i < S; i++) w = mul(w, w);
  return w;
}
}  // namespace fr

namespace arithmetic {
// best_multiexp(coeffs, bases): asserts equal lengths like the crate
inline G1 best_multiexp(const std::vector<Fr>& coeffs, const std::vector<G1Affine>& bases) {
  if (coeffs.size() != bases.size()) throw Error(H2MI_EINVAL, "best_multiexp: coeffs.len() != bases.len()");
  G1 out;
  check(h2mi_msm_bn254_g1(0, (const uint64_t*)bases.data(), (const uint64_t*)coeffs.data(), coeffs.size(), (uint64_t*)&out), "best_multiexp");
  return out;
}
inline void best_fft(std::vector<Fr>& a, const Fr& omega, uint32_t log_n) {
  if (a.size() != ((size_t)1 << log_n)) throw Error(H2MI_EINVAL, "best_fft: a.len() != 1 << log_n");
  check(h2mi_ntt_bn254_fr((uint64_t*)a.data(), omega.l, log_n), "best_fft");
}
// shorthands of the phase entry (exports of their own until round 4): a lone in-order commitment, a batch, a sparse batch
inline int msm_inorder_dev(uint64_t handle, const void* d_scalars, size_t n, void* d_out_jacobian, h2mi_stream_t stream = nullptr) {
  return h2mi_msm_bn254_g1_phase_dev(handle, &d_scalars, 1, n, d_out_jacobian, H2MI_MSM_INORDER, stream);
}
inline int msm_batch_dev(uint64_t handle, const void* const* d_scalars, size_t count, size_t n, void* d_out_jacobian, h2mi_stream_t stream = nullptr) {
  return h2mi_msm_bn254_g1_phase_dev(handle, d_scalars, count, n, d_out_jacobian, 0, stream);
}
inline int msm_batch_sparse_dev(uint64_t handle, const void* const* d_scalars, size_t count, size_t n, void* d_out_jacobian, h2mi_stream_t stream = nullptr) {
  return h2mi_msm_bn254_g1_phase_dev(handle, d_scalars, count, n, d_out_jacobian, H2MI_MSM_SPARSE, stream);
}
struct DeviceVec {  // RAII device copy of a coefficient vector
  void* p = nullptr;
  size_t n = 0;
  explicit DeviceVec(size_t count) : n(count) { check(h2mi_malloc(count * 32, &p), "h2mi_malloc"); }
  explicit DeviceVec(const std::vector<Fr>& v) : DeviceVec(v.size()) { check(h2mi_memcpy_h2d(p, v.data(), n * 32), "h2d"); }
  DeviceVec(const DeviceVec&) = delete;
  ~DeviceVec() { if (p) h2mi_free(p); }
  std::vector<Fr> download() const {
    std::vector<Fr> v(n);
    check(h2mi_memcpy_d2h(v.data(), p, n * 32), "d2h");
    return v;
  }
};
inline Fr eval_polynomial(const std::vector<Fr>& poly, const Fr& point) {
  DeviceVec d(poly), o(1);
  check(h2mi_fr_eval_poly_dev(d.p, poly.size(), point.l, o.p, nullptr), "eval_polynomial");
  return o.download()[0];
}
inline std::vector<Fr> kate_division(const std::vector<Fr>& a, const Fr& b) {
  if (a.size() < 2) return {};
  Fr zero = {{0, 0, 0, 0}};
  if (b == zero) return std::vector<Fr>(a.begin() + 1, a.end());
  DeviceVec d(a), o(a.size() - 1);
  Fr binv = fr::invert(b);
  check(h2mi_fr_kate_division_dev(d.p, a.size(), b.l, binv.l, o.p, nullptr), "kate_division");
  return o.download();
}
}  // namespace arithmetic

namespace poly {
class EvaluationDomain {
 public:
  // EvaluationDomain::new(j, k): j = constraint-system degree, k = log2(rows)
  EvaluationDomain(uint32_t j, uint32_t k) : k_(k), n_((uint64_t)1 << k), quotient_poly_degree_(j - 1) {
    extended_k_ = k;
    while (((uint64_t)1 << extended_k_) < n_ * quotient_poly_degree_) extended_k_++;
    omega_ = fr::omega_for(k);
    omega_inv_ = fr::invert(omega_);
    extended_omega_ = fr::omega_for(extended_k_);
    extended_omega_inv_ = fr::invert(extended_omega_);
    g_coset_ = fr::zeta();
    g_coset_inv_ = fr::mul(g_coset_, g_coset_);
    ifft_divisor_ = fr::invert(fr::from_u64(n_));
    extended_ifft_divisor_ = fr::invert(fr::from_u64((uint64_t)1 << extended_k_));
  }
  uint32_t k() const { return k_; }
  uint32_t extended_k() const { return extended_k_; }
  size_t extended_len() const { return (size_t)1 << extended_k_; }
  const Fr& get_omega() const { return omega_; }
  const Fr& get_omega_inv() const { return omega_inv_; }
  const Fr& get_extended_omega() const { return extended_omega_; }
  const Fr& get_extended_omega_inv() const { return extended_omega_inv_; }
  const Fr& get_g_coset() const { return g_coset_; }
  const Fr& get_g_coset_inv() const { return g_coset_inv_; }
  const Fr& get_ifft_divisor() const { return ifft_divisor_; }
  const Fr& get_extended_ifft_divisor() const { return extended_ifft_divisor_; }
  // (X^n - 1)^-1 on the extended coset: 2^(extended_k - k) distinct values (what evaluate_h divides by); computed once
  const std::vector<Fr>& t_inv() const {
    if (t_inv_.empty()) {
      std::vector<Fr> t;
      for (uint64_t i = 0; i < ((uint64_t)1 << (extended_k_ - k_)); i++) {
        Fr X = fr::mul(g_coset_, fr::pow_u64(extended_omega_, i));
        t.push_back(fr::sub(fr::pow_u64(X, n_), fr::ONE));
      }
      t_inv_ = fr::batch_invert(t);
    }
    return t_inv_;
  }

  std::vector<Fr> lagrange_to_coeff(std::vector<Fr> a) const {
    require(a.size() == n_, "lagrange_to_coeff");
    check(h2mi_ntt_ext_bn254_fr((uint64_t*)a.data(), k_, omega_inv_.l, nullptr, ifft_divisor_.l), "lagrange_to_coeff");
    return a;
  }
  std::vector<Fr> coeff_to_lagrange(std::vector<Fr> a) const {
    require(a.size() == n_, "coeff_to_lagrange");
    check(h2mi_ntt_bn254_fr((uint64_t*)a.data(), omega_.l, k_), "coeff_to_lagrange");
    return a;
  }
  std::vector<Fr> coeff_to_extended(std::vector<Fr> a) const {
    require(a.size() == n_, "coeff_to_extended");
    a.resize(extended_len(), Fr{{0, 0, 0, 0}});
    check(h2mi_ntt_ext_bn254_fr((uint64_t*)a.data(), extended_k_, extended_omega_.l, g_coset_.l, nullptr), "coeff_to_extended");
    return a;
  }
  std::vector<Fr> extended_to_coeff(const std::vector<Fr>& a) const {
    require(a.size() == extended_len(), "extended_to_coeff");
    arithmetic::DeviceVec d(a);
    check(h2mi_ntt_bn254_fr_dev(d.p, extended_k_, extended_omega_inv_.l, nullptr, nullptr, nullptr), "extended_to_coeff");
    check(h2mi_fr_scale_powers_dev(d.p, a.size(), g_coset_inv_.l, extended_ifft_divisor_.l, nullptr), "distribute_powers_zeta");
    std::vector<Fr> v = d.download();
    v.resize(n_ * quotient_poly_degree_);
    return v;
  }

 private:
  static void require(bool ok, const char* what) {
    if (!ok) throw Error(H2MI_EINVAL, std::string(what) + ": wrong vector length");
  }
  uint32_t k_, extended_k_;
  uint64_t n_, quotient_poly_degree_;
  Fr omega_, omega_inv_, extended_omega_, extended_omega_inv_, g_coset_, g_coset_inv_, ifft_divisor_, extended_ifft_divisor_;
  mutable std::vector<Fr> t_inv_;
};

namespace kzg {
class ParamsKZG {
 public:
  // ParamsKZG::setup(k, rng): the toxic-waste scalar is passed explicitly (the reference draws it from OsRng)
  static ParamsKZG setup(uint32_t k, const Fr& s) {
    ParamsKZG p(k);
    const size_t n = p.n_;
    arithmetic::DeviceVec pw(n);
    check(h2mi_fr_powers_dev(pw.p, n, s.l, nullptr), "powers of s");
    check(h2mi_malloc(n * 64, &p.d_g_), "h2mi_malloc");
    check(h2mi_g1_fixed_base_mul_dev(pw.p, n, p.d_g_, nullptr), "g");
    Fr w_inv = fr::invert(fr::omega_for(k)), n_inv = fr::invert(fr::from_u64(n));
    check(h2mi_ntt_bn254_fr_dev(pw.p, k, w_inv.l, nullptr, n_inv.l, nullptr), "lagrange scalars");
    check(h2mi_malloc(n * 64, &p.d_gl_), "h2mi_malloc");
    check(h2mi_g1_fixed_base_mul_dev(pw.p, n, p.d_gl_, nullptr), "g_lagrange");
    check(h2mi_sync(), "sync");
    check(h2mi_bases_register_dev(p.d_g_, n, &p.h_g_), "register g");
    check(h2mi_bases_register_dev(p.d_gl_, n, &p.h_gl_), "register g_lagrange");
    return p;
  }
  ParamsKZG(ParamsKZG&& o) noexcept { *this = std::move(o); }
  ParamsKZG& operator=(ParamsKZG&& o) noexcept {
    k_ = o.k_; n_ = o.n_; d_g_ = o.d_g_; d_gl_ = o.d_gl_; h_g_ = o.h_g_; h_gl_ = o.h_gl_;
    g2_ = o.g2_; s_g2_ = o.s_g2_; have_g2_ = o.have_g2_;
    o.d_g_ = o.d_gl_ = nullptr; o.h_g_ = o.h_gl_ = 0;
    return *this;
  }
  ~ParamsKZG() {
    if (h_g_) h2mi_bases_release(h_g_);
    if (h_gl_) h2mi_bases_release(h_gl_);
    if (d_g_) h2mi_free(d_g_);
    if (d_gl_) h2mi_free(d_gl_);
  }
  uint32_t k() const { return k_; }
  uint64_t n() const { return n_; }
  // registered-bases handles for callers that keep their polynomials on the device (h2mi_msm_bn254_g1_dev)
  uint64_t g_handle() const { return h_g_; }
  uint64_t g_lagrange_handle() const { return h_gl_; }
  std::vector<G1Affine> get_g() const { return download(d_g_); }
  std::vector<G1Affine> get_g_lagrange() const { return download(d_gl_); }
  // commit / commit_lagrange: best_multiexp against the matching base set (KZG ignores the blind)
  G1 commit(const std::vector<Fr>& poly) const { return msm(h_g_, poly); }
  G1 commit_lagrange(const std::vector<Fr>& poly) const { return msm(h_gl_, poly); }

  // ParamsKZG::write / read [RECALL poly/kzg/commitment.rs of v2023_02_02; the file the scaffold caches as
  // params/kzg_bn254_{k}.srs]: k as u32 LE, g then g_lagrange as 32-byte compressed points, g2, s_g2 (64-byte
  // compressed G2).  The prover never uses the two G2 elements; they are carried as opaque bytes (set_g2 after
  // setup — the one G2 scalar multiplication is the Python host's or the ceremony's job — or taken from `read`).
  typedef std::array<uint8_t, 64> G2Bytes;
  void set_g2(const G2Bytes& g2, const G2Bytes& s_g2) { g2_ = g2; s_g2_ = s_g2; have_g2_ = true; }
  void write(std::ostream& w) const {
    if (!have_g2_) throw Error(H2MI_EINVAL, "ParamsKZG::write: g2 / s_g2 not set");
    const uint32_t k32 = k_;  // little-endian host
    w.write(reinterpret_cast<const char*>(&k32), 4);
    void* tmp = nullptr;
    check(h2mi_malloc(n_ * 32, &tmp), "h2mi_malloc");
    std::vector<char> host(n_ * 32);
    for (void* d : {d_g_, d_gl_}) {
      int rc = h2mi_g1_compress_dev(d, n_, tmp, nullptr);
      if (!rc) rc = h2mi_memcpy_d2h(host.data(), tmp, n_ * 32);
      if (rc) { h2mi_free(tmp); check(rc, "ParamsKZG::write"); }
      w.write(host.data(), (std::streamsize)host.size());
    }
    h2mi_free(tmp);
    w.write(reinterpret_cast<const char*>(g2_.data()), 64);
    w.write(reinterpret_cast<const char*>(s_g2_.data()), 64);
  }
  static ParamsKZG read(std::istream& r) {
    uint32_t k32 = 0;
    r.read(reinterpret_cast<char*>(&k32), 4);
    if (!r || k32 < 1 || k32 > 26) throw Error(H2MI_EINVAL, "ParamsKZG::read: bad header");
    ParamsKZG p(k32);
    const size_t n = p.n_;
    void* tmp = nullptr;
    check(h2mi_malloc(n * 32, &tmp), "h2mi_malloc");
    std::vector<char> host(n * 32);
    void** dst[2] = {&p.d_g_, &p.d_gl_};
    for (int i = 0; i < 2; i++) {
      r.read(host.data(), (std::streamsize)host.size());
      uint64_t bad = 0;
      int rc = r ? H2MI_OK : H2MI_EINVAL;
      if (!rc) rc = h2mi_malloc(n * 64, dst[i]);
      if (!rc) rc = h2mi_memcpy_h2d(tmp, host.data(), n * 32);
      if (!rc) rc = h2mi_g1_decompress_dev(tmp, n, *dst[i], &bad);
      if (!rc && bad) rc = H2MI_EINVAL;
      if (rc) { h2mi_free(tmp); check(rc, "ParamsKZG::read: truncated file or invalid point"); }
    }
    h2mi_free(tmp);
    r.read(reinterpret_cast<char*>(p.g2_.data()), 64);
    r.read(reinterpret_cast<char*>(p.s_g2_.data()), 64);
    if (!r) throw Error(H2MI_EINVAL, "ParamsKZG::read: truncated file");
    p.have_g2_ = true;
    check(h2mi_bases_register_dev(p.d_g_, n, &p.h_g_), "register g");
    check(h2mi_bases_register_dev(p.d_gl_, n, &p.h_gl_), "register g_lagrange");
    return p;
  }

 private:
  explicit ParamsKZG(uint32_t k) : k_(k), n_((uint64_t)1 << k) {}
  G1 msm(uint64_t h, const std::vector<Fr>& poly) const {
    if (poly.size() > n_) throw Error(H2MI_ERANGE, "commit: polynomial longer than the SRS");
    G1 out;
    check(h2mi_msm_bn254_g1(h, nullptr, (const uint64_t*)poly.data(), poly.size(), (uint64_t*)&out), "commit");
    return out;
  }
  std::vector<G1Affine> download(void* d) const {
    std::vector<G1Affine> v(n_);
    check(h2mi_memcpy_d2h(v.data(), d, n_ * 64), "d2h");
    return v;
  }
  uint32_t k_ = 0;
  uint64_t n_ = 0;
  void *d_g_ = nullptr, *d_gl_ = nullptr;
  uint64_t h_g_ = 0, h_gl_ = 0;
  G2Bytes g2_{}, s_g2_{};
  bool have_g2_ = false;
};
}  // namespace kzg
}  // namespace poly

// G1::batch_normalize
inline std::vector<G1Affine> batch_normalize(const std::vector<G1>& pts) {
  std::vector<G1Affine> out(pts.size());
  if (!pts.empty()) check(h2mi_g1_batch_normalize((const uint64_t*)pts.data(), pts.size(), (uint64_t*)out.data()), "batch_normalize");
  return out;
}

}  // namespace h2mi
