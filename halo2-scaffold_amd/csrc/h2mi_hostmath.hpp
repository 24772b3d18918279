// h2mi_hostmath.hpp — host-side arithmetic on SINGLE elements for the prover inside libh2mi.so (csrc/h2mi_prover.cpp): the
// base-field Montgomery product and inversion behind G1::batch_normalize of a phase's handful of points, the canonical order
// of Fr (BTreeSet<Fr> in ProverSHPLONK), the counter-based SplitMix64 stream of h2mi_fr_random_dev, and permutation/keygen.rs'
// Assembly.  Internal to the library build (tests/host/inv_host.cpp includes it to cross-check the inversions); bulk data never
// passes through here.
#pragma once
#include <algorithm>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/h2mi.hpp"

namespace h2mi {
namespace plonk {

inline Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
inline Fr to_canonical(const Fr& a) { return fr::mul(a, Fr{{1, 0, 0, 0}}); }  // a R^-1: the integer behind the Montgomery form
inline bool canonical_less(const Fr& a, const Fr& b) {                        // Fr's Ord: by canonical integer value
  Fr x = to_canonical(a), y = to_canonical(b);
  for (int i = 3; i >= 0; i--)
    if (x.l[i] != y.l[i]) return x.l[i] < y.l[i];
  return false;
}
inline Fr fr_delta() {  // halo2curves Fr::DELTA = 7^(2^28)
  Fr d = fr::from_u64(7);
  for (uint32_t i = 0; i < fr::S; i++) d = fr::mul(d, d);
  return d;
}
inline Fr pow_signed(const Fr& base, const Fr& base_inv, int64_t e) { return e >= 0 ? fr::pow_u64(base, (uint64_t)e) : fr::pow_u64(base_inv, (uint64_t)(-e)); }

// base-field Montgomery arithmetic, generic CIOS over 4 x 64 bits
namespace fq {
typedef unsigned __int128 u128;
constexpr uint64_t MODULUS[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
constexpr uint64_t INV = 0x87d20782e4866389ULL;
struct E {
  uint64_t l[4];
};
constexpr E ONE = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};  // R mod q
inline E mul(const E& a, const E& b) {
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
    const uint64_t m = t[0] * INV;
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
  E r;
  std::memcpy(r.l, t, 32);
  return r;
}
constexpr E R3 = {{0xb1cd6dafda1530dfULL, 0x62f210e6a7283db6ULL, 0xef7f0b0c0ada0afbULL, 0x20fd6e902d592544ULL}};  // 2^768 mod q
inline E invert_fermat(const E& a) {  // a^(q-2): the definition, kept as the cross-check of invert()
  const uint64_t e[4] = {MODULUS[0] - 2, MODULUS[1], MODULUS[2], MODULUS[3]};
  E r = ONE;
  for (int i = 255; i >= 0; i--) {
    r = mul(r, r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a);
  }
  return r;
}
inline E invert(const E& a) {  // division steps on the integer behind the Montgomery form (h2mi.hpp detail::inv_mod_odd), then back: ~1.5 us, not ~18
  if ((a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0) return a;
  E t;
  if (!h2mi::detail::inv_mod_odd(a.l, MODULUS, t.l)) return invert_fermat(a);
  return mul(t, R3);
}
}  // namespace fq

// G1::batch_normalize for the points of one phase: one inversion (Montgomery's trick), as the crate does before hashing them.
// The identity (z = 0) comes back as (0, 0): the caller's transcript refuses it, as the crate's does.
inline std::vector<G1Affine> normalize_host_batch(const std::vector<G1>& pts) {
  std::vector<fq::E> z(pts.size()), pre(pts.size());
  fq::E acc = fq::ONE;
  for (size_t i = 0; i < pts.size(); i++) {
    if (pts[i].is_identity()) z[i] = fq::ONE;
    else std::memcpy(z[i].l, pts[i].z, 32);
    pre[i] = acc;
    acc = fq::mul(acc, z[i]);
  }
  fq::E inv = pts.empty() ? fq::ONE : fq::invert(acc);
  std::vector<G1Affine> out(pts.size());
  for (size_t i = pts.size(); i-- > 0;) {
    const fq::E zi = fq::mul(pre[i], inv), zi2 = fq::mul(zi, zi);
    inv = fq::mul(inv, z[i]);
    if (pts[i].is_identity()) {
      std::memset(&out[i], 0, sizeof(G1Affine));
      continue;
    }
    fq::E x, y;
    std::memcpy(x.l, pts[i].x, 32);
    std::memcpy(y.l, pts[i].y, 32);
    const fq::E ax = fq::mul(x, zi2), ay = fq::mul(fq::mul(y, zi2), zi);
    std::memcpy(out[i].x, ax.l, 32);
    std::memcpy(out[i].y, ay.l, 32);
  }
  return out;
}

// counter-based SplitMix64 field elements — the stream h2mi_fr_random_dev produces on the device
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  uint64_t z = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline std::vector<Fr> uniform_fr(uint64_t seed, size_t count, uint64_t start = 0) {
  std::vector<Fr> v(count);
  for (size_t i = 0; i < count; i++) {
    Fr a;
    for (int j = 0; j < 4; j++) a.l[j] = splitmix64((seed << 32) + 4 * (start + i) + (uint64_t)j);
    a.l[3] &= (1ULL << 62) - 1;
    bool ge = true;
    for (int j = 3; j >= 0; j--) {
      if (a.l[j] > fr::MODULUS[j]) break;
      if (a.l[j] < fr::MODULUS[j]) { ge = false; break; }
    }
    if (ge) {
      unsigned __int128 bo = 0;
      for (int j = 0; j < 4; j++) {
        unsigned __int128 d = (unsigned __int128)a.l[j] - fr::MODULUS[j] - (uint64_t)bo;
        a.l[j] = (uint64_t)d;
        bo = (d >> 64) & 1;
      }
    }
    v[i] = a;  // the limbs ARE the Montgomery representation
  }
  return v;
}

// the same blinding sweep from a 256-bit key: element i of stream `stream` = Fr::from_u512 of ChaCha20 block i (RFC 7539 block function;
// 64-bit block counter, 64-bit stream id: rand_chacha's layout) — what h2mi_fr_random_chacha_dev produces on the device
inline void chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t stream, uint32_t out[16]) {
  const uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                           (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t x[16];
  for (int j = 0; j < 16; j++) x[j] = st[j];
  auto rotl = [](uint32_t v, int c) { return (v << c) | (v >> (32 - c)); };
  auto qr = [&](int a, int b, int c, int d) {
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
    x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
    x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
  };
  for (int r = 0; r < 10; r++) {
    qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
    qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
  }
  for (int j = 0; j < 16; j++) out[j] = x[j] + st[j];
}
inline std::vector<Fr> chacha_fr(const uint8_t key[32], uint64_t stream, size_t count, uint64_t start = 0) {
  uint32_t kw[8];
  std::memcpy(kw, key, 32);
  std::vector<Fr> v(count);
  for (size_t i = 0; i < count; i++) {
    uint32_t b[16];
    chacha20_block(kw, start + i, stream, b);
    Fr lo, hi;
    std::memcpy(lo.l, b, 32);
    std::memcpy(hi.l, b + 8, 32);
    v[i] = fr::add(fr::mul(lo, fr::R2), fr::mul(fr::mul(hi, fr::R2), fr::R2));  // from_u512
  }
  return v;
}

// plonk/permutation/keygen.rs Assembly: cycles merged smaller-into-larger, then the two mapping entries swapped.  Cells are
// (column within the permutation argument, row); identity entries are not stored, so the cost follows the number of constrained
// cells, not n.
typedef std::pair<uint32_t, uint32_t> Cell;
class PermutationAssembly {
 public:
  void copy(const Cell& left, const Cell& right) {
    const uint64_t l = key(left), r = key(right);
    uint64_t lc = get(aux_, l), rc = get(aux_, r);
    if (lc == rc) return;
    if (size(lc) < size(rc)) std::swap(lc, rc);
    sizes_[lc] = size(lc) + size(rc);
    uint64_t i = rc;
    do {
      aux_[i] = lc;
      i = get(mapping_, i);
    } while (i != rc);
    const uint64_t ml = get(mapping_, l), mr = get(mapping_, r);
    mapping_[l] = mr;
    mapping_[r] = ml;
  }
  // every stored (cell, image) pair, identity entries included if a later copy restored them
  template <class F>
  void for_each(F f) const {
    for (const auto& kv : mapping_) f(cell(kv.first), cell(kv.second));
  }

 private:
  static uint64_t key(const Cell& c) { return ((uint64_t)c.first << 32) | c.second; }
  static Cell cell(uint64_t k) { return Cell((uint32_t)(k >> 32), (uint32_t)k); }
  static uint64_t get(const std::unordered_map<uint64_t, uint64_t>& m, uint64_t c) {
    auto it = m.find(c);
    return it == m.end() ? c : it->second;
  }
  uint32_t size(uint64_t c) const {
    auto it = sizes_.find(c);
    return it == sizes_.end() ? 1u : it->second;
  }
  std::unordered_map<uint64_t, uint64_t> mapping_, aux_;
  std::unordered_map<uint64_t, uint32_t> sizes_;
};

}  // namespace plonk
}  // namespace h2mi
