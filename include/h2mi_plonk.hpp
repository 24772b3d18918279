// h2mi_plonk.hpp — C++17 host layer for the prover itself: keygen_vk / keygen_pk / create_proof of the reference's
// StandardPlonk circuit with every vector resident in HBM, over the C ABI (h2mi.h).
//
// Mirrors what the reference calls (examples/standard_plonk.rs:29-50):
//     let params = ParamsKZG::<Bn256>::setup(k, OsRng);
//     let vk = keygen_vk(&params, &circuit)?;  let pk = keygen_pk(&params, vk, &circuit)?;
//     let mut transcript = Blake2bWrite::<_, _, Challenge255<_>>::init(vec![]);
//     create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _, _>(
//         &params, &pk, &[circuit], &[&[]], OsRng, &mut transcript)?;
//     let proof = transcript.finalize();
// with the same names, argument meaning and failure behaviour (h2mi::Error where the crate panics / returns Err).
// halo2_proofs itself is an un-vendored dependency (reference Cargo.toml:13); the order of operations is restated from
// memory of v2023_02_02 (plonk/{keygen,prover}.rs, plonk/permutation/*, plonk/vanishing/*, poly/kzg/multiopen/shplonk*)
// and is the same restatement as the Python host (halo2-scaffold_amd/{circuits,keygen,prover,shplonk}.py) and the oracle
// (oracle/prover.py): the three produce identical proof bytes (tests/test_gpu_prover.py).
//
// The host runs what the crate runs single-threaded: witness cells, Blake2b, arithmetic on single field elements,
// launch order.  Every pass over a length-n vector is a device kernel; only 64-byte points and 32-byte evaluations
// cross PCIe.  rng: the reference passes OsRng; here `seed` drives counter-based SplitMix64 streams (seed + 1 advice
// blinding rows, seed + 2 permutation-product blinding rows, seed + 3 the random polynomial, generated on the device).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <map>
#include <memory>
#include <optional>
#include <utility>

#include <chrono>
#include "h2mi.hpp"
#include "h2mi_transcript.hpp"

namespace h2mi {
namespace plonk {

using arithmetic::DeviceVec;
typedef std::unique_ptr<DeviceVec> Dev;

// ---- a few scalar helpers ---------------------------------------------------------------------------------------
inline Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
inline Fr to_canonical(const Fr& a) { return fr::mul(a, Fr{{1, 0, 0, 0}}); }  // a R^-1: the integer behind the Montgomery form
inline bool canonical_less(const Fr& a, const Fr& b) {                        // Fr's Ord: by canonical integer value
  Fr x = to_canonical(a), y = to_canonical(b);
  for (int i = 3; i >= 0; i--)
    if (x.l[i] != y.l[i]) return x.l[i] < y.l[i];
  return false;
}
struct FrLess {
  bool operator()(const Fr& a, const Fr& b) const { return canonical_less(a, b); }
};
inline Fr fr_delta() {  // halo2curves Fr::DELTA = 7^(2^28)
  Fr d = fr::from_u64(7);
  for (uint32_t i = 0; i < fr::S; i++) d = fr::mul(d, d);
  return d;
}
inline Fr pow_signed(const Fr& base, const Fr& base_inv, int64_t e) { return e >= 0 ? fr::pow_u64(base, (uint64_t)e) : fr::pow_u64(base_inv, (uint64_t)(-e)); }

// G1::batch_normalize on the host for the handful of points a phase writes to the transcript (a lone device thread takes
// 0.3 ms for the inversion; here it is microseconds): base-field Montgomery arithmetic, generic CIOS over 4 x 64 bits
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
inline G1Affine normalize_host(const G1& p) {
  if (p.is_identity()) throw Error(H2MI_EINVAL, "cannot write points at infinity to the transcript");
  fq::E x, y, z;
  std::memcpy(x.l, p.x, 32);
  std::memcpy(y.l, p.y, 32);
  std::memcpy(z.l, p.z, 32);
  const fq::E zi = fq::invert(z), zi2 = fq::mul(zi, zi);
  const fq::E ax = fq::mul(x, zi2), ay = fq::mul(fq::mul(y, zi2), zi);
  G1Affine a;
  std::memcpy(a.x, ax.l, 32);
  std::memcpy(a.y, ay.l, 32);
  return a;
}

// G1::batch_normalize: one inversion for the whole phase (Montgomery's trick), as the crate does before hashing the points
inline std::vector<G1Affine> normalize_host_batch(const std::vector<G1>& pts) {
  std::vector<fq::E> z(pts.size()), pre(pts.size());
  fq::E acc = fq::ONE;
  for (size_t i = 0; i < pts.size(); i++) {
    if (pts[i].is_identity()) throw Error(H2MI_EINVAL, "cannot write points at infinity to the transcript");
    std::memcpy(z[i].l, pts[i].z, 32);
    pre[i] = acc;
    acc = fq::mul(acc, z[i]);
  }
  fq::E inv = pts.empty() ? fq::ONE : fq::invert(acc);
  std::vector<G1Affine> out(pts.size());
  for (size_t i = pts.size(); i-- > 0;) {
    const fq::E zi = fq::mul(pre[i], inv), zi2 = fq::mul(zi, zi);
    inv = fq::mul(inv, z[i]);
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

// ---- the circuit (reference src/circuits/standard_plonk.rs) -------------------------------------------------------
typedef std::pair<uint32_t, uint32_t> Cell;  // (column within its kind / within the permutation, row)
struct Synthesis {
  std::map<uint32_t, Fr> advice[3], fixed[5];
  std::vector<std::pair<Cell, Cell>> copies;  // constrain_equal(left, right) in call order
};
struct StandardPlonk {
  static constexpr uint32_t N_ADVICE = 3, N_FIXED = 5, CS_DEGREE = 3, BLINDING_FACTORS = 5;
  std::optional<Fr> x;  // Value::unknown() for keygen (examples/standard_plonk.rs:32)
  StandardPlonk() {}
  explicit StandardPlonk(const Fr& v) : x(v) {}
  Synthesis synthesize() const {
    Synthesis s;
    const Fr xv = x ? *x : fr_zero(), one = fr::ONE, minus_one = fr::neg(fr::ONE);
    auto copy_advice = [&](uint32_t col, uint32_t row) {  // assign, then constrain_equal(new cell, x's cell (a, 0))
      s.advice[col][row] = xv;
      s.copies.push_back({Cell(col, row), Cell(0, 0)});
    };
    s.advice[0][0] = xv;
    copy_advice(0, 1);
    copy_advice(1, 1);
    s.advice[2][1] = fr::mul(xv, xv);
    s.fixed[2][1] = minus_one;  // q_c
    s.fixed[3][1] = one;        // q_ab
    copy_advice(0, 2);
    copy_advice(1, 2);
    s.advice[2][2] = fr::add(fr::mul(xv, xv), fr::from_u64(72));
    s.fixed[2][2] = minus_one;
    s.fixed[3][2] = one;
    s.fixed[4][2] = fr::from_u64(72);  // constant
    return s;
  }
};

// plonk/permutation/keygen.rs Assembly: cycles merged smaller-into-larger, then the two mapping entries swapped
class PermutationAssembly {
 public:
  void copy(const Cell& left, const Cell& right) {
    Cell lc = get(aux_, left), rc = get(aux_, right);
    if (lc == rc) return;
    if (size(lc) < size(rc)) std::swap(lc, rc);
    sizes_[lc] = size(lc) + size(rc);
    Cell i = rc;
    do {
      aux_[i] = lc;
      i = get(mapping_, i);
    } while (i != rc);
    Cell ml = get(mapping_, left), mr = get(mapping_, right);
    mapping_[left] = mr;
    mapping_[right] = ml;
  }
  const std::map<Cell, Cell>& mapping() const { return mapping_; }

 private:
  static Cell get(const std::map<Cell, Cell>& m, const Cell& c) {
    auto it = m.find(c);
    return it == m.end() ? c : it->second;
  }
  uint32_t size(const Cell& c) const {
    auto it = sizes_.find(c);
    return it == sizes_.end() ? 1u : it->second;
  }
  std::map<Cell, Cell> mapping_, aux_;
  std::map<Cell, uint32_t> sizes_;
};

// ---- keys ---------------------------------------------------------------------------------------------------------
struct VerifyingKey {
  uint32_t k = 0, cs_degree = 0;
  std::vector<G1Affine> fixed_commitments, permutation_commitments;
  Fr transcript_repr;
  // stand-in for the crate's pinned-Debug-text hash (not reproducible without the crate): the same Blake2b-512 /
  // "Halo2-Verify-Key" over k, the degree and the compressed commitments
  std::vector<uint8_t> to_bytes() const {
    std::vector<uint8_t> out(8);
    std::memcpy(out.data(), &k, 4);
    std::memcpy(out.data() + 4, &cs_degree, 4);
    for (const auto* set : {&fixed_commitments, &permutation_commitments})
      for (const G1Affine& p : *set) {
        auto b = serde::g1_to_bytes(p);
        out.insert(out.end(), b.begin(), b.end());
      }
    return out;
  }
  void compute_transcript_repr() {
    blake2b::State h("Halo2-Verify-Key");
    std::vector<uint8_t> s = to_bytes();
    uint64_t len = s.size();
    h.update(&len, 8);
    h.update(s.data(), s.size());
    transcript_repr = serde::fr_from_bytes_wide(h.digest());
  }
};

struct Columns {  // Lagrange (optional) / coefficient / extended-coset forms of a group of columns
  std::vector<Dev> values, polys, cosets;
};
struct ProvingKey {
  VerifyingKey vk;
  poly::EvaluationDomain domain;
  Columns fixed, permutation;
  Dev l0, l_last, l_active;
  // support of the copy constraints: sorted positions set * usable_rows + row (uint32) at which a column of the set is
  // moved by the permutation — what h2mi_plonk_permutation_products_sparse_dev runs over
  Dev active_rows;
  uint32_t n_active = 0;
  explicit ProvingKey(const VerifyingKey& v) : vk(v), domain(v.cs_degree, v.k) {}
  const VerifyingKey& get_vk() const { return vk; }
};

namespace detail {
inline void patch(DeviceVec& d, uint32_t row, const Fr& v) { check(h2mi_memcpy_h2d_async((char*)d.p + (size_t)row * 32, v.l, 32), "patch"); }
inline Dev zeros(size_t n) {
  Dev d(new DeviceVec(n));
  check(h2mi_memset_zero(d->p, n * 32), "zero");
  return d;
}
inline std::vector<Dev> fixed_columns(const Synthesis& syn, size_t n) {
  std::vector<Dev> cols;
  for (const auto& assigned : syn.fixed) {
    Dev d = zeros(n);
    for (const auto& kv : assigned) patch(*d, kv.first, kv.second);
    cols.push_back(std::move(d));
  }
  return cols;
}
// sigma_j[i] = DELTA^(j') omega^(i') for (j', i') = mapping[(j, i)]
inline std::vector<Dev> sigma_columns(const Synthesis& syn, const poly::EvaluationDomain& dom, size_t n, uint32_t n_perm) {
  PermutationAssembly asm_;
  for (const auto& c : syn.copies) asm_.copy(c.first, c.second);
  DeviceVec omega_pows(n);
  check(h2mi_fr_powers_dev(omega_pows.p, n, dom.get_omega().l, nullptr), "powers");
  const Fr delta = fr_delta();
  std::vector<Dev> cols;
  for (uint32_t j = 0; j < n_perm; j++) {
    Dev d(new DeviceVec(n));
    const void* ptrs[1] = {omega_pows.p};
    Fr sc = fr::pow_u64(delta, j);
    check(h2mi_fr_lincomb_dev(ptrs, sc.l, 1, n, d->p, nullptr), "identity permutation");
    cols.push_back(std::move(d));
  }
  for (const auto& kv : asm_.mapping())
    if (kv.first != kv.second)
      patch(*cols[kv.first.first], kv.first.second, fr::mul(fr::pow_u64(delta, kv.second.first), fr::pow_u64(dom.get_omega(), kv.second.second)));
  check(h2mi_sync(), "sync");
  return cols;
}
inline void to_poly_and_coset(const poly::EvaluationDomain& dom, const DeviceVec& lagr, Dev& poly, Dev& coset) {
  const size_t n = (size_t)1 << dom.k();
  poly.reset(new DeviceVec(n));
  coset.reset(new DeviceVec(dom.extended_len()));
  check(h2mi_ntt_bn254_fr_oop_dev(lagr.p, n, poly->p, dom.k(), dom.get_omega_inv().l, nullptr, dom.get_ifft_divisor().l, nullptr), "lagrange_to_coeff");
  check(h2mi_ntt_bn254_fr_oop_dev(poly->p, n, coset->p, dom.extended_k(), dom.get_extended_omega().l, dom.get_g_coset().l, nullptr, nullptr),
        "coeff_to_extended");
}
inline void to_poly_and_coset_into(const poly::EvaluationDomain& dom, const DeviceVec& lagr, DeviceVec& poly, DeviceVec& coset,
                                   h2mi_stream_t stream = nullptr) {
  const size_t n = (size_t)1 << dom.k();
  // the domain holds n^-1 and the coset generator: two 254-bit exponentiations per call otherwise (round 3)
  check(h2mi_ntt_bn254_fr_oop_dev(lagr.p, n, poly.p, dom.k(), dom.get_omega_inv().l, nullptr, dom.get_ifft_divisor().l, stream), "lagrange_to_coeff");
  check(h2mi_ntt_bn254_fr_oop_dev(poly.p, n, coset.p, dom.extended_k(), dom.get_extended_omega().l, dom.get_g_coset().l, nullptr, stream),
        "coeff_to_extended");
}
// commit columns (device-resident, n elements) -> affine points on the host
inline std::vector<G1Affine> commit_points(uint64_t handle, const std::vector<const void*>& cols, size_t n) {
  const size_t k = cols.size();
  DeviceVec out(3 * k), aff(2 * k);  // 96 B / 64 B per point
  for (size_t i = 0; i < k; i++) check(h2mi_msm_bn254_g1_dev(handle, cols[i], n, (char*)out.p + 96 * i, nullptr), "commit");
  check(h2mi_join(), "join");
  check(h2mi_g1_batch_normalize_dev(out.p, k, aff.p, nullptr), "batch_normalize");
  std::vector<G1Affine> pts(k);
  check(h2mi_memcpy_d2h(pts.data(), aff.p, k * 64), "d2h");
  return pts;
}
}  // namespace detail

inline VerifyingKey keygen_vk(const poly::kzg::ParamsKZG& params, const StandardPlonk& circuit) {
  const size_t n = params.n();
  poly::EvaluationDomain dom(StandardPlonk::CS_DEGREE, params.k());
  Synthesis syn = StandardPlonk().synthesize();  // without_witnesses()
  (void)circuit;
  auto fixed = detail::fixed_columns(syn, n);
  auto sigma = detail::sigma_columns(syn, dom, n, StandardPlonk::N_ADVICE);
  VerifyingKey vk;
  vk.k = params.k();
  vk.cs_degree = StandardPlonk::CS_DEGREE;
  std::vector<const void*> f, s;
  for (auto& d : fixed) f.push_back(d->p);
  for (auto& d : sigma) s.push_back(d->p);
  vk.fixed_commitments = detail::commit_points(params.g_lagrange_handle(), f, n);
  vk.permutation_commitments = detail::commit_points(params.g_lagrange_handle(), s, n);
  vk.compute_transcript_repr();
  return vk;
}

inline std::unique_ptr<ProvingKey> keygen_pk(const poly::kzg::ParamsKZG& params, const VerifyingKey& vk, const StandardPlonk& circuit) {
  (void)circuit;
  std::unique_ptr<ProvingKey> pk(new ProvingKey(vk));
  const poly::EvaluationDomain& dom = pk->domain;
  const size_t n = params.n();
  Synthesis syn = StandardPlonk().synthesize();
  auto fixed = detail::fixed_columns(syn, n);
  auto sigma = detail::sigma_columns(syn, dom, n, StandardPlonk::N_ADVICE);
  for (auto& col : fixed) {
    Dev p, e;
    detail::to_poly_and_coset(dom, *col, p, e);
    pk->fixed.polys.push_back(std::move(p));
    pk->fixed.cosets.push_back(std::move(e));
  }
  for (auto& col : sigma) {
    Dev p, e;
    detail::to_poly_and_coset(dom, *col, p, e);
    pk->permutation.polys.push_back(std::move(p));
    pk->permutation.cosets.push_back(std::move(e));
    pk->permutation.values.push_back(std::move(col));
  }
  const uint32_t u = (uint32_t)n - (StandardPlonk::BLINDING_FACTORS + 1);
  Dev l0 = detail::zeros(n), ll = detail::zeros(n), la(new DeviceVec(n));
  detail::patch(*l0, 0, fr::ONE);
  detail::patch(*ll, u, fr::ONE);
  check(h2mi_fr_fill_dev(la->p, n, fr::ONE.l, nullptr), "fill");
  check(h2mi_memset_zero((char*)la->p + (size_t)u * 32, (n - u) * 32), "zero");
  Dev unused;
  detail::to_poly_and_coset(dom, *l0, unused, pk->l0);
  detail::to_poly_and_coset(dom, *ll, unused, pk->l_last);
  detail::to_poly_and_coset(dom, *la, unused, pk->l_active);
  {
    PermutationAssembly asm_;
    for (const auto& c : syn.copies) asm_.copy(c.first, c.second);
    const uint32_t chunk = StandardPlonk::CS_DEGREE - 2;
    std::vector<uint32_t> pos;
    for (const auto& kv : asm_.mapping())
      if (kv.first != kv.second && kv.first.second < u) pos.push_back((kv.first.first / chunk) * u + kv.first.second);
    std::sort(pos.begin(), pos.end());
    pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
    pk->n_active = (uint32_t)pos.size();
    pk->active_rows.reset(new DeviceVec(pos.size() / 8 + 1));
    if (!pos.empty()) check(h2mi_memcpy_h2d(pk->active_rows->p, pos.data(), pos.size() * 4), "active rows");
  }
  check(h2mi_sync(), "sync");
  return pk;
}

// ---- ProverSHPLONK (poly/kzg/multiopen/shplonk.rs construct_intermediate_sets + shplonk/prover.rs) -------------------
struct ProverQuery {
  const DeviceVec* poly;
  Fr point, eval;
};
namespace detail {
// coefficient lists (low to high) of the Lagrange basis polynomials of `pts`: the part of an interpolation that depends on the
// points alone — once per rotation set, with ONE inversion, instead of one 254-bit exponentiation per basis polynomial per member
inline std::vector<std::vector<Fr>> lagrange_basis(const std::vector<Fr>& pts) {
  const size_t m = pts.size();
  std::vector<std::vector<Fr>> nums;
  std::vector<Fr> dens;
  for (size_t j = 0; j < m; j++) {
    std::vector<Fr> num = {fr::ONE};
    Fr den = fr::ONE;
    for (size_t t = 0; t < m; t++) {
      if (t == j) continue;
      std::vector<Fr> nxt(num.size() + 1, fr_zero());
      for (size_t i = 0; i < num.size(); i++) {  // num *= (X - pts[t])
        nxt[i + 1] = fr::add(nxt[i + 1], num[i]);
        nxt[i] = fr::sub(nxt[i], fr::mul(num[i], pts[t]));
      }
      num = nxt;
      den = fr::mul(den, fr::sub(pts[j], pts[t]));
    }
    nums.push_back(num);
    dens.push_back(den);
  }
  const std::vector<Fr> inv = fr::batch_invert(dens);
  for (size_t j = 0; j < m; j++)
    for (Fr& c : nums[j]) c = fr::mul(c, inv[j]);
  return nums;
}
inline std::vector<Fr> interpolate(const std::vector<std::vector<Fr>>& basis, const std::vector<Fr>& evals) {
  std::vector<Fr> out(basis.size(), fr_zero());
  for (size_t j = 0; j < basis.size(); j++)
    for (size_t i = 0; i < basis[j].size(); i++) out[i] = fr::add(out[i], fr::mul(basis[j][i], evals[j]));
  return out;
}
inline std::vector<Fr> interpolate(const std::vector<Fr>& pts, const std::vector<Fr>& evals) { return interpolate(lagrange_basis(pts), evals); }
inline Fr horner(const std::vector<Fr>& c, const Fr& x) {
  Fr acc = fr_zero();
  for (size_t i = c.size(); i-- > 0;) acc = fr::add(fr::mul(acc, x), c[i]);
  return acc;
}
inline Fr vanishing_at(const std::vector<Fr>& roots, const Fr& z) {
  Fr acc = fr::ONE;
  for (const Fr& r : roots) acc = fr::mul(acc, fr::sub(z, r));
  return acc;
}
inline bool contains(const std::vector<Fr>& v, const Fr& x) { return std::find(v.begin(), v.end(), x) != v.end(); }
inline void lincomb(const std::vector<const DeviceVec*>& polys, const std::vector<Fr>& scalars, size_t n, DeviceVec& out, h2mi_stream_t stream = nullptr) {
  std::vector<const void*> ptrs;
  for (auto* p : polys) ptrs.push_back(p->p);
  check(h2mi_fr_lincomb_dev(ptrs.data(), (const uint64_t*)scalars.data(), polys.size(), n, out.p, stream), "lincomb");
}
inline void add_head(DeviceVec& poly, const std::vector<Fr>& head, h2mi_stream_t stream = nullptr) {
  check(h2mi_fr_add_head_dev(poly.p, (const uint64_t*)head.data(), head.size(), stream), "add_head");
}
// out = src / prod (X - root); `out` must have been zeroed (the quotient has n - #roots coefficients, the rest stay
// zero); intermediate quotients alternate between tmp and tmp2 — without a tmp2, src is clobbered when there are >= 2 roots
inline void kate_chain(DeviceVec& src, size_t n, const std::vector<Fr>& roots, DeviceVec& tmp, DeviceVec& out, h2mi_stream_t stream = nullptr,
                       DeviceVec* tmp2 = nullptr) {
  if (roots.size() >= 2 && roots.size() <= 4) {  // one round: independent divisions weighted by 1 / prod_{k != i} (r_i - r_k)
    std::vector<Fr> both(roots);  // the roots and the partial-fraction denominators, inverted together
    for (size_t i = 0; i < roots.size(); i++) {
      Fr d = fr::ONE;
      for (size_t k = 0; k < roots.size(); k++)
        if (k != i) d = fr::mul(d, fr::sub(roots[i], roots[k]));
      both.push_back(d);
    }
    both = fr::batch_invert(both);
    const std::vector<Fr> inv(both.begin(), both.begin() + roots.size()), w(both.begin() + roots.size(), both.end());
    check(h2mi_fr_kate_division_multi_dev(src.p, n, (const uint64_t*)roots.data(), (const uint64_t*)inv.data(), (const uint64_t*)w.data(), roots.size(),
                                          out.p, stream),
          "kate_division_multi");
    return;
  }
  DeviceVec* cur = &src;
  DeviceVec* bufs[2] = {&tmp, tmp2 ? tmp2 : &src};
  size_t len = n;
  for (size_t i = 0; i < roots.size(); i++) {
    DeviceVec* dst = i + 1 == roots.size() ? &out : bufs[i % 2];
    Fr binv = fr::invert(roots[i]);
    check(h2mi_fr_kate_division_dev(cur->p, len, roots[i].l, binv.l, dst->p, stream), "kate_division");
    cur = dst;
    len--;
  }
}
struct RotationSet {
  std::vector<Fr> points;                                               // ascending canonical value (BTreeSet<Fr>)
  std::vector<std::pair<const DeviceVec*, std::vector<Fr>>> members;  // (polynomial, evaluation at each point)
};
}  // namespace detail

struct ShplonkScratch {  // n-element device vectors the argument works in (at least as many q as rotation sets)
  DeviceVec *nx, *tmp, *h_x, *l_x, *h2_x;
  std::vector<Dev>* q;
  std::vector<Dev>* s;  // per rotation set: sum_j y^j P_ij(X) - R_i(X), kept from the quotient step for the linearisation
  // optional extra lanes (side stream + its own nx / tmp): the rotation sets' quotient chains are independent, so set i
  // runs on lane i mod (1 + lanes.size()) and the longest chain, not their sum, is waited for
  struct Lane {
    h2mi_stream_t stream;
    DeviceVec *nx, *tmp;
  };
  std::vector<Lane> lanes;
};
struct ShplonkLanes {  // two side lanes, owned by a prover workspace
  Dev nx[2], tmp[2];
  h2mi_stream_t stream[2] = {nullptr, nullptr};
  ShplonkLanes(const ShplonkLanes&) = delete;
  ShplonkLanes& operator=(const ShplonkLanes&) = delete;
  explicit ShplonkLanes(size_t n) {
    for (int i = 0; i < 2; i++) {
      nx[i].reset(new DeviceVec(n));
      tmp[i].reset(new DeviceVec(n));
      check(h2mi_stream_create(&stream[i]), "stream_create");
    }
  }
  ~ShplonkLanes() {
    for (h2mi_stream_t s : stream)
      if (s) h2mi_stream_destroy(s);
  }
  std::vector<ShplonkScratch::Lane> lanes() const {
    static const bool off = std::getenv("H2MI_SHPLONK_LANES") && std::getenv("H2MI_SHPLONK_LANES")[0] == '0';  // A/B knob
    if (off) return {};
    return {{stream[0], nx[0].get(), tmp[0].get()}, {stream[1], nx[1].get(), tmp[1].get()}};
  }
};
template <class CommitAndWrite>
inline void shplonk_create_proof(size_t n, transcript::Blake2bWrite& tr, const std::vector<ProverQuery>& queries, CommitAndWrite commit_and_write,
                                 const ShplonkScratch& sc) {
  using namespace detail;
  const Fr y = tr.squeeze_challenge();
  // construct_intermediate_sets
  std::vector<std::pair<const DeviceVec*, std::vector<std::pair<Fr, Fr>>>> by_poly;  // first-appearance order
  std::vector<Fr> super_points;
  for (const ProverQuery& q : queries) {
    if (!contains(super_points, q.point)) super_points.push_back(q.point);
    auto it = std::find_if(by_poly.begin(), by_poly.end(), [&](const auto& e) { return e.first == q.poly; });
    if (it == by_poly.end()) {
      by_poly.push_back({q.poly, {{q.point, q.eval}}});
    } else if (std::none_of(it->second.begin(), it->second.end(), [&](const auto& pe) { return pe.first == q.point; })) {
      it->second.push_back({q.point, q.eval});
    }
  }
  std::sort(super_points.begin(), super_points.end(), canonical_less);
  std::vector<RotationSet> sets;
  for (auto& e : by_poly) {
    std::vector<Fr> pts;
    for (auto& pe : e.second) pts.push_back(pe.first);
    std::sort(pts.begin(), pts.end(), canonical_less);
    auto rs = std::find_if(sets.begin(), sets.end(), [&](const RotationSet& s) { return s.points == pts; });
    if (rs == sets.end()) {
      sets.push_back(RotationSet{pts, {}});
      rs = sets.end() - 1;
    }
    std::vector<Fr> evals;
    for (const Fr& p : rs->points) evals.push_back(std::find_if(e.second.begin(), e.second.end(), [&](const auto& pe) { return pe.first == p; })->second);
    rs->members.push_back({e.first, evals});
  }
  const Fr v = tr.squeeze_challenge();
  {  // the divisions below need the power tables of every opening point and of its inverse: built now, in one launch
    std::vector<Fr> bases = super_points;
    const std::vector<Fr> inv = fr::batch_invert(super_points);
    bases.insert(bases.end(), inv.begin(), inv.end());
    if (!bases.empty() && bases.size() <= 32) check(h2mi_fr_powtab_prefetch_dev((const uint64_t*)bases.data(), bases.size(), n, nullptr), "powtab_prefetch");
  }
  DeviceVec &nx = *sc.nx, &tmp = *sc.tmp, &h_x = *sc.h_x, &l_x = *sc.l_x, &h2_x = *sc.h2_x;
  std::vector<Dev>& q = *sc.q;
  std::vector<Dev>& ssum = *sc.s;
  if (sets.size() > q.size() || sets.size() > ssum.size()) throw Error(H2MI_ERANGE, "shplonk: more rotation sets than scratch vectors");
  std::vector<std::vector<Fr>> remainders;  // R_i(X) = sum_j y^j R_ij(X), low to high
  for (size_t i = 0; i < sets.size(); i++) check(h2mi_memset_zero(q[i]->p, n * 32), "zero");
  for (const auto& lane : sc.lanes) check(h2mi_stream_wait(lane.stream, nullptr), "stream_wait");
  for (size_t i = 0; i < sets.size(); i++) {
    const RotationSet& rs = sets[i];
    const size_t lane = i % (1 + sc.lanes.size());
    h2mi_stream_t stream = lane ? sc.lanes[lane - 1].stream : nullptr;
    DeviceVec& lnx = lane ? *sc.lanes[lane - 1].nx : nx;
    DeviceVec& ltmp = lane ? *sc.lanes[lane - 1].tmp : tmp;
    std::vector<Fr> ypow(rs.members.size(), fr::ONE);
    for (size_t j = 1; j < ypow.size(); j++) ypow[j] = fr::mul(ypow[j - 1], y);
    std::vector<const DeviceVec*> polys;
    for (auto& m : rs.members) polys.push_back(m.first);
    lincomb(polys, ypow, n, *ssum[i], stream);
    std::vector<Fr> rsum(rs.points.size(), fr_zero());
    const std::vector<std::vector<Fr>> basis = lagrange_basis(rs.points);
    for (size_t j = 0; j < rs.members.size(); j++) {
      std::vector<Fr> r = interpolate(basis, rs.members[j].second);
      for (size_t t = 0; t < r.size(); t++) rsum[t] = fr::sub(rsum[t], fr::mul(ypow[j], r[t]));
    }
    add_head(*ssum[i], rsum, stream);
    std::vector<Fr> rem;
    for (const Fr& c : rsum) rem.push_back(fr::neg(c));
    remainders.push_back(rem);
    kate_chain(*ssum[i], n, rs.points, ltmp, *q[i], stream, &lnx);
  }
  for (const auto& lane : sc.lanes) check(h2mi_stream_wait(nullptr, lane.stream), "stream_wait");
  {
    std::vector<const DeviceVec*> polys;
    std::vector<Fr> vpow(sets.size(), fr::ONE);
    for (size_t i = 1; i < vpow.size(); i++) vpow[i] = fr::mul(vpow[i - 1], v);
    for (size_t i = 0; i < sets.size(); i++) polys.push_back(q[i].get());
    lincomb(polys, vpow, n, h_x);
  }
  commit_and_write(h_x);
  const Fr u = tr.squeeze_challenge();
  const Fr zt_eval = vanishing_at(super_points, u);
  std::vector<Fr> z_diffs;
  for (const RotationSet& rs : sets) {
    std::vector<Fr> diffs;
    for (const Fr& p : super_points)
      if (!contains(rs.points, p)) diffs.push_back(p);
    z_diffs.push_back(vanishing_at(diffs, u));
  }
  const Fr norm = fr::invert(z_diffs[0]);
  // linearisation: sum_j y^j (P_ij(X) - R_ij(u)) = S_i(X) + R_i(X) - R_i(u) with S_i the vector the quotient step left in
  // ssum[i] — one linear combination over the rotation sets' sums and h(X) instead of every opened polynomial again
  std::vector<const DeviceVec*> polys;
  std::vector<Fr> scalars;
  size_t head_len = 0;
  for (const RotationSet& rs : sets) head_len = std::max(head_len, rs.points.size());
  std::vector<Fr> head(head_len, fr_zero());
  Fr vp = fr::ONE;
  for (size_t i = 0; i < sets.size(); i++) {
    const Fr w = fr::mul(fr::mul(vp, z_diffs[i]), norm);
    polys.push_back(ssum[i].get());
    scalars.push_back(w);
    for (size_t t = 0; t < remainders[i].size(); t++) head[t] = fr::add(head[t], fr::mul(w, remainders[i][t]));
    head[0] = fr::sub(head[0], fr::mul(w, horner(remainders[i], u)));
    vp = fr::mul(vp, v);
  }
  polys.push_back(&h_x);
  scalars.push_back(fr::neg(fr::mul(zt_eval, norm)));
  lincomb(polys, scalars, n, l_x);
  add_head(l_x, head);
  check(h2mi_memset_zero(h2_x.p, n * 32), "zero");
  kate_chain(l_x, n, {u}, tmp, h2_x);
  commit_and_write(h2_x);
}

// device buffers of one prover, reused from proof to proof (the reference's examples prove repeatedly against one pk:
// examples/linear_regression.rs:178-185); allocating and freeing ~1 GB of vectors per proof costs more than the proof
struct ProverWorkspace {
  size_t n, ext;
  std::vector<Dev> advice, advice_polys, advice_cosets, z, z_polys, z_cosets, shplonk_q, shplonk_s;
  Dev random_poly, h, h_poly, points, evals, nx, tmp, h_x, l_x, h2_x;
  std::unique_ptr<ShplonkLanes> lanes;
  h2mi_stream_t side = nullptr;  // transforms of the advice columns run here, beside the permutation argument's chain
  // optional host-side phase clock (untraced: a kernel tracer distorts exactly the host-paced stretches one wants to see): when
  // time_phases is set, create_proof adds the wall clock between its transcript joins to phase_us[0 .. 5] = advice committed,
  // z + random committed, h pieces committed, evaluations written, SHPLONK's first and second commitment written
  bool time_phases = false;
  double phase_us[6] = {0, 0, 0, 0, 0, 0};
  ProverWorkspace(const ProverWorkspace&) = delete;
  ProverWorkspace& operator=(const ProverWorkspace&) = delete;
  ~ProverWorkspace() {
    if (side) h2mi_stream_destroy(side);
  }
  ProverWorkspace(const poly::kzg::ParamsKZG& params, const ProvingKey& pk) : n(params.n()), ext(pk.domain.extended_len()) {
    auto vec = [&](size_t cnt) { return Dev(new DeviceVec(cnt)); };
    for (uint32_t j = 0; j < StandardPlonk::N_ADVICE; j++) {
      advice.push_back(vec(n)); advice_polys.push_back(vec(n)); advice_cosets.push_back(vec(ext));
      z.push_back(vec(n)); z_polys.push_back(vec(n)); z_cosets.push_back(vec(ext));
    }
    for (int i = 0; i < 4; i++) shplonk_q.push_back(vec(n));
    for (int i = 0; i < 4; i++) shplonk_s.push_back(vec(n));
    random_poly = vec(n); h = vec(ext); h_poly = vec(n); points = vec(12); evals = vec(32);
    nx = vec(n); tmp = vec(n); h_x = vec(n); l_x = vec(n); h2_x = vec(n);
    lanes.reset(new ShplonkLanes(n));
    check(h2mi_stream_create(&side), "stream_create");
  }
};

// ---- create_proof ---------------------------------------------------------------------------------------------------
inline void create_proof(const poly::kzg::ParamsKZG& params, const ProvingKey& pk, const StandardPlonk& circuit, uint64_t seed,
                         transcript::Blake2bWrite& tr, ProverWorkspace* workspace = nullptr) {
  using namespace detail;
  if (!circuit.x) throw Error(H2MI_EINVAL, "create_proof: the circuit has no witness (Value::unknown())");
  std::unique_ptr<ProverWorkspace> own;
  if (!workspace) {
    own.reset(new ProverWorkspace(params, pk));
    workspace = own.get();
  }
  ProverWorkspace& ws = *workspace;
  const poly::EvaluationDomain& d = pk.domain;
  const size_t n = params.n(), ext = d.extended_len();
  const uint32_t bf = StandardPlonk::BLINDING_FACTORS, u = (uint32_t)n - (bf + 1), na = StandardPlonk::N_ADVICE;
  const Fr omega = d.get_omega(), omega_inv = d.get_omega_inv();
  DeviceVec& points = *ws.points;  // 4 x 96 B
  auto write_phase_points = [&](size_t k) {  // join, G1::batch_normalize (host: k modular inversions), write_point
    std::vector<G1> jac(k);
    check(h2mi_memcpy_d2h(jac.data(), points.p, k * 96), "d2h");  // joins the MSM pipeline
    for (const G1Affine& a : normalize_host_batch(jac)) tr.write_point(a);
  };
  auto commit = [&](uint64_t handle, const void* col, size_t slot) { check(h2mi_msm_bn254_g1_dev(handle, col, n, (char*)points.p + 96 * slot, nullptr), "commit"); };
  // the commitments of one phase (results in slots 0 .. k-1): one call, so that small circuits get one set of launches for all of them
  // `sparse`: witness columns and grand products of this circuit (a handful of assigned rows; constant but for the copy constraints)
  // `inorder`: the group is everything its phase commits and is read back next — its reductions follow its accumulation on one stream
  auto commit_phase = [&](uint64_t handle, const std::vector<const void*>& cols, bool sparse = false, bool inorder = false) {
    const unsigned flags = (sparse ? H2MI_MSM_SPARSE : 0u) | (inorder ? H2MI_MSM_INORDER : 0u);
    check(h2mi_msm_bn254_g1_phase_dev(handle, cols.data(), cols.size(), n, points.p, flags, nullptr), "commit");
  };

  auto phase_t0 = std::chrono::steady_clock::now();
  auto mark = [&](int phase) {
    if (!ws.time_phases) return;
    const auto t = std::chrono::steady_clock::now();
    ws.phase_us[phase] += std::chrono::duration<double, std::micro>(t - phase_t0).count();
    phase_t0 = t;
  };
  tr.common_scalar(pk.vk.transcript_repr);  // vk.hash_into

  // advice columns: witness cells + blinding rows, committed in the Lagrange basis
  Synthesis syn = circuit.synthesize();
  std::vector<Fr> blind = uniform_fr(seed + 1, (size_t)na * (bf + 1));
  std::vector<Dev>&advice = ws.advice, &advice_polys = ws.advice_polys, &advice_cosets = ws.advice_cosets, &z = ws.z, &z_polys = ws.z_polys,
              &z_cosets = ws.z_cosets;
  {
    // assigned cells and blinding rows of every column in ONE launch (h2mi_fr_patch_cells_dev) instead of a dozen 32-byte copies
    std::vector<void*> cells;
    std::vector<Fr> vals;
    for (uint32_t j = 0; j < na; j++) {
      DeviceVec& col = *advice[j];
      check(h2mi_memset_zero(col.p, n * 32), "zero");
      for (const auto& kv : syn.advice[j]) {
        cells.push_back((char*)col.p + (size_t)kv.first * 32);
        vals.push_back(kv.second);
      }
      for (uint32_t r = 0; r <= bf; r++) {
        cells.push_back((char*)col.p + (size_t)(u + r) * 32);
        vals.push_back(blind[(size_t)j * (bf + 1) + r]);
      }
    }
    check(h2mi_fr_patch_cells_dev(cells.data(), (const uint64_t*)vals.data(), cells.size(), nullptr), "advice cells");
  }
  {
    std::vector<const void*> cols;
    for (uint32_t j = 0; j < na; j++) cols.push_back(advice[j]->p);
    commit_phase(params.g_lagrange_handle(), cols, /*sparse=*/true, /*inorder=*/true);
  }
  check(h2mi_msm_flush(), "flush");  // the bucket reductions start now, not when the host reaches the join
  // the advice columns' coefficient / extended forms wait for no challenge: on the side stream they run beside the
  // transcript round trip and the permutation argument's latency-bound scans
  check(h2mi_stream_wait(ws.side, nullptr), "stream_wait");
  for (uint32_t j = 0; j < na; j++) to_poly_and_coset_into(d, *advice[j], *advice_polys[j], *advice_cosets[j], ws.side);
  write_phase_points(na);
  mark(0);
  (void)tr.squeeze_challenge();  // theta: drawn even without lookups
  const Fr beta = tr.squeeze_challenge(), gamma = tr.squeeze_challenge();
  // vanishing argument's random polynomial: written after the z commitments, dependent on nothing — its dense MSM is
  // queued first and accumulates beside the grand products
  DeviceVec& random_poly = *ws.random_poly;
  check(h2mi_fr_random_dev(random_poly.p, n, seed + 3, 0, nullptr), "random_poly");
  commit(params.g_handle(), random_poly.p, na);

  // permutation argument: every set in one device pass (chunk length cs.degree() - 2 = 1)
  const Fr delta = fr_delta();
  std::vector<const void*> vals, sigs;
  std::vector<void*> zs;
  std::vector<Fr> bd;
  for (uint32_t j = 0; j < na; j++) {
    vals.push_back(advice[j]->p);
    sigs.push_back(pk.permutation.values[j]->p);
    zs.push_back(z[j]->p);
    bd.push_back(fr::mul(beta, fr::pow_u64(delta, j)));
  }
  check(h2mi_plonk_permutation_products_sparse_dev(vals.data(), sigs.data(), na, StandardPlonk::CS_DEGREE - 2, d.k(), u, beta.l, gamma.l,
                                                   (const uint64_t*)bd.data(), omega.l, pk.active_rows->p, pk.n_active, zs.data(), nullptr),
        "permutation_products");
  std::vector<Fr> zblind = uniform_fr(seed + 2, (size_t)na * bf);
  {
    std::vector<void*> cells;
    for (uint32_t m = 0; m < na; m++)
      for (uint32_t r = 0; r < bf; r++) cells.push_back((char*)z[m]->p + (size_t)(u + 1 + r) * 32);
    check(h2mi_fr_patch_cells_dev(cells.data(), (const uint64_t*)zblind.data(), cells.size(), nullptr), "z blinding rows");
  }
  {
    std::vector<const void*> cols;
    for (uint32_t m = 0; m < na; m++) cols.push_back(z[m]->p);
    commit_phase(params.g_lagrange_handle(), cols, /*sparse=*/true);
  }
  check(h2mi_msm_flush(), "flush");
  // coefficient / extended forms of z, queued behind the commitments on the library stream.  (Round 4 tried the side stream, so that the
  // read-back of the phase's points would not queue behind them: at 2^16 rows the proof got 0.2 ms SLOWER — the transforms then run beside
  // the commitments' partition and bucket-reduction chain, whose latency is what the phase waits for, and slow it: DESIGN 4.5)
  for (uint32_t j = 0; j < na; j++) to_poly_and_coset_into(d, *z[j], *z_polys[j], *z_cosets[j]);
  check(h2mi_stream_wait(nullptr, ws.side), "stream_wait");  // evaluate_h and the openings read the advice forms
  write_phase_points(na + 1);
  mark(1);
  const Fr y = tr.squeeze_challenge();

  // quotient: evaluate_h on the extended coset (divided by X^n - 1), back to coefficients, commit the pieces
  DeviceVec& h = *ws.h;
  {
    h2mi_standard_plonk_cosets cs;
    for (int i = 0; i < 3; i++) {
      cs.advice[i] = advice_cosets[i]->p;
      cs.sigma[i] = pk.permutation.cosets[i]->p;
      cs.z[i] = z_cosets[i]->p;
    }
    for (int i = 0; i < 5; i++) cs.fixed[i] = pk.fixed.cosets[i]->p;
    cs.l0 = pk.l0->p;
    cs.l_last = pk.l_last->p;
    cs.l_active = pk.l_active->p;
    const Fr& zeta = d.get_g_coset();
    const std::vector<Fr>& t_inv = d.t_inv();  // (X^n - 1)^-1 on the coset: cached in the domain
    check(h2mi_plonk_evaluate_h_standard_dev(&cs, d.k(), d.extended_k(), bf, beta.l, gamma.l, y.l, delta.l, zeta.l, d.get_extended_omega().l,
                                             (const uint64_t*)t_inv.data(), h.p, nullptr), "evaluate_h");
    check(h2mi_ntt_bn254_fr_dev(h.p, d.extended_k(), d.get_extended_omega_inv().l, nullptr, nullptr, nullptr), "extended_to_coeff");
    check(h2mi_fr_scale_powers_dev(h.p, ext, d.get_g_coset_inv().l, d.get_extended_ifft_divisor().l, nullptr), "distribute_powers_zeta");
  }
  const uint32_t pieces = StandardPlonk::CS_DEGREE - 1;
  {
    std::vector<const void*> cols;
    for (uint32_t i = 0; i < pieces; i++) cols.push_back((char*)h.p + (size_t)i * n * 32);
    commit_phase(params.g_handle(), cols, /*sparse=*/false, /*inorder=*/true);
  }
  write_phase_points(pieces);
  mark(2);
  const Fr x = tr.squeeze_challenge();
  const Fr xn = fr::pow_u64(x, n);

  // evaluations
  auto rot = [&](int64_t r) { return fr::mul(x, pow_signed(omega, omega_inv, r)); };
  const Fr x_next = rot(1), x_last = rot(-(int64_t)(bf + 1));
  DeviceVec& h_poly = *ws.h_poly;
  {
    const void* ptrs[2] = {h.p, (char*)h.p + n * 32};
    Fr sc[2] = {fr::ONE, xn};
    check(h2mi_fr_lincomb_dev(ptrs, (const uint64_t*)sc, pieces, n, h_poly.p, nullptr), "h_poly");
  }
  struct Q {
    const DeviceVec* poly;
    Fr point;
  };
  std::vector<Q> written;
  for (uint32_t c = 0; c < na; c++) written.push_back({advice_polys[c].get(), x});
  for (uint32_t c = 0; c < StandardPlonk::N_FIXED; c++) written.push_back({pk.fixed.polys[c].get(), x});
  written.push_back({&random_poly, x});
  for (uint32_t c = 0; c < na; c++) written.push_back({pk.permutation.polys[c].get(), x});
  for (uint32_t i = 0; i < na; i++) {
    written.push_back({z_polys[i].get(), x});
    written.push_back({z_polys[i].get(), x_next});
    if (i + 1 < na) written.push_back({z_polys[i].get(), x_last});
  }
  std::vector<Q> todo = written;
  todo.push_back({&h_poly, x});  // opened but not written
  DeviceVec& evals = *ws.evals;
  std::vector<size_t> slot(todo.size());
  size_t next_slot = 0;
  std::vector<Fr> distinct;
  for (const Q& q : todo)
    if (!contains(distinct, q.point)) distinct.push_back(q.point);
  {  // every evaluation in one call: groups by distinct point (h2mi_fr_eval_polys_multi_dev)
    std::vector<const void*> polys;
    std::vector<size_t> counts;
    for (const Fr& pt : distinct) {
      size_t cnt = 0;
      for (size_t i = 0; i < todo.size(); i++)
        if (todo[i].point == pt) {
          slot[i] = next_slot + cnt++;
          polys.push_back(todo[i].poly->p);
        }
      counts.push_back(cnt);
      next_slot += cnt;
    }
    check(h2mi_fr_eval_polys_multi_dev(polys.data(), counts.data(), (const uint64_t*)distinct.data(), distinct.size(), n, evals.p, nullptr), "eval");
  }
  std::vector<Fr> ev(todo.size());
  check(h2mi_memcpy_d2h(ev.data(), evals.p, todo.size() * 32), "d2h");
  for (size_t i = 0; i < written.size(); i++) tr.write_scalar(ev[slot[i]]);
  mark(3);
  auto value_of = [&](const DeviceVec* poly, const Fr& pt) {
    for (size_t i = 0; i < todo.size(); i++)
      if (todo[i].poly == poly && todo[i].point == pt) return ev[slot[i]];
    throw Error(H2MI_EINVAL, "query without an evaluation");
  };

  // queries in create_proof's order, then SHPLONK
  std::vector<ProverQuery> queries;
  auto q = [&](const DeviceVec* poly, const Fr& pt) { queries.push_back({poly, pt, value_of(poly, pt)}); };
  for (uint32_t c = 0; c < na; c++) q(advice_polys[c].get(), x);
  for (uint32_t i = 0; i < na; i++) {
    q(z_polys[i].get(), x);
    q(z_polys[i].get(), x_next);
  }
  for (uint32_t i = na - 1; i-- > 0;) q(z_polys[i].get(), x_last);
  for (uint32_t c = 0; c < StandardPlonk::N_FIXED; c++) q(pk.fixed.polys[c].get(), x);
  for (uint32_t c = 0; c < na; c++) q(pk.permutation.polys[c].get(), x);
  q(&h_poly, x);
  q(&random_poly, x);
  ShplonkScratch scratch{ws.nx.get(), ws.tmp.get(), ws.h_x.get(), ws.l_x.get(), ws.h2_x.get(), &ws.shplonk_q, &ws.shplonk_s, ws.lanes->lanes()};
  int shplonk_commit = 0;
  shplonk_create_proof(n, tr, queries, [&](DeviceVec& poly) {
    // a lone commitment, read back at once: in order on one stream, nothing deferred
    check(h2mi_msm_bn254_g1_inorder_dev(params.g_handle(), poly.p, n, points.p, nullptr), "commit");
    write_phase_points(1);
    mark(4 + shplonk_commit++);
  }, scratch);
}

}  // namespace plonk
}  // namespace h2mi
