// h2mi_plonk.hpp — C++17 caller of the library's prover (h2mi_prover.h) for the reference's StandardPlonk circuit, with the
// crate's names: keygen_vk / keygen_pk / create_proof.
//
// Mirrors what the reference calls (examples/standard_plonk.rs:29-50):
//     let params = ParamsKZG::<Bn256>::setup(k, OsRng);
//     let vk = keygen_vk(&params, &circuit)?;  let pk = keygen_pk(&params, vk, &circuit)?;
//     let mut transcript = Blake2bWrite::<_, _, Challenge255<_>>::init(vec![]);
//     create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<'_, Bn256>, Challenge255<G1Affine>, _, _, _>(
//         &params, &pk, &[circuit], &[&[]], OsRng, &mut transcript)?;
//     let proof = transcript.finalize();
// with the same argument meaning and failure behaviour (h2mi::Error where the crate panics / returns Err).
//
// What lives HERE is what lives in the caller of a Rust fork as well: the circuit's synthesize() (witness cells, fixed cells, copy
// constraints: reference src/circuits/standard_plonk.rs:58-114), the Blake2b transcript and the verifying key's transcript_repr.
// Everything between two transcript challenges is ONE call into libh2mi.so (`drive_proof` below: seven phase calls) — the
// orchestration that used to be restated in this header and in the Python host is csrc/h2mi_prover.cpp now.
// rng: the reference passes OsRng; here `seed` (< 2^32) is handed to the library's counter-based streams (h2mi_prover.h).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <map>
#include <memory>
#include <optional>
#include <utility>

#include "h2mi.hpp"
#include "h2mi_prover.h"
#include "h2mi_transcript.hpp"

namespace h2mi {
namespace plonk {

inline Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
inline Fr to_canonical(const Fr& a) { return fr::mul(a, Fr{{1, 0, 0, 0}}); }  // a R^-1: the integer behind the Montgomery form

// ---- the circuit (reference src/circuits/standard_plonk.rs) -------------------------------------------------------
typedef std::pair<uint32_t, uint32_t> Cell;  // (column within the permutation argument, row)
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
  // StandardPlonkConfig::configure (src/circuits/standard_plonk.rs:29-48) as the numbers create_proof reads off the constraint system
  static h2mi_constraint_system constraint_system(uint32_t k) {
    h2mi_constraint_system cs;
    std::memset(&cs, 0, sizeof(cs));
    cs.k = k;
    cs.n_advice = N_ADVICE;
    cs.n_fixed = N_FIXED;
    cs.degree = CS_DEGREE;
    cs.blinding_factors = BLINDING_FACTORS;
    cs.gates = H2MI_GATES_STANDARD_PLONK;
    cs.n_perm = 3;  // enable_equality(a), (b), (c)
    for (uint32_t j = 0; j < 3; j++) cs.perm_columns[j] = {H2MI_COL_ADVICE, j};
    cs.n_advice_queries = 3;
    for (uint32_t j = 0; j < 3; j++) cs.advice_queries[j] = {j, 0};
    cs.n_fixed_queries = 5;
    for (uint32_t j = 0; j < 5; j++) cs.fixed_queries[j] = {j, 0};
    return cs;
  }
};

// sparse cells of a column as the C ABI takes them (the vectors own the memory the descriptor points into)
struct ColumnCells {
  std::vector<uint32_t> rows;
  std::vector<Fr> values;
  ColumnCells() {}
  explicit ColumnCells(const std::map<uint32_t, Fr>& cells) {
    for (const auto& kv : cells) {
      rows.push_back(kv.first);
      values.push_back(kv.second);
    }
  }
  h2mi_column_cells view() const { return {rows.empty() ? nullptr : rows.data(), (const uint64_t*)values.data(), values.size(), 0}; }
  h2mi_column_cells dense_view() const { return {nullptr, (const uint64_t*)values.data(), values.size(), 0}; }  // rows 0 .. count - 1
};

// ---- keys ---------------------------------------------------------------------------------------------------------
struct VerifyingKey {
  uint32_t k = 0, cs_degree = 0;
  std::vector<G1Affine> fixed_commitments, permutation_commitments;
  Fr transcript_repr;
  // stand-in for the crate's pinned-Debug-text hash (not reproducible without the crate): the same Blake2b-512 /
  // "Halo2-Verify-Key" over k, the degree and the compressed commitments.  A fork replaces this one function; the library
  // never sees the value.
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

// keygen through the library for any constraint system: fixed cells + copy constraints in, key handle + vk commitments out
struct KeygenInput {
  std::vector<ColumnCells> fixed;
  std::vector<uint32_t> copies;  // 4 per constrain_equal: (left column, left row, right column, right row), columns within the permutation
};
struct PkHandle {  // RAII over h2mi_pk_t
  h2mi_pk_t h = nullptr;
  PkHandle() {}
  PkHandle(const PkHandle&) = delete;
  PkHandle& operator=(const PkHandle&) = delete;
  ~PkHandle() {
    if (h) h2mi_prover_pk_release(h);
  }
};
inline void run_keygen(const h2mi_constraint_system& cs, const poly::kzg::ParamsKZG& params, const KeygenInput& in, unsigned flags, PkHandle& pk,
                       VerifyingKey& vk) {
  std::vector<h2mi_column_cells> fixed;
  for (const ColumnCells& c : in.fixed) fixed.push_back(c.view());
  check(h2mi_prover_keygen(&cs, params.g_lagrange_handle(), fixed.data(), in.copies.data(), in.copies.size() / 4, flags, &pk.h), "keygen");
  vk.k = cs.k;
  vk.cs_degree = cs.degree;
  vk.fixed_commitments.resize(cs.n_fixed);
  vk.permutation_commitments.resize(cs.n_perm);
  check(h2mi_prover_vk_commitments(pk.h, (uint64_t*)vk.fixed_commitments.data(), (uint64_t*)vk.permutation_commitments.data()), "vk commitments");
  vk.compute_transcript_repr();
}

struct ProvingKey {
  VerifyingKey vk;
  PkHandle pk;
  const VerifyingKey& get_vk() const { return vk; }
};

namespace detail {
inline KeygenInput standard_plonk_keygen_input() {
  const Synthesis syn = StandardPlonk().synthesize();  // without_witnesses()
  KeygenInput in;
  for (const auto& col : syn.fixed) in.fixed.push_back(ColumnCells(col));
  for (const auto& c : syn.copies) in.copies.insert(in.copies.end(), {c.first.first, c.first.second, c.second.first, c.second.second});
  return in;
}
}  // namespace detail

inline VerifyingKey keygen_vk(const poly::kzg::ParamsKZG& params, const StandardPlonk& circuit) {
  (void)circuit;
  VerifyingKey vk;
  PkHandle tmp;
  run_keygen(StandardPlonk::constraint_system(params.k()), params, detail::standard_plonk_keygen_input(), H2MI_KEYGEN_VK_ONLY, tmp, vk);
  return vk;
}
inline std::unique_ptr<ProvingKey> keygen_pk(const poly::kzg::ParamsKZG& params, const VerifyingKey& vk, const StandardPlonk& circuit) {
  (void)circuit;
  std::unique_ptr<ProvingKey> pk(new ProvingKey);
  run_keygen(StandardPlonk::constraint_system(params.k()), params, detail::standard_plonk_keygen_input(), 0, pk->pk, pk->vk);
  if (pk->vk.to_bytes() != vk.to_bytes()) throw Error(H2MI_EINVAL, "keygen_pk: the verifying key belongs to another circuit or SRS");
  return pk;
}

// ---- one prover (device buffers, streams) per proving key, reused from proof to proof (the reference's examples prove repeatedly
// against one pk: examples/linear_regression.rs:178-185) ----------------------------------------------------------------------------
struct ProverWorkspace {
  h2mi_prover_t prover = nullptr;
  h2mi_prover_counts counts{};
  // optional host-side phase clock (untraced: a kernel tracer distorts exactly the host-paced stretches one wants to see): when
  // time_phases is set, drive_proof adds the wall clock of each phase to phase_us[0 .. 6] = advice, lookups, products (z + random),
  // quotient (h pieces), evaluations, SHPLONK's first and second commitment
  bool time_phases = false;
  double phase_us[7] = {0, 0, 0, 0, 0, 0, 0};
  ProverWorkspace(const ProverWorkspace&) = delete;
  ProverWorkspace& operator=(const ProverWorkspace&) = delete;
  ProverWorkspace(const poly::kzg::ParamsKZG& params, const PkHandle& pk) {
    check(h2mi_prover_create(pk.h, params.g_handle(), params.g_lagrange_handle(), 0, params.n(), &prover), "prover_create");
    check(h2mi_prover_get_counts(prover, &counts), "prover counts");
  }
  ProverWorkspace(const poly::kzg::ParamsKZG& params, const ProvingKey& pk) : ProverWorkspace(params, pk.pk) {}
  ~ProverWorkspace() {
    if (prover) h2mi_prover_destroy(prover);
  }
};

// create_proof between the transcript's challenges: what the body of a fork's create_proof looks like.  The
// caller has already hashed vk.transcript_repr and the public inputs into `tr`.
inline void drive_proof(ProverWorkspace& ws, const std::vector<h2mi_column_cells>& advice, const std::vector<Fr>& instance, uint64_t seed,
                        transcript::Blake2bWrite& tr) {
  h2mi_prover_t p = ws.prover;
  const h2mi_prover_counts& c = ws.counts;
  std::vector<G1Affine> pts(std::max({c.advice, c.lookups, c.products, c.quotient, 1u}));
  auto write_points = [&](size_t k) {
    for (size_t i = 0; i < k; i++) tr.write_point(pts[i]);  // throws on the identity, as the crate's transcript does
  };
  auto t0 = std::chrono::steady_clock::now();
  auto mark = [&](int phase) {
    if (!ws.time_phases) return;
    const auto t = std::chrono::steady_clock::now();
    ws.phase_us[phase] += std::chrono::duration<double, std::micro>(t - t0).count();
    t0 = t;
  };
  check(h2mi_prover_advice(p, advice.data(), (const uint64_t*)instance.data(), instance.size(), seed, (uint64_t*)pts.data()), "advice");
  write_points(c.advice);
  mark(0);
  const Fr theta = tr.squeeze_challenge();  // drawn even without lookups
  if (c.lookups) {
    int rc = h2mi_prover_lookups(p, theta.l, (uint64_t*)pts.data());
    if (rc == H2MI_EUNSAT) throw Error(rc, "lookup input not in the table (ConstraintSystemFailure)");
    check(rc, "lookups");
    write_points(c.lookups);
  }
  mark(1);
  const Fr beta = tr.squeeze_challenge(), gamma = tr.squeeze_challenge();
  check(h2mi_prover_products(p, beta.l, gamma.l, (uint64_t*)pts.data()), "products");
  write_points(c.products);
  mark(2);
  const Fr y = tr.squeeze_challenge();
  check(h2mi_prover_quotient(p, y.l, (uint64_t*)pts.data()), "quotient");
  write_points(c.quotient);
  mark(3);
  const Fr x = tr.squeeze_challenge();
  std::vector<Fr> evals(c.evaluations);
  check(h2mi_prover_evaluations(p, x.l, (uint64_t*)evals.data()), "evaluations");
  for (const Fr& e : evals) tr.write_scalar(e);
  mark(4);
  const Fr sy = tr.squeeze_challenge(), sv = tr.squeeze_challenge();  // ProverSHPLONK: y, v
  check(h2mi_prover_shplonk_quotient(p, sy.l, sv.l, (uint64_t*)pts.data()), "shplonk quotient");
  write_points(1);
  mark(5);
  const Fr su = tr.squeeze_challenge();
  check(h2mi_prover_shplonk_open(p, su.l, (uint64_t*)pts.data()), "shplonk open");
  write_points(1);
  mark(6);
}

// ---- create_proof ---------------------------------------------------------------------------------------------------
inline void create_proof(const poly::kzg::ParamsKZG& params, const ProvingKey& pk, const StandardPlonk& circuit, uint64_t seed,
                         transcript::Blake2bWrite& tr, ProverWorkspace* workspace = nullptr) {
  if (!circuit.x) throw Error(H2MI_EINVAL, "create_proof: the circuit has no witness (Value::unknown())");
  std::unique_ptr<ProverWorkspace> own;
  if (!workspace) {
    own.reset(new ProverWorkspace(params, pk));
    workspace = own.get();
  }
  tr.common_scalar(pk.vk.transcript_repr);  // vk.hash_into
  const Synthesis syn = circuit.synthesize();
  std::vector<ColumnCells> cells;
  for (const auto& col : syn.advice) cells.push_back(ColumnCells(col));
  std::vector<h2mi_column_cells> advice;
  for (const ColumnCells& c : cells) advice.push_back(c.view());
  drive_proof(*workspace, advice, {}, seed, tr);
}

}  // namespace plonk
}  // namespace h2mi
