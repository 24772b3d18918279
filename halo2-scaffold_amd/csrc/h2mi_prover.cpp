// h2mi_prover.cpp — the device-resident prover behind include/h2mi_prover.h: keygen_vk / keygen_pk and create_proof, cut at
// the transcript's challenges (reference examples/standard_plonk.rs:33-34,40-50; src/scaffold.rs:284,287,322-331).
//
// Host code only: every pass over a vector is one of the library's own *_dev entry points (h2mi.h), called here in
// create_proof's order [halo2_proofs v2023_02_02 plonk/{keygen,prover}.rs, plonk/permutation/*, plonk/lookup/*, plonk/vanishing/*,
// poly/kzg/multiopen/shplonk* — an un-vendored dependency of the reference (Cargo.toml:13), restated from memory; the oracle
// (oracle/prover.py, oracle/flex.py) is the same restatement and the parity tests compare proof bytes].  What this file decides is
// the SCHEDULE: which stream a launch goes to, when queued bucket reductions are flushed, which commitments share one set of
// launches — the part rounds 2-4 tuned on the two former hosts (include/h2mi_plonk.hpp, halo2-scaffold_amd/prover.py + flex.py),
// which are now thin callers of these phases.
#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <vector>

#include "../../include/h2mi_prover.h"
#include "h2mi_hostmath.hpp"

namespace {

using namespace h2mi;
using namespace h2mi::plonk;
using arithmetic::DeviceVec;
typedef std::unique_ptr<DeviceVec> Dev;

Dev vec(size_t count) { return Dev(new DeviceVec(count)); }
Dev zeros(size_t count) {
  Dev d = vec(count);
  check(h2mi_memset_zero(d->p, count * 32), "zero");
  return d;
}
inline void* at(const DeviceVec& d, size_t row) { return (char*)d.p + row * 32; }

// cells that travel in kernel arguments (h2mi_fr_patch_cells_dev: one launch per 64), collected across the columns of a phase
struct PatchList {
  std::vector<void*> cells;
  std::vector<Fr> vals;
  void add(const DeviceVec& d, size_t row, const Fr& v) {
    cells.push_back(at(d, row));
    vals.push_back(v);
  }
  void flush() {
    if (!cells.empty()) check(h2mi_fr_patch_cells_dev(cells.data(), (const uint64_t*)vals.data(), cells.size(), nullptr), "patch_cells");
    cells.clear();
    vals.clear();
  }
};

Fr cell_value(const h2mi_column_cells& c, size_t i) {
  Fr v;
  std::memcpy(v.l, c.values + 4 * i, 32);
  if (!(c.flags & H2MI_CELLS_CANONICAL)) return v;
  for (int j = 3; j >= 0; j--) {
    if (v.l[j] < fr::MODULUS[j]) return fr::mul(v, fr::R2);
    if (v.l[j] > fr::MODULUS[j]) break;
  }
  throw Error(H2MI_EINVAL, "a cell value is not reduced modulo r");
}

// a column of n rows: zero, then the assigned cells.  Short runs ride in the phase's patch launch, long dense runs are one
// upload (canonical values converted where they land), long scattered runs are staged over their span.
void fill_column(DeviceVec& d, size_t n, const h2mi_column_cells& c, size_t row_limit, PatchList& pl) {
  check(h2mi_memset_zero(d.p, n * 32), "zero");
  if (c.count == 0) return;
  if (!c.values) throw Error(H2MI_EINVAL, "column cells without values");
  if (!c.rows) {
    if (c.count > row_limit) throw Error(H2MI_ERANGE, "assignment reaches into the blinding rows (NotEnoughRowsAvailable)");
    if (c.count <= 16) {
      for (size_t i = 0; i < c.count; i++) pl.add(d, i, cell_value(c, i));
      return;
    }
    check(h2mi_memcpy_h2d(d.p, c.values, c.count * 32), "column cells");
    if (c.flags & H2MI_CELLS_CANONICAL) {
      uint64_t bad = 0;
      check(h2mi_fe_from_repr_dev(1, d.p, c.count, d.p, &bad), "from_repr");
      if (bad) throw Error(H2MI_EINVAL, "a cell value is not reduced modulo r");
    }
    return;
  }
  uint32_t lo = c.rows[0], hi = c.rows[0];
  for (size_t i = 0; i < c.count; i++) {
    lo = std::min(lo, c.rows[i]);
    hi = std::max(hi, c.rows[i]);
  }
  if (hi >= row_limit) throw Error(H2MI_ERANGE, "assignment reaches into the blinding rows (NotEnoughRowsAvailable)");
  if (c.count <= 4096) {
    for (size_t i = 0; i < c.count; i++) pl.add(d, c.rows[i], cell_value(c, i));
    return;
  }
  std::vector<Fr> stage((size_t)hi - lo + 1, fr_zero());
  for (size_t i = 0; i < c.count; i++) stage[c.rows[i] - lo] = cell_value(c, i);
  check(h2mi_memcpy_h2d(at(d, lo), stage.data(), stage.size() * 32), "column cells");
}

void to_poly_and_coset(const poly::EvaluationDomain& dom, const DeviceVec& lagr, DeviceVec& poly, DeviceVec& coset, h2mi_stream_t stream = nullptr) {
  const size_t n = (size_t)1 << dom.k();
  check(h2mi_ntt_bn254_fr_oop_dev(lagr.p, n, poly.p, dom.k(), dom.get_omega_inv().l, nullptr, dom.get_ifft_divisor().l, stream), "lagrange_to_coeff");
  check(h2mi_ntt_bn254_fr_oop_dev(poly.p, n, coset.p, dom.extended_k(), dom.get_extended_omega().l, dom.get_g_coset().l, nullptr, stream),
        "coeff_to_extended");
}

// commit device-resident columns against a registered base set -> affine points on the host (keygen: fixed and sigma columns)
std::vector<G1Affine> commit_points(uint64_t handle, const std::vector<Dev>& cols, size_t n) {
  const size_t k = cols.size();
  std::vector<G1Affine> pts(k);
  if (!k) return pts;
  DeviceVec out(3 * k), aff(2 * k);  // 96 B / 64 B per point
  for (size_t i = 0; i < k; i++) check(h2mi_msm_bn254_g1_dev(handle, cols[i]->p, n, (char*)out.p + 96 * i, nullptr), "commit");
  check(h2mi_join(), "join");
  check(h2mi_g1_batch_normalize_dev(out.p, k, aff.p, nullptr), "batch_normalize");
  check(h2mi_memcpy_d2h(pts.data(), aff.p, k * 64), "d2h");
  return pts;
}

// ---- the constraint system ---------------------------------------------------------------------------------------------------
void validate(const h2mi_constraint_system& cs) {
  auto bad = [](const char* what) { throw Error(H2MI_EINVAL, std::string("constraint system: ") + what); };
  if (cs.k == 0 || cs.k > H2MI_MAX_LOG_N) throw Error(H2MI_ERANGE, "constraint system: k");
  if (cs.degree < 3 || cs.degree > 9) bad("degree");
  if (((uint64_t)1 << cs.k) <= (uint64_t)cs.blinding_factors + 2) throw Error(H2MI_ERANGE, "constraint system: no usable rows");
  if (cs.n_advice == 0 || cs.n_advice > H2MI_MAX_ADVICE || cs.n_fixed > H2MI_MAX_FIXED || cs.n_instance > 1) bad("column counts");
  if (cs.n_perm > H2MI_MAX_PERM || cs.n_lookups > H2MI_MAX_LOOKUPS) bad("permutation / lookup counts");
  if (cs.n_advice_queries > H2MI_MAX_QUERIES || cs.n_fixed_queries > H2MI_MAX_QUERIES) bad("query counts");
  for (uint32_t j = 0; j < cs.n_perm; j++) {
    const h2mi_column& c = cs.perm_columns[j];
    const uint32_t lim = c.kind == H2MI_COL_ADVICE ? cs.n_advice : c.kind == H2MI_COL_FIXED ? cs.n_fixed : c.kind == H2MI_COL_INSTANCE ? cs.n_instance : 0;
    if (c.index >= lim) bad("permutation column");
  }
  for (uint32_t i = 0; i < cs.n_advice_queries; i++)
    if (cs.advice_queries[i].column >= cs.n_advice) bad("advice query");
  for (uint32_t i = 0; i < cs.n_fixed_queries; i++)
    if (cs.fixed_queries[i].column >= cs.n_fixed) bad("fixed query");
  if (cs.gates == H2MI_GATES_STANDARD_PLONK) {
    // the circuit of reference src/circuits/standard_plonk.rs:29-48, which the specialised quotient kernel evaluates
    if (cs.n_advice != 3 || cs.n_fixed != 5 || cs.n_instance != 0 || cs.n_lookups != 0 || cs.degree != 3 || cs.n_perm != 3) bad("not the StandardPlonk shape");
    for (uint32_t j = 0; j < 3; j++)
      if (cs.perm_columns[j].kind != H2MI_COL_ADVICE || cs.perm_columns[j].index != j) bad("StandardPlonk's permutation runs over a, b, c");
  } else if (cs.gates == H2MI_GATES_FLEX_VERTICAL) {
    if (cs.n_gates == 0 || cs.n_gates > H2MI_MAX_GATES) bad("gate count");
    for (uint32_t g = 0; g < cs.n_gates; g++)
      if (cs.gate_advice[g] >= cs.n_advice || cs.gate_selector[g] >= cs.n_fixed) bad("gate columns");
    if (cs.n_perm && cs.degree - 2 > 3) bad("permutation chunks longer than three columns");
    for (uint32_t l = 0; l < cs.n_lookups; l++) {
      const h2mi_lookup& lk = cs.lookups[l];
      if (lk.input.kind != H2MI_COL_ADVICE || lk.input.index >= cs.n_advice || lk.table_fixed >= cs.n_fixed) bad("lookup columns");
      if (lk.selector_fixed >= (int32_t)cs.n_fixed || lk.selector_fixed < -1) bad("lookup selector");
    }
  } else {
    bad("unknown gate shape");
  }
  uint32_t ext_k = cs.k;
  while (((uint64_t)1 << ext_k) < ((uint64_t)1 << cs.k) * (cs.degree - 1)) ext_k++;
  if (ext_k - cs.k > 4 || ext_k > H2MI_MAX_LOG_N) throw Error(H2MI_ERANGE, "constraint system: extended domain");
}

struct Table {  // a lookup table's distinct usable values in ascending canonical order, for the device's counting sort
  Dev sorted, sorted_mont, mult;
  uint32_t n_unique = 0;
};

}  // namespace

struct h2mi_pk_s {
  h2mi_constraint_system cs;
  poly::EvaluationDomain domain;
  size_t n, ext;
  uint32_t u, chunk, n_sets;
  bool vk_only = false;
  int users = 0;  // provers created against this key and not yet destroyed
  Fr delta;
  std::vector<Dev> fixed_values, fixed_polys, fixed_cosets, sigma_values, sigma_polys, sigma_cosets;
  Dev l0, l_last, l_active, active_rows;
  uint32_t n_active = 0;
  Table tables[H2MI_MAX_LOOKUPS];
  std::vector<G1Affine> fixed_commitments, permutation_commitments;
  explicit h2mi_pk_s(const h2mi_constraint_system& c)
      : cs(c), domain(c.degree, c.k), n((size_t)1 << c.k), ext(domain.extended_len()), u((uint32_t)n - (c.blinding_factors + 1)),
        chunk(c.degree - 2), n_sets(c.n_perm ? (c.n_perm + c.degree - 3) / (c.degree - 2) : 0), delta(fr_delta()) {}
};

namespace {

std::mutex g_reg_mu;
std::set<const void*> g_live_pks, g_live_provers;
template <class T>
bool alive(const std::set<const void*>& s, T p) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  return p && s.count((const void*)p);
}

void build_table(h2mi_pk_s& pk, Table& t, const h2mi_column_cells& cells) {
  // values on the usable rows, canonical and Montgomery; unassigned usable rows hold zero
  std::vector<std::pair<Fr, Fr>> vals;  // (canonical, montgomery)
  vals.reserve(cells.count);
  for (size_t i = 0; i < cells.count; i++) {
    const uint32_t row = cells.rows ? cells.rows[i] : (uint32_t)i;
    if (row >= pk.u) throw Error(H2MI_ERANGE, "lookup table larger than the usable rows (LOOKUP_BITS must be below DEGREE)");
    const Fr m = cell_value(cells, i);
    vals.push_back({to_canonical(m), m});
  }
  const uint64_t zeros_extra = pk.u - cells.count;
  auto less = [](const std::pair<Fr, Fr>& a, const std::pair<Fr, Fr>& b) {
    for (int i = 3; i >= 0; i--)
      if (a.first.l[i] != b.first.l[i]) return a.first.l[i] < b.first.l[i];
    return false;
  };
  if (!std::is_sorted(vals.begin(), vals.end(), less)) std::sort(vals.begin(), vals.end(), less);
  std::vector<Fr> canon, mont;
  std::vector<uint32_t> mult;
  if (zeros_extra && (vals.empty() || !(vals[0].first == fr_zero()))) {
    canon.push_back(fr_zero());
    mont.push_back(fr_zero());
    mult.push_back(0);
  }
  for (size_t i = 0; i < vals.size(); i++) {
    if (!canon.empty() && canon.back() == vals[i].first) {
      mult.back()++;
    } else {
      canon.push_back(vals[i].first);
      mont.push_back(vals[i].second);
      mult.push_back(1);
    }
  }
  if (zeros_extra) mult[0] += (uint32_t)zeros_extra;  // zero sorts first
  t.n_unique = (uint32_t)mult.size();
  t.sorted = vec(mult.size());
  t.sorted_mont = vec(mult.size());
  t.mult = vec(mult.size() / 8 + 1);
  check(h2mi_memcpy_h2d(t.sorted->p, canon.data(), canon.size() * 32), "table");
  check(h2mi_memcpy_h2d(t.sorted_mont->p, mont.data(), mont.size() * 32), "table");
  check(h2mi_memcpy_h2d(t.mult->p, mult.data(), mult.size() * 4), "table");
}

std::unique_ptr<h2mi_pk_s> keygen(const h2mi_constraint_system& cs, uint64_t g_lagrange, const h2mi_column_cells* fixed, const uint32_t* copies,
                                  size_t n_copies, unsigned flags) {
  validate(cs);
  std::unique_ptr<h2mi_pk_s> pkp(new h2mi_pk_s(cs));
  h2mi_pk_s& pk = *pkp;
  pk.vk_only = (flags & H2MI_KEYGEN_VK_ONLY) != 0;
  const poly::EvaluationDomain& dom = pk.domain;
  const size_t n = pk.n;
  const uint32_t u = pk.u, m = cs.n_perm;
  {
    uint64_t base_n = 0;
    check(h2mi_bases_info(g_lagrange, nullptr, nullptr, nullptr, &base_n), "g_lagrange handle");
    if (base_n < n) throw Error(H2MI_ERANGE, "keygen: the Lagrange SRS is shorter than 2^k");
  }
  PatchList pl;
  // fixed columns as synthesize() assigns them.  keygen's Assembly refuses a fixed cell or a copy constraint outside the usable rows
  // (Error::NotEnoughRowsAvailable [RECALL plonk/keygen.rs assign_fixed / copy]): the permutation argument's product does not run over
  // the blinding rows, so a cycle through one would not be enforced and the proof would not verify
  for (uint32_t c = 0; c < cs.n_fixed; c++) {
    Dev d = vec(n);
    fill_column(*d, n, fixed[c], u, pl);
    pk.fixed_values.push_back(std::move(d));
  }
  pl.flush();
  // sigma_j[i] = DELTA^(j') omega^(i') for (j', i') = mapping[(j, i)]  (permutation/keygen.rs build_vk / build_pk)
  PermutationAssembly asm_;
  for (size_t i = 0; i < n_copies; i++) {
    const uint32_t* c = copies + 4 * i;
    if (c[0] >= m || c[2] >= m) throw Error(H2MI_EINVAL, "copy constraint on a column without equality enabled");
    if (c[1] >= u || c[3] >= u) throw Error(H2MI_ERANGE, "copy constraint beyond the usable rows (NotEnoughRowsAvailable)");
    asm_.copy(Cell(c[0], c[1]), Cell(c[2], c[3]));
  }
  if (m) {
    DeviceVec omega_pows(n);
    check(h2mi_fr_powers_dev(omega_pows.p, n, dom.get_omega().l, nullptr), "powers");
    std::vector<Fr> dpow(m, fr::ONE);
    for (uint32_t j = 1; j < m; j++) dpow[j] = fr::mul(dpow[j - 1], pk.delta);
    for (uint32_t j = 0; j < m; j++) {
      Dev d = vec(n);
      const void* ptrs[1] = {omega_pows.p};
      check(h2mi_fr_lincomb_dev(ptrs, dpow[j].l, 1, n, d->p, nullptr), "identity permutation");
      pk.sigma_values.push_back(std::move(d));
    }
    std::vector<uint32_t> pos;
    asm_.for_each([&](const Cell& from, const Cell& to) {
      if (from == to) return;
      pl.add(*pk.sigma_values[from.first], from.second, fr::mul(dpow[to.first], fr::pow_u64(dom.get_omega(), to.second)));
      if (from.second < u) pos.push_back((from.first / pk.chunk) * u + from.second);
    });
    pl.flush();
    check(h2mi_sync(), "sync");  // omega_pows is released below
    std::sort(pos.begin(), pos.end());
    pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
    pk.n_active = (uint32_t)pos.size();
    pk.active_rows = vec(pos.size() / 8 + 1);
    if (!pos.empty()) check(h2mi_memcpy_h2d(pk.active_rows->p, pos.data(), pos.size() * 4), "active rows");
  }
  pk.fixed_commitments = commit_points(g_lagrange, pk.fixed_values, n);
  pk.permutation_commitments = commit_points(g_lagrange, pk.sigma_values, n);
  if (pk.vk_only) return pkp;
  for (auto& col : pk.fixed_values) {
    Dev p = vec(n), e = vec(pk.ext);
    to_poly_and_coset(dom, *col, *p, *e);
    pk.fixed_polys.push_back(std::move(p));
    pk.fixed_cosets.push_back(std::move(e));
  }
  for (auto& col : pk.sigma_values) {
    Dev p = vec(n), e = vec(pk.ext);
    to_poly_and_coset(dom, *col, *p, *e);
    pk.sigma_polys.push_back(std::move(p));
    pk.sigma_cosets.push_back(std::move(e));
  }
  {  // l_0, l_last (row u), l_active = ones on the usable rows: extended-coset forms only
    Dev l0 = zeros(n), ll = zeros(n), la = vec(n), tmp = vec(n);
    pl.add(*l0, 0, fr::ONE);
    pl.add(*ll, u, fr::ONE);
    pl.flush();
    check(h2mi_fr_fill_dev(la->p, n, fr::ONE.l, nullptr), "fill");
    check(h2mi_memset_zero(at(*la, u), (n - u) * 32), "zero");
    pk.l0 = vec(pk.ext);
    pk.l_last = vec(pk.ext);
    pk.l_active = vec(pk.ext);
    to_poly_and_coset(dom, *l0, *tmp, *pk.l0);
    to_poly_and_coset(dom, *ll, *tmp, *pk.l_last);
    to_poly_and_coset(dom, *la, *tmp, *pk.l_active);
    check(h2mi_sync(), "sync");
  }
  for (uint32_t l = 0; l < cs.n_lookups; l++) build_table(pk, pk.tables[l], fixed[cs.lookups[l].table_fixed]);
  return pkp;
}

// ---- ProverSHPLONK (poly/kzg/multiopen/shplonk.rs construct_intermediate_sets + shplonk/prover.rs) -----------------------------
struct ProverQuery {
  const DeviceVec* poly;
  Fr point, eval;
};
struct RotationSet {
  std::vector<Fr> points;                                              // ascending canonical value (BTreeSet<Fr>)
  std::vector<std::pair<const DeviceVec*, std::vector<Fr>>> members;  // (polynomial, evaluation at each point)
};
// coefficient lists (low to high) of the Lagrange basis polynomials of `pts`: the part of an interpolation that depends on the
// points alone — once per rotation set, with ONE inversion
std::vector<std::vector<Fr>> lagrange_basis(const std::vector<Fr>& pts) {
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
std::vector<Fr> interpolate(const std::vector<std::vector<Fr>>& basis, const std::vector<Fr>& evals) {
  std::vector<Fr> out(basis.size(), fr_zero());
  for (size_t j = 0; j < basis.size(); j++)
    for (size_t i = 0; i < basis[j].size(); i++) out[i] = fr::add(out[i], fr::mul(basis[j][i], evals[j]));
  return out;
}
Fr horner(const std::vector<Fr>& c, const Fr& x) {
  Fr acc = fr_zero();
  for (size_t i = c.size(); i-- > 0;) acc = fr::add(fr::mul(acc, x), c[i]);
  return acc;
}
Fr vanishing_at(const std::vector<Fr>& roots, const Fr& z) {
  Fr acc = fr::ONE;
  for (const Fr& r : roots) acc = fr::mul(acc, fr::sub(z, r));
  return acc;
}
bool contains(const std::vector<Fr>& v, const Fr& x) { return std::find(v.begin(), v.end(), x) != v.end(); }
// out = sum_k scalars[k] polys[k]; beyond one launch's 24 operands the sum continues in place (out re-enters with weight one: the
// kernel reads and writes row i in one thread, and every partial sum is the canonical value, so the split changes no bit)
void lincomb(const std::vector<const DeviceVec*>& polys, const std::vector<Fr>& scalars, size_t n, DeviceVec& out, h2mi_stream_t stream = nullptr) {
  constexpr size_t MAX = 24;
  for (size_t k0 = 0; k0 < polys.size();) {
    std::vector<const void*> ptrs;
    std::vector<Fr> sc;
    if (k0) {
      ptrs.push_back(out.p);
      sc.push_back(fr::ONE);
    }
    while (k0 < polys.size() && ptrs.size() < MAX) {
      ptrs.push_back(polys[k0]->p);
      sc.push_back(scalars[k0]);
      k0++;
    }
    check(h2mi_fr_lincomb_dev(ptrs.data(), (const uint64_t*)sc.data(), ptrs.size(), n, out.p, stream), "lincomb");
  }
}
void add_head(DeviceVec& poly, const std::vector<Fr>& head, h2mi_stream_t stream = nullptr) {
  check(h2mi_fr_add_head_dev(poly.p, (const uint64_t*)head.data(), head.size(), stream), "add_head");
}
// out = src / prod (X - root); `out` must have been zeroed (the quotient has n - #roots coefficients, the rest stay zero).
// Two to four roots: ONE round of independent divisions weighted by the partial-fraction coefficients; otherwise a chain through
// tmp / tmp2.
void kate_chain(DeviceVec& src, size_t n, const std::vector<Fr>& roots, DeviceVec& tmp, DeviceVec& out, h2mi_stream_t stream, DeviceVec* tmp2) {
  if (roots.size() >= 2 && roots.size() <= 4) {
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

struct Forms {  // a column of the proof in Lagrange, coefficient and extended-coset form
  Dev value, poly, coset;
  void alloc(size_t n, size_t ext) {
    value = vec(n);
    poly = vec(n);
    coset = vec(ext);
  }
};

enum Phase { IDLE = 0, ADVICE, LOOKUPS, PRODUCTS, QUOTIENT, EVALUATIONS, SHPLONK_Q };

}  // namespace

struct h2mi_prover_s {
  h2mi_pk_s* pk;
  uint64_t g, gl;
  size_t lo, cnt;
  // sliced SRS
  void *d_partial = nullptr, *d_combined = nullptr;
  h2mi_combine_fn combine = nullptr;
  void* combine_ctx = nullptr;
  // buffers
  std::vector<Forms> advice, z;
  Dev instance, instance_poly, instance_coset;
  struct Lk {
    Dev input;  // selector * column rows (single-column form); otherwise the advice column itself is the input
    Forms a, s, z;
  } lk[H2MI_MAX_LOOKUPS];
  Dev random_poly, h, h_poly, points, evals;
  // SHPLONK: three lanes (the library stream and two side streams), each with its own scratch
  Dev nx[3], tmp[3], h_x, l_x, h2_x;
  std::vector<Dev> q, s;
  h2mi_stream_t lane[2] = {nullptr, nullptr}, side = nullptr;
  // state of the proof in flight
  Phase phase = IDLE;
  // blinding source: counter-based SplitMix64 streams of `seed` (reproducible; tests, benchmarks, the goldens), or — once a key is
  // set — ChaCha20 blocks under a 256-bit key, one block per scalar as Fr::random draws them, `seed` the per-proof nonce
  bool keyed = false;
  uint8_t key[32] = {0};
  uint64_t seed = 0;
  std::vector<Fr> instance_vals;
  bool advice_sparse = true;
  Fr beta, gamma;
  std::vector<ProverQuery> queries;
  size_t n_written = 0;
  std::vector<RotationSet> sets;
  std::vector<Fr> super_points;
  std::vector<std::vector<Fr>> remainders;
  Fr v;

  ~h2mi_prover_s() {
    for (h2mi_stream_t st : {lane[0], lane[1], side})
      if (st) h2mi_stream_destroy(st);
  }
  const DeviceVec& lookup_input(uint32_t l) const {
    const h2mi_lookup& d = pk->cs.lookups[l];
    return d.selector_fixed >= 0 ? *lk[l].input : *advice[d.input.index].value;
  }
  void* out_slot(size_t slot) const { return (char*)(d_partial ? d_partial : points->p) + 96 * slot; }
  const void* col(const DeviceVec& d, size_t offset_elems = 0) const { return (const char*)d.p + (offset_elems + lo) * 32; }

  void commit(bool lagrange, const void* column, size_t slot) {
    check(h2mi_msm_bn254_g1_dev(lagrange ? gl : g, column, cnt, out_slot(slot), nullptr), "commit");
  }
  // the commitments of one phase into slots slot0 ..: ONE call, so that small circuits get one set of launches for all of them.
  // sparse: the columns are mostly zeros or one repeated value; inorder: the group is everything its phase commits and is read back next
  void commit_phase(bool lagrange, const std::vector<const void*>& cols, size_t slot0, bool sparse, bool inorder) {
    const unsigned flags = (sparse ? H2MI_MSM_SPARSE : 0u) | (inorder ? H2MI_MSM_INORDER : 0u);
    check(h2mi_msm_bn254_g1_phase_dev(lagrange ? gl : g, cols.data(), cols.size(), cnt, out_slot(slot0), flags, nullptr), "commit");
  }
  // fetch the k Jacobian results of a phase (the copy joins the MSM pipeline; sliced SRS: the caller's all-gather + fold first),
  // G1::batch_normalize on the host (microseconds for a handful of points; a lone device thread takes 0.3 ms)
  void read_points(size_t k, uint64_t* out) {
    if (!k) return;
    if (!out) throw Error(H2MI_EINVAL, "points_out");
    std::vector<G1> jac(k);
    if (d_partial) {
      check(h2mi_join(), "join");
      if (combine(combine_ctx, k)) throw Error(H2MI_EHIP, "the combiner failed");
      check(h2mi_memcpy_d2h(jac.data(), d_combined, k * 96), "d2h");
    } else {
      check(h2mi_memcpy_d2h(jac.data(), points->p, k * 96), "d2h");
    }
    const std::vector<G1Affine> aff = normalize_host_batch(jac);
    std::memcpy(out, aff.data(), k * 64);
  }
  void forms(Forms& f, h2mi_stream_t stream) { to_poly_and_coset(pk->domain, *f.value, *f.poly, *f.coset, stream); }
  // `count` blinding scalars for purpose 1 .. 5 (advice rows, permutation products, the random polynomial, permuted lookup columns,
  // lookup products): the streams seed + purpose of the seeded generator, or ChaCha20 stream id (nonce << 3 | purpose) under the key
  std::vector<Fr> blinding(uint32_t purpose, size_t count) const {
    return keyed ? chacha_fr(key, (seed << 3) | purpose, count) : uniform_fr(seed + purpose, count);
  }
  void random_vector(uint32_t purpose, DeviceVec& out, size_t count) const {
    if (keyed) check(h2mi_fr_random_chacha_dev(out.p, count, key, (seed << 3) | purpose, 0, nullptr), "random_poly");
    else check(h2mi_fr_random_dev(out.p, count, seed + purpose, 0, nullptr), "random_poly");
  }
};

namespace {

// commitments of the largest phase: the result slots a prover (and a combiner's two buffers) must hold
uint32_t max_phase_points(const h2mi_pk_s& pk) {
  const h2mi_constraint_system& cs = pk.cs;
  return std::max({cs.n_advice, 2 * cs.n_lookups, pk.n_sets + cs.n_lookups + 1, cs.degree - 1, 8u});
}

std::unique_ptr<h2mi_prover_s> create_prover(h2mi_pk_s* pk, uint64_t g, uint64_t gl, size_t lo, size_t cnt) {
  if (pk->vk_only) throw Error(H2MI_EINVAL, "prover_create: the key was built with H2MI_KEYGEN_VK_ONLY");
  if (cnt == 0 || lo + cnt > pk->n) throw Error(H2MI_ERANGE, "prover_create: base slice");
  for (uint64_t h : {g, gl}) {
    uint64_t base_n = 0;
    check(h2mi_bases_info(h, nullptr, nullptr, nullptr, &base_n), "bases handle");
    if (base_n < cnt) throw Error(H2MI_ERANGE, "prover_create: base set shorter than the slice");
  }
  std::unique_ptr<h2mi_prover_s> pp(new h2mi_prover_s);
  h2mi_prover_s& p = *pp;
  p.pk = pk;
  p.g = g;
  p.gl = gl;
  p.lo = lo;
  p.cnt = cnt;
  const h2mi_constraint_system& cs = pk->cs;
  const size_t n = pk->n, ext = pk->ext;
  p.advice.resize(cs.n_advice);
  for (Forms& f : p.advice) f.alloc(n, ext);
  p.z.resize(pk->n_sets);
  for (Forms& f : p.z) f.alloc(n, ext);
  if (cs.n_instance) {
    p.instance = vec(n);
    p.instance_coset = vec(ext);
  }
  for (uint32_t l = 0; l < cs.n_lookups; l++) {
    if (cs.lookups[l].selector_fixed >= 0) p.lk[l].input = vec(n);
    p.lk[l].a.alloc(n, ext);
    p.lk[l].s.alloc(n, ext);
    p.lk[l].z.alloc(n, ext);
  }
  p.random_poly = vec(n);
  p.h = vec(ext);
  p.h_poly = vec(n);
  p.points = vec(3 * (size_t)max_phase_points(*pk));  // 96 B per commitment of the largest phase
  for (int i = 0; i < 3; i++) {
    p.nx[i] = vec(n);
    p.tmp[i] = vec(n);
  }
  p.h_x = vec(n);
  p.l_x = vec(n);
  p.h2_x = vec(n);
  for (int i = 0; i < 2; i++) check(h2mi_stream_create(&p.lane[i]), "stream_create");
  check(h2mi_stream_create(&p.side), "stream_create");
  return pp;
}

void require_phase(h2mi_prover_s& p, Phase want) {
  if (p.phase != want) {
    p.phase = IDLE;
    throw Error(H2MI_EINVAL, "prover phases out of order");
  }
}

// ---- phase 1: advice columns ------------------------------------------------------------------------------------------------
void phase_advice(h2mi_prover_s& p, const h2mi_column_cells* advice, const uint64_t* instance, size_t n_inst, uint64_t seed, uint64_t* points_out) {
  const h2mi_pk_s& pk = *p.pk;
  const h2mi_constraint_system& cs = pk.cs;
  const size_t n = pk.n;
  const uint32_t bf = cs.blinding_factors, u = pk.u, na = cs.n_advice;
  p.phase = IDLE;
  if (seed >> (p.keyed ? 61 : 32)) throw Error(H2MI_EINVAL, p.keyed ? "nonce must be below 2^61" : "seed must be below 2^32");
  if (n_inst && (!cs.n_instance || !instance)) throw Error(H2MI_EINVAL, "public inputs without an instance column");
  if (n_inst > u) throw Error(H2MI_ERANGE, "more public inputs than usable rows");
  p.seed = seed;
  p.instance_vals.resize(n_inst);
  if (n_inst) std::memcpy(p.instance_vals.data(), instance, n_inst * 32);
  PatchList pl;
  if (cs.n_instance) {
    h2mi_column_cells ic = {nullptr, instance, n_inst, 0};
    fill_column(*p.instance, n, ic, u, pl);
  }
  // witness cells + blinding rows.  Assigned cells and blinding rows of every column travel in ONE launch's arguments when the
  // columns are short (h2mi_fr_patch_cells_dev) instead of a dozen 32-byte copies in front of the phase's commitments
  const std::vector<Fr> blind = p.blinding(1, (size_t)na * (bf + 1));
  size_t assigned = 0;
  for (uint32_t j = 0; j < na; j++) {
    DeviceVec& colv = *p.advice[j].value;
    fill_column(colv, n, advice[j], u, pl);
    assigned = std::max(assigned, advice[j].count);
    for (uint32_t r = 0; r <= bf; r++) pl.add(colv, u + r, blind[(size_t)j * (bf + 1) + r]);
  }
  pl.flush();
  // the sparse promise (batched launches at every size) holds for the padded witness columns of the reference's circuits; a
  // column assigned on more than a quarter of its rows takes the pipelined loop
  p.advice_sparse = assigned * 4 <= n;
  {
    std::vector<const void*> cols;
    for (uint32_t j = 0; j < na; j++) cols.push_back(p.col(*p.advice[j].value));
    p.commit_phase(true, cols, 0, p.advice_sparse, /*inorder=*/true);
  }
  check(h2mi_msm_flush(), "flush");  // the bucket reductions start now, not when the host reaches the join
  // The coefficient / extended forms of the advice and instance columns depend on no challenge (create_proof computes them after
  // y): on the side stream they run beside the transcript round trips, the lookups' counting sorts and the grand products' chains
  // of small scans instead of queueing behind the next phases' commitments
  check(h2mi_stream_wait(p.side, nullptr), "stream_wait");
  for (uint32_t j = 0; j < na; j++) p.forms(p.advice[j], p.side);
  if (cs.n_instance) {
    if (n_inst <= 16) {  // a handful of public inputs: sum_r v_r * (l_0's coset rotated by r rows), no transform
      check(h2mi_plonk_instance_coset_dev(pk.l0->p, cs.k, pk.domain.extended_k(), instance, n_inst, p.instance_coset->p, p.side), "instance coset");
    } else {
      if (!p.instance_poly) p.instance_poly = vec(n);
      to_poly_and_coset(pk.domain, *p.instance, *p.instance_poly, *p.instance_coset, p.side);
    }
  }
  p.read_points(na, points_out);
  p.phase = ADVICE;
}

// ---- phase 2: lookups' permuted columns (plonk/lookup/prover.rs commit_permuted) -----------------------------------------------
void phase_lookups(h2mi_prover_s& p, uint64_t* points_out) {
  const h2mi_pk_s& pk = *p.pk;
  const h2mi_constraint_system& cs = pk.cs;
  const size_t n = pk.n;
  const uint32_t bf = cs.blinding_factors, u = pk.u, L = cs.n_lookups;
  require_phase(p, ADVICE);
  p.phase = IDLE;
  if (L) {
    const std::vector<Fr> lb = p.blinding(4, (size_t)2 * (bf + 1) * L);
    PatchList pl;
    std::vector<const void*> cols;
    for (uint32_t l = 0; l < L; l++) {
      const h2mi_lookup& d = cs.lookups[l];
      if (d.selector_fixed >= 0)  // the input expression's rows: q_lookup * a, zero wherever the selector is off
        check(h2mi_fr_mul_dev(pk.fixed_values[d.selector_fixed]->p, p.advice[d.input.index].value->p, n, p.lk[l].input->p, nullptr), "lookup input");
      DeviceVec &a_perm = *p.lk[l].a.value, &s_perm = *p.lk[l].s.value;
      uint64_t missing = 0;
      const Table& t = pk.tables[l];
      check(h2mi_plonk_lookup_permute_dev(p.lookup_input(l).p, t.sorted->p, t.sorted_mont->p, t.mult->p, t.n_unique, cs.k, u, a_perm.p, s_perm.p,
                                          &missing, nullptr),
            "lookup_permute");
      if (missing) throw Error(H2MI_EUNSAT, "lookup input not in the table (ConstraintSystemFailure)");
      const size_t o0 = (size_t)2 * (bf + 1) * l;
      for (uint32_t r = 0; r <= bf; r++) pl.add(a_perm, u + r, lb[o0 + r]);
      for (uint32_t r = 0; r <= bf; r++) pl.add(s_perm, u + r, lb[o0 + bf + 1 + r]);
      cols.push_back(p.col(a_perm));
      cols.push_back(p.col(s_perm));
    }
    pl.flush();
    p.commit_phase(true, cols, 0, /*sparse=*/false, /*inorder=*/true);  // the permuted pairs are all this phase commits
    check(h2mi_msm_flush(), "flush");
    check(h2mi_stream_wait(p.side, nullptr), "stream_wait");
    for (uint32_t l = 0; l < L; l++) {
      p.forms(p.lk[l].a, p.side);
      p.forms(p.lk[l].s, p.side);
    }
    p.read_points(2 * L, points_out);
  }
  p.phase = LOOKUPS;
}

// ---- phase 3: grand products + the vanishing argument's random polynomial ----------------------------------------------------
void phase_products(h2mi_prover_s& p, const Fr& beta, const Fr& gamma, uint64_t* points_out) {
  const h2mi_pk_s& pk = *p.pk;
  const h2mi_constraint_system& cs = pk.cs;
  const size_t n = pk.n;
  const uint32_t bf = cs.blinding_factors, u = pk.u, L = cs.n_lookups, m = cs.n_perm, n_sets = pk.n_sets;
  if (p.phase == ADVICE && L == 0) p.phase = LOOKUPS;  // nothing to permute: the lookup phase may be skipped
  require_phase(p, LOOKUPS);
  p.phase = IDLE;
  p.beta = beta;
  p.gamma = gamma;
  // the random polynomial's commitment is written after the grand products' but depends on nothing: queued first, the one dense
  // MSM of this phase accumulates beside their latency-bound scans.  Result slot: where the transcript expects it.
  p.random_vector(3, *p.random_poly, n);
  p.commit(false, p.col(*p.random_poly), n_sets + L);
  // permutation argument: every set in one device pass, over the copy constraints' support when that is sparse
  const bool perm_sparse = (uint64_t)pk.n_active * 8 <= (uint64_t)n_sets * u;
  if (m) {
    std::vector<const void*> vals, sigs;
    std::vector<void*> zptr;
    std::vector<Fr> bd;
    Fr dp = fr::ONE;
    for (uint32_t j = 0; j < m; j++) {
      const h2mi_column& c = cs.perm_columns[j];
      vals.push_back(c.kind == H2MI_COL_ADVICE ? p.advice[c.index].value->p : c.kind == H2MI_COL_INSTANCE ? p.instance->p : pk.fixed_values[c.index]->p);
      sigs.push_back(pk.sigma_values[j]->p);
      bd.push_back(fr::mul(beta, dp));
      dp = fr::mul(dp, pk.delta);
    }
    for (uint32_t s = 0; s < n_sets; s++) zptr.push_back(p.z[s].value->p);
    if (perm_sparse)
      check(h2mi_plonk_permutation_products_sparse_dev(vals.data(), sigs.data(), m, pk.chunk, cs.k, u, beta.l, gamma.l, (const uint64_t*)bd.data(),
                                                       pk.domain.get_omega().l, pk.active_rows->p, pk.n_active, zptr.data(), nullptr),
            "permutation_products");
    else
      check(h2mi_plonk_permutation_products_dev(vals.data(), sigs.data(), m, pk.chunk, cs.k, u, beta.l, gamma.l, (const uint64_t*)bd.data(),
                                                pk.domain.get_omega().l, zptr.data(), nullptr),
            "permutation_products");
    const std::vector<Fr> zblind = p.blinding(2, (size_t)n_sets * bf);
    PatchList pl;
    for (uint32_t s = 0; s < n_sets; s++)
      for (uint32_t r = 0; r < bf; r++) pl.add(*p.z[s].value, u + 1 + r, zblind[(size_t)s * bf + r]);
    pl.flush();
  }
  std::vector<const void*> zcols;
  for (uint32_t s = 0; s < n_sets; s++) zcols.push_back(p.col(*p.z[s].value));
  if (cs.gates == H2MI_GATES_STANDARD_PLONK) {
    // StandardPlonk's schedule (measured in rounds 3-4 at 2^5 .. 2^20 rows): the products' forms queue BEHIND their commitments on
    // the library stream — on the side stream they ran beside the commitments' partition and bucket-reduction chain, whose latency
    // is what the phase waits for, and slowed it (2^16 rows: +0.2 ms)
    if (n_sets) p.commit_phase(true, zcols, 0, perm_sparse, false);
    check(h2mi_msm_flush(), "flush");
    for (uint32_t s = 0; s < n_sets; s++) p.forms(p.z[s], nullptr);
    check(h2mi_stream_wait(nullptr, p.side), "stream_wait");  // evaluate_h and the openings read the advice forms
    p.read_points(n_sets + 1, points_out);
  } else {
    // the halo2-lib shapes' schedule (measured at DEGREE 20 / 22): forms on the side stream, ordered behind the columns and AHEAD
    // of their commitments' partition kernels, which a dense accumulation in flight starves for milliseconds at DEGREE 22
    check(h2mi_stream_wait(p.side, nullptr), "stream_wait");
    for (uint32_t s = 0; s < n_sets; s++) p.forms(p.z[s], p.side);
    if (n_sets) p.commit_phase(true, zcols, 0, perm_sparse, false);
    size_t slot = n_sets;
    if (L) {
      const std::vector<Fr> lzb = p.blinding(5, (size_t)bf * L);
      for (uint32_t l = 0; l < L; l++) {
        DeviceVec& lz = *p.lk[l].z.value;
        check(h2mi_plonk_lookup_product_dev(p.lookup_input(l).p, pk.fixed_values[cs.lookups[l].table_fixed]->p, p.lk[l].a.value->p, p.lk[l].s.value->p, cs.k,
                                            u, beta.l, gamma.l, lz.p, nullptr),
              "lookup_product");
        PatchList pl;
        for (uint32_t r = 0; r < bf; r++) pl.add(lz, u + 1 + r, lzb[(size_t)bf * l + r]);
        pl.flush();
        check(h2mi_stream_wait(p.side, nullptr), "stream_wait");
        p.forms(p.lk[l].z, p.side);
        p.commit(true, p.col(lz), slot++);
      }
    }
    slot++;  // the random polynomial's commitment, queued before the grand products
    check(h2mi_msm_flush(), "flush");
    p.read_points(slot, points_out);
    // joined AFTER the read-back: the copy runs on the library stream, and a join in front of it made the transcript wait for every
    // transform of the side stream instead of the bucket reductions only
    check(h2mi_stream_wait(nullptr, p.side), "stream_wait");
  }
  p.phase = PRODUCTS;
}

// ---- phase 4: the quotient (plonk/evaluation.rs evaluate_h + vanishing division), its pieces committed ---------------------------
void phase_quotient(h2mi_prover_s& p, const Fr& y, uint64_t* points_out) {
  const h2mi_pk_s& pk = *p.pk;
  const h2mi_constraint_system& cs = pk.cs;
  const poly::EvaluationDomain& d = pk.domain;
  const size_t n = pk.n;
  const uint32_t bf = cs.blinding_factors, L = cs.n_lookups, m = cs.n_perm, n_sets = pk.n_sets;
  require_phase(p, PRODUCTS);
  p.phase = IDLE;
  DeviceVec& h = *p.h;
  const Fr& zeta = d.get_g_coset();
  const std::vector<Fr>& t_inv = d.t_inv();  // (X^n - 1)^-1 on the coset: cached in the domain
  auto coset_col = [&](const h2mi_column& c) -> const void* {
    return c.kind == H2MI_COL_ADVICE ? p.advice[c.index].coset->p : c.kind == H2MI_COL_INSTANCE ? p.instance_coset->p : pk.fixed_cosets[c.index]->p;
  };
  if (cs.gates == H2MI_GATES_STANDARD_PLONK) {
    h2mi_standard_plonk_cosets sc;
    for (int i = 0; i < 3; i++) {
      sc.advice[i] = p.advice[i].coset->p;
      sc.sigma[i] = pk.sigma_cosets[i]->p;
      sc.z[i] = p.z[i].coset->p;
    }
    for (int i = 0; i < 5; i++) sc.fixed[i] = pk.fixed_cosets[i]->p;
    sc.l0 = pk.l0->p;
    sc.l_last = pk.l_last->p;
    sc.l_active = pk.l_active->p;
    check(h2mi_plonk_evaluate_h_standard_dev(&sc, d.k(), d.extended_k(), bf, p.beta.l, p.gamma.l, y.l, pk.delta.l, zeta.l, d.get_extended_omega().l,
                                             (const uint64_t*)t_inv.data(), h.p, nullptr),
          "evaluate_h");
  } else if (cs.n_gates == 1 && m >= 1 && m <= 4 && L <= 1) {
    // one gate column: the specialised kernel (its level bookkeeping is written for this shape; faster per point than the general one)
    h2mi_range_cosets rc;
    std::memset(&rc, 0, sizeof(rc));
    rc.a = p.advice[cs.gate_advice[0]].coset->p;
    rc.q = pk.fixed_cosets[cs.gate_selector[0]]->p;
    for (uint32_t j = 0; j < m; j++) {
      rc.perm_value[j] = coset_col(cs.perm_columns[j]);
      rc.perm_sigma[j] = pk.sigma_cosets[j]->p;
    }
    for (uint32_t s = 0; s < n_sets; s++) rc.perm_z[s] = p.z[s].coset->p;
    rc.l0 = pk.l0->p;
    rc.l_last = pk.l_last->p;
    rc.l_active = pk.l_active->p;
    rc.n_perm = m;
    rc.chunk_len = pk.chunk;
    rc.has_lookup = L ? 1 : 0;
    if (L) {
      const h2mi_lookup& lk = cs.lookups[0];
      if (lk.selector_fixed >= 0) {
        if (lk.input.index != cs.gate_advice[0]) throw Error(H2MI_EINVAL, "the selector form looks up the gate column");
        rc.lookup_selector = pk.fixed_cosets[lk.selector_fixed]->p;
      } else {
        rc.lookup_advice = p.advice[lk.input.index].coset->p;
      }
      rc.table = pk.fixed_cosets[lk.table_fixed]->p;
      rc.lookup_permuted_input = p.lk[0].a.coset->p;
      rc.lookup_permuted_table = p.lk[0].s.coset->p;
      rc.lookup_z = p.lk[0].z.coset->p;
    }
    check(h2mi_plonk_evaluate_h_range_dev(&rc, d.k(), d.extended_k(), bf, p.beta.l, p.gamma.l, y.l, pk.delta.l, zeta.l, d.get_extended_omega().l,
                                          (const uint64_t*)t_inv.data(), h.p, nullptr),
          "evaluate_h");
  } else {
    // several gate columns: the general quotient kernel (one gate per column, one lookup per lookup-advice column)
    h2mi_flex_cosets fc;
    std::memset(&fc, 0, sizeof(fc));
    fc.n_gates = cs.n_gates;
    for (uint32_t g = 0; g < cs.n_gates; g++) {
      fc.gate_a[g] = p.advice[cs.gate_advice[g]].coset->p;
      fc.gate_q[g] = pk.fixed_cosets[cs.gate_selector[g]]->p;
    }
    fc.n_perm = m;
    fc.chunk_len = pk.chunk;
    for (uint32_t j = 0; j < m; j++) {
      fc.perm_value[j] = coset_col(cs.perm_columns[j]);
      fc.perm_sigma[j] = pk.sigma_cosets[j]->p;
    }
    for (uint32_t s = 0; s < n_sets; s++) fc.perm_z[s] = p.z[s].coset->p;
    fc.n_lookups = L;
    for (uint32_t l = 0; l < L; l++) {
      const h2mi_lookup& lk = cs.lookups[l];
      fc.lookup_input[l] = p.advice[lk.input.index].coset->p;
      fc.lookup_input_b[l] = lk.selector_fixed >= 0 ? pk.fixed_cosets[lk.selector_fixed]->p : nullptr;
      fc.lookup_table[l] = pk.fixed_cosets[lk.table_fixed]->p;
      fc.lookup_permuted_input[l] = p.lk[l].a.coset->p;
      fc.lookup_permuted_table[l] = p.lk[l].s.coset->p;
      fc.lookup_z[l] = p.lk[l].z.coset->p;
    }
    fc.l0 = pk.l0->p;
    fc.l_last = pk.l_last->p;
    fc.l_active = pk.l_active->p;
    check(h2mi_plonk_evaluate_h_flex_dev(&fc, d.k(), d.extended_k(), bf, p.beta.l, p.gamma.l, y.l, pk.delta.l, zeta.l, d.get_extended_omega().l,
                                         (const uint64_t*)t_inv.data(), h.p, nullptr),
          "evaluate_h");
  }
  check(h2mi_ntt_bn254_fr_dev(h.p, d.extended_k(), d.get_extended_omega_inv().l, nullptr, nullptr, nullptr), "extended_to_coeff");
  check(h2mi_fr_scale_powers_dev(h.p, pk.ext, d.get_g_coset_inv().l, d.get_extended_ifft_divisor().l, nullptr), "distribute_powers_zeta");
  const uint32_t pieces = cs.degree - 1;
  std::vector<const void*> cols;
  for (uint32_t i = 0; i < pieces; i++) cols.push_back(p.col(h, (size_t)i * n));
  p.commit_phase(false, cols, 0, /*sparse=*/false, /*inorder=*/true);
  p.read_points(pieces, points_out);
  p.phase = QUOTIENT;
}

size_t num_evaluations(const h2mi_pk_s& pk) {
  const h2mi_constraint_system& cs = pk.cs;
  return cs.n_advice_queries + cs.n_fixed_queries + 1 + cs.n_perm + (pk.n_sets ? 3 * (size_t)pk.n_sets - 1 : 0) + 5 * (size_t)cs.n_lookups;
}

// ---- phase 5: every evaluation create_proof writes -------------------------------------------------------------------------------
void phase_evaluations(h2mi_prover_s& p, const Fr& x, uint64_t* evals_out) {
  const h2mi_pk_s& pk = *p.pk;
  const h2mi_constraint_system& cs = pk.cs;
  const poly::EvaluationDomain& d = pk.domain;
  const size_t n = pk.n;
  const uint32_t bf = cs.blinding_factors, L = cs.n_lookups, n_sets = pk.n_sets, pieces = cs.degree - 1;
  require_phase(p, QUOTIENT);
  p.phase = IDLE;
  if (!evals_out) throw Error(H2MI_EINVAL, "evals_out");
  const Fr omega = d.get_omega(), omega_inv = d.get_omega_inv();
  const Fr xn = fr::pow_u64(x, n);
  auto rot = [&](int64_t r) { return r == 0 ? x : fr::mul(x, pow_signed(omega, omega_inv, r)); };
  const Fr x_next = rot(1), x_last = rot(-(int64_t)(bf + 1)), x_inv = rot(-1);
  DeviceVec &h = *p.h, &h_poly = *p.h_poly, &random_poly = *p.random_poly;
  {  // h(X) = sum_i xn^i h_i(X): the polynomial vanishing.open() queries
    std::vector<const void*> ptrs;
    std::vector<Fr> sc;
    Fr pw = fr::ONE;
    for (uint32_t i = 0; i < pieces; i++) {
      ptrs.push_back(at(h, (size_t)i * n));
      sc.push_back(pw);
      pw = fr::mul(pw, xn);
    }
    check(h2mi_fr_lincomb_dev(ptrs.data(), (const uint64_t*)sc.data(), pieces, n, h_poly.p, nullptr), "h_poly");
  }
  struct Q {
    const DeviceVec* poly;
    Fr point;
  };
  std::vector<Q> written;
  for (uint32_t i = 0; i < cs.n_advice_queries; i++) written.push_back({p.advice[cs.advice_queries[i].column].poly.get(), rot(cs.advice_queries[i].rotation)});
  for (uint32_t i = 0; i < cs.n_fixed_queries; i++) written.push_back({pk.fixed_polys[cs.fixed_queries[i].column].get(), rot(cs.fixed_queries[i].rotation)});
  written.push_back({&random_poly, x});
  for (auto& sp : pk.sigma_polys) written.push_back({sp.get(), x});
  for (uint32_t i = 0; i < n_sets; i++) {
    written.push_back({p.z[i].poly.get(), x});
    written.push_back({p.z[i].poly.get(), x_next});
    if (i + 1 < n_sets) written.push_back({p.z[i].poly.get(), x_last});
  }
  for (uint32_t l = 0; l < L; l++) {
    written.push_back({p.lk[l].z.poly.get(), x});
    written.push_back({p.lk[l].z.poly.get(), x_next});
    written.push_back({p.lk[l].a.poly.get(), x});
    written.push_back({p.lk[l].a.poly.get(), x_inv});
    written.push_back({p.lk[l].s.poly.get(), x});
  }
  std::vector<Q> todo = written;
  todo.push_back({&h_poly, x});  // opened but not written (the verifier recomputes it)
  // every evaluation in ONE call: groups of up to 24 distinct polynomials per distinct point (h2mi_fr_eval_polys_multi_dev)
  struct Slot {
    const DeviceVec* poly;
    Fr point;
  };
  std::vector<Slot> slots;
  std::vector<const void*> polys;
  std::vector<size_t> counts;
  std::vector<Fr> group_pts, distinct;
  for (const Q& q : todo)
    if (!contains(distinct, q.point)) distinct.push_back(q.point);
  for (const Fr& pt : distinct) {
    const size_t first = slots.size();
    for (const Q& q : todo) {
      if (!(q.point == pt)) continue;
      bool seen = false;
      for (size_t i = first; i < slots.size(); i++) seen = seen || slots[i].poly == q.poly;
      if (seen) continue;
      slots.push_back({q.poly, pt});
      polys.push_back(q.poly->p);
    }
    for (size_t c0 = first; c0 < slots.size(); c0 += 24) {
      counts.push_back(std::min<size_t>(24, slots.size() - c0));
      group_pts.push_back(pt);
    }
  }
  if (!p.evals || p.evals->n < slots.size()) p.evals = vec(slots.size() + 8);
  check(h2mi_fr_eval_polys_multi_dev(polys.data(), counts.data(), (const uint64_t*)group_pts.data(), counts.size(), n, p.evals->p, nullptr), "eval");
  std::vector<Fr> ev(slots.size());
  check(h2mi_memcpy_d2h(ev.data(), p.evals->p, slots.size() * 32), "d2h");
  auto value_of = [&](const DeviceVec* poly, const Fr& pt) {
    for (size_t i = 0; i < slots.size(); i++)
      if (slots[i].poly == poly && slots[i].point == pt) return ev[i];
    throw Error(H2MI_EINVAL, "query without an evaluation");
  };
  for (size_t i = 0; i < written.size(); i++) {
    const Fr v = value_of(written[i].poly, written[i].point);
    std::memcpy(evals_out + 4 * i, v.l, 32);
  }
  p.n_written = written.size();
  // the queries in create_proof's order (advice; permutation.open: every set at x and omega x, then all but the last at
  // omega^last x in reverse; lookups; fixed; sigma; vanishing.open: h, the random polynomial)
  p.queries.clear();
  auto q = [&](const DeviceVec* poly, const Fr& pt) { p.queries.push_back({poly, pt, value_of(poly, pt)}); };
  for (uint32_t i = 0; i < cs.n_advice_queries; i++) q(p.advice[cs.advice_queries[i].column].poly.get(), rot(cs.advice_queries[i].rotation));
  for (uint32_t i = 0; i < n_sets; i++) {
    q(p.z[i].poly.get(), x);
    q(p.z[i].poly.get(), x_next);
  }
  for (uint32_t i = n_sets > 0 ? n_sets - 1 : 0; i-- > 0;) q(p.z[i].poly.get(), x_last);
  for (uint32_t l = 0; l < L; l++) {
    q(p.lk[l].z.poly.get(), x);
    q(p.lk[l].a.poly.get(), x);
    q(p.lk[l].s.poly.get(), x);
    q(p.lk[l].a.poly.get(), x_inv);
    q(p.lk[l].z.poly.get(), x_next);
  }
  for (uint32_t i = 0; i < cs.n_fixed_queries; i++) q(pk.fixed_polys[cs.fixed_queries[i].column].get(), rot(cs.fixed_queries[i].rotation));
  for (auto& sp : pk.sigma_polys) q(sp.get(), x);
  q(&h_poly, x);
  q(&random_poly, x);
  p.phase = EVALUATIONS;
}

// a lone commitment whose point the caller reads next: in order on one stream, nothing deferred
void shplonk_commit(h2mi_prover_s& p, const DeviceVec& poly, uint64_t* point_out) {
  std::vector<const void*> cols = {p.col(poly)};
  p.commit_phase(false, cols, 0, false, /*inorder=*/true);
  p.read_points(1, point_out);
}

// ---- phase 6: ProverSHPLONK up to the commitment of h(X) = sum_i v^i Q_i(X) ------------------------------------------------------
void phase_shplonk_quotient(h2mi_prover_s& p, const Fr& y, const Fr& v, uint64_t* point_out) {
  const size_t n = p.pk->n;
  require_phase(p, EVALUATIONS);
  p.phase = IDLE;
  // construct_intermediate_sets
  std::vector<std::pair<const DeviceVec*, std::vector<std::pair<Fr, Fr>>>> by_poly;  // first-appearance order
  std::vector<Fr>& super_points = p.super_points;
  super_points.clear();
  for (const ProverQuery& q : p.queries) {
    if (!contains(super_points, q.point)) super_points.push_back(q.point);
    auto it = std::find_if(by_poly.begin(), by_poly.end(), [&](const auto& e) { return e.first == q.poly; });
    if (it == by_poly.end()) {
      by_poly.push_back({q.poly, {{q.point, q.eval}}});
    } else if (std::none_of(it->second.begin(), it->second.end(), [&](const auto& pe) { return pe.first == q.point; })) {
      it->second.push_back({q.point, q.eval});
    }
  }
  std::sort(super_points.begin(), super_points.end(), canonical_less);
  std::vector<RotationSet>& sets = p.sets;
  sets.clear();
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
    for (const Fr& pt : rs->points) evals.push_back(std::find_if(e.second.begin(), e.second.end(), [&](const auto& pe) { return pe.first == pt; })->second);
    rs->members.push_back({e.first, evals});
  }
  p.v = v;
  {  // the divisions below need the power tables of every opening point and of its inverse: built now, in one launch
    std::vector<Fr> bases = super_points;
    const std::vector<Fr> inv = fr::batch_invert(super_points);
    bases.insert(bases.end(), inv.begin(), inv.end());
    if (!bases.empty() && bases.size() <= 32) check(h2mi_fr_powtab_prefetch_dev((const uint64_t*)bases.data(), bases.size(), n, nullptr), "powtab_prefetch");
  }
  while (p.q.size() < sets.size()) p.q.push_back(vec(n));
  while (p.s.size() < sets.size()) p.s.push_back(vec(n));
  // quotient contributions Q_i = (sum_j y^j (P_ij - R_ij)) / Z_i.  The sets are independent chains of small launches (one linear
  // combination, one division round per set): set i runs on lane i mod 3, so the longest chain, not their sum, is waited for
  p.remainders.clear();
  for (size_t i = 0; i < sets.size(); i++) check(h2mi_memset_zero(p.q[i]->p, n * 32), "zero");
  for (h2mi_stream_t st : p.lane) check(h2mi_stream_wait(st, nullptr), "stream_wait");
  for (size_t i = 0; i < sets.size(); i++) {
    const RotationSet& rs = sets[i];
    const size_t lane = i % 3;
    h2mi_stream_t stream = lane ? p.lane[lane - 1] : nullptr;
    std::vector<Fr> ypow(rs.members.size(), fr::ONE);
    for (size_t j = 1; j < ypow.size(); j++) ypow[j] = fr::mul(ypow[j - 1], y);
    std::vector<const DeviceVec*> polys;
    for (auto& mb : rs.members) polys.push_back(mb.first);
    lincomb(polys, ypow, n, *p.s[i], stream);
    std::vector<Fr> rsum(rs.points.size(), fr_zero());
    const std::vector<std::vector<Fr>> basis = lagrange_basis(rs.points);
    for (size_t j = 0; j < rs.members.size(); j++) {
      const std::vector<Fr> r = interpolate(basis, rs.members[j].second);
      for (size_t t = 0; t < r.size(); t++) rsum[t] = fr::sub(rsum[t], fr::mul(ypow[j], r[t]));
    }
    add_head(*p.s[i], rsum, stream);
    std::vector<Fr> rem;
    for (const Fr& c : rsum) rem.push_back(fr::neg(c));
    p.remainders.push_back(rem);  // R_i(X) = sum_j y^j R_ij(X), low to high
    kate_chain(*p.s[i], n, rs.points, *p.tmp[lane], *p.q[i], stream, p.nx[lane].get());
  }
  for (h2mi_stream_t st : p.lane) check(h2mi_stream_wait(nullptr, st), "stream_wait");
  {
    std::vector<const DeviceVec*> polys;
    std::vector<Fr> vpow(sets.size(), fr::ONE);
    for (size_t i = 1; i < vpow.size(); i++) vpow[i] = fr::mul(vpow[i - 1], v);
    for (size_t i = 0; i < sets.size(); i++) polys.push_back(p.q[i].get());
    lincomb(polys, vpow, n, *p.h_x);
  }
  shplonk_commit(p, *p.h_x, point_out);
  p.phase = SHPLONK_Q;
}

// ---- phase 7: the linearisation L(X) at u, divided by (X - u) -------------------------------------------------------------------
void phase_shplonk_open(h2mi_prover_s& p, const Fr& u, uint64_t* point_out) {
  const size_t n = p.pk->n;
  require_phase(p, SHPLONK_Q);
  p.phase = IDLE;
  const std::vector<RotationSet>& sets = p.sets;
  const std::vector<Fr>& super_points = p.super_points;
  const Fr zt_eval = vanishing_at(super_points, u);
  std::vector<Fr> z_diffs;
  for (const RotationSet& rs : sets) {
    std::vector<Fr> diffs;
    for (const Fr& pt : super_points)
      if (!contains(rs.points, pt)) diffs.push_back(pt);
    z_diffs.push_back(vanishing_at(diffs, u));
  }
  const Fr norm = fr::invert(z_diffs[0]);  // "normalize coefficients by the coefficient of the first polynomial"
  // sum_j y^j (P_ij(X) - R_ij(u)) = S_i(X) + R_i(X) - R_i(u) with S_i the vector the quotient step left in s[i] — one linear
  // combination over the rotation sets' sums and h(X) instead of every opened polynomial again, and a low-degree head
  std::vector<const DeviceVec*> polys;
  std::vector<Fr> scalars;
  size_t head_len = 0;
  for (const RotationSet& rs : sets) head_len = std::max(head_len, rs.points.size());
  std::vector<Fr> head(head_len, fr_zero());
  Fr vp = fr::ONE;
  for (size_t i = 0; i < sets.size(); i++) {
    const Fr w = fr::mul(fr::mul(vp, z_diffs[i]), norm);
    polys.push_back(p.s[i].get());
    scalars.push_back(w);
    for (size_t t = 0; t < p.remainders[i].size(); t++) head[t] = fr::add(head[t], fr::mul(w, p.remainders[i][t]));
    head[0] = fr::sub(head[0], fr::mul(w, horner(p.remainders[i], u)));
    vp = fr::mul(vp, p.v);
  }
  polys.push_back(p.h_x.get());
  scalars.push_back(fr::neg(fr::mul(zt_eval, norm)));
  lincomb(polys, scalars, n, *p.l_x);
  add_head(*p.l_x, head);
  check(h2mi_memset_zero(p.h2_x->p, n * 32), "zero");
  kate_chain(*p.l_x, n, {u}, *p.tmp[0], *p.h2_x, nullptr, nullptr);
  shplonk_commit(p, *p.h2_x, point_out);
  p.phase = IDLE;  // the proof is complete; the next one starts with h2mi_prover_advice
}

Fr load_fr(const uint64_t* l) {
  Fr a;
  std::memcpy(a.l, l, 32);
  return a;
}

// no exception crosses the C boundary
template <class F>
int guarded(F f) {
  try {
    f();
    return H2MI_OK;
  } catch (const Error& e) {
    return e.code;
  } catch (const std::bad_alloc&) {
    return H2MI_ENOMEM;
  } catch (...) {
    return H2MI_EINVAL;
  }
}

int pick(const std::vector<Dev>& v, uint32_t index, void** d_ptr, size_t* count) {
  if (index >= v.size() || !v[index]) return H2MI_ERANGE;
  *d_ptr = v[index]->p;
  if (count) *count = v[index]->n;
  return H2MI_OK;
}
int pick(const Dev& d, uint32_t index, void** d_ptr, size_t* count) {
  if (index != 0 || !d) return H2MI_ERANGE;
  *d_ptr = d->p;
  if (count) *count = d->n;
  return H2MI_OK;
}

}  // namespace

extern "C" {

int h2mi_prover_keygen(const h2mi_constraint_system* cs, uint64_t g_lagrange_handle, const h2mi_column_cells* fixed, const uint32_t* copies, size_t n_copies,
                       unsigned flags, h2mi_pk_t* pk_out) {
  if (!cs || !pk_out || (cs->n_fixed && !fixed) || (n_copies && !copies) || (flags & ~(unsigned)H2MI_KEYGEN_VK_ONLY)) return H2MI_EINVAL;
  *pk_out = nullptr;
  if (h2mi_device_count() == 0) return H2MI_ENODEV;  // no CPU fallback: the prover exists on a GPU or not at all
  return guarded([&] {
    std::unique_ptr<h2mi_pk_s> pk = keygen(*cs, g_lagrange_handle, fixed, copies, n_copies, flags);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_live_pks.insert(pk.get());
    *pk_out = pk.release();
  });
}

int h2mi_prover_pk_release(h2mi_pk_t pk) {
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (!pk || !g_live_pks.count(pk)) return H2MI_EHANDLE;
    if (pk->users) return H2MI_EINVAL;  // destroy its provers first
    g_live_pks.erase(pk);
  }
  h2mi_sync();  // nothing queued may still read the key's vectors
  delete pk;
  return H2MI_OK;
}

int h2mi_prover_vk_commitments(h2mi_pk_t pk, uint64_t* fixed_out, uint64_t* permutation_out) {
  if (!alive(g_live_pks, pk)) return H2MI_EHANDLE;
  if (fixed_out && !pk->fixed_commitments.empty()) std::memcpy(fixed_out, pk->fixed_commitments.data(), pk->fixed_commitments.size() * 64);
  if (permutation_out && !pk->permutation_commitments.empty())
    std::memcpy(permutation_out, pk->permutation_commitments.data(), pk->permutation_commitments.size() * 64);
  return H2MI_OK;
}

int h2mi_prover_create(h2mi_pk_t pk, uint64_t g_handle, uint64_t g_lagrange_handle, size_t base_lo, size_t base_count, h2mi_prover_t* prover_out) {
  if (!prover_out) return H2MI_EINVAL;
  *prover_out = nullptr;
  if (h2mi_device_count() == 0) return H2MI_ENODEV;
  if (!alive(g_live_pks, pk)) return H2MI_EHANDLE;
  return guarded([&] {
    std::unique_ptr<h2mi_prover_s> p = create_prover(pk, g_handle, g_lagrange_handle, base_lo, base_count);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_live_provers.insert(p.get());
    pk->users++;
    *prover_out = p.release();
  });
}

int h2mi_prover_destroy(h2mi_prover_t prover) {
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (!prover || !g_live_provers.erase(prover)) return H2MI_EHANDLE;
    prover->pk->users--;
  }
  h2mi_sync();
  delete prover;
  return H2MI_OK;
}

int h2mi_prover_set_combiner(h2mi_prover_t prover, void* d_partial, void* d_combined, h2mi_combine_fn combine, void* ctx) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!d_partial && !d_combined && !combine) {  // back to whole-SRS commitments
    prover->d_partial = prover->d_combined = nullptr;
    prover->combine = nullptr;
    return H2MI_OK;
  }
  if (!d_partial || !d_combined || !combine) return H2MI_EINVAL;
  prover->d_partial = d_partial;
  prover->d_combined = d_combined;
  prover->combine = combine;
  prover->combine_ctx = ctx;
  return H2MI_OK;
}

int h2mi_prover_set_rng_key(h2mi_prover_t prover, const uint8_t key[32]) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  prover->keyed = key != nullptr;
  if (key) std::memcpy(prover->key, key, 32);
  else std::memset(prover->key, 0, 32);
  prover->phase = IDLE;  // a proof in flight would mix two sources
  return H2MI_OK;
}

int h2mi_prover_get_counts(h2mi_prover_t prover, h2mi_prover_counts* out) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!out) return H2MI_EINVAL;
  const h2mi_pk_s& pk = *prover->pk;
  out->advice = pk.cs.n_advice;
  out->lookups = 2 * pk.cs.n_lookups;
  out->products = pk.n_sets + pk.cs.n_lookups + 1;
  out->quotient = pk.cs.degree - 1;
  out->evaluations = (uint32_t)num_evaluations(pk);
  return H2MI_OK;
}

int h2mi_prover_advice(h2mi_prover_t prover, const h2mi_column_cells* advice, const uint64_t* instance, size_t n_instance_values, uint64_t seed,
                       uint64_t* points_out) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!advice) return H2MI_EINVAL;
  return guarded([&] { phase_advice(*prover, advice, instance, n_instance_values, seed, points_out); });
}
int h2mi_prover_lookups(h2mi_prover_t prover, const uint64_t theta[4], uint64_t* points_out) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  (void)theta;  // single-expression lookups have nothing to compress
  return guarded([&] { phase_lookups(*prover, points_out); });
}
int h2mi_prover_products(h2mi_prover_t prover, const uint64_t beta[4], const uint64_t gamma[4], uint64_t* points_out) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!beta || !gamma) return H2MI_EINVAL;
  return guarded([&] { phase_products(*prover, load_fr(beta), load_fr(gamma), points_out); });
}
int h2mi_prover_quotient(h2mi_prover_t prover, const uint64_t y[4], uint64_t* points_out) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!y) return H2MI_EINVAL;
  return guarded([&] { phase_quotient(*prover, load_fr(y), points_out); });
}
int h2mi_prover_num_evaluations(h2mi_prover_t prover, size_t* count_out) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!count_out) return H2MI_EINVAL;
  *count_out = num_evaluations(*prover->pk);
  return H2MI_OK;
}
int h2mi_prover_evaluations(h2mi_prover_t prover, const uint64_t x[4], uint64_t* evals_out) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!x) return H2MI_EINVAL;
  return guarded([&] { phase_evaluations(*prover, load_fr(x), evals_out); });
}
int h2mi_prover_shplonk_quotient(h2mi_prover_t prover, const uint64_t y[4], const uint64_t v[4], uint64_t point_out[8]) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!y || !v) return H2MI_EINVAL;
  return guarded([&] { phase_shplonk_quotient(*prover, load_fr(y), load_fr(v), point_out); });
}
int h2mi_prover_shplonk_open(h2mi_prover_t prover, const uint64_t u[4], uint64_t point_out[8]) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!u) return H2MI_EINVAL;
  return guarded([&] { phase_shplonk_open(*prover, load_fr(u), point_out); });
}

int h2mi_prover_buffer(h2mi_prover_t prover, uint32_t kind, uint32_t index, void** d_ptr_out, size_t* count_out) {
  if (!alive(g_live_provers, prover)) return H2MI_EHANDLE;
  if (!d_ptr_out) return H2MI_EINVAL;
  h2mi_prover_s& p = *prover;
  auto form = [&](std::vector<Forms>& v, int which) -> int {
    if (index >= v.size()) return H2MI_ERANGE;
    const Dev& d = which == 0 ? v[index].value : which == 1 ? v[index].poly : v[index].coset;
    return pick(d, 0, d_ptr_out, count_out);
  };
  switch (kind) {
    case H2MI_BUF_ADVICE: return form(p.advice, 0);
    case H2MI_BUF_ADVICE_POLY: return form(p.advice, 1);
    case H2MI_BUF_ADVICE_COSET: return form(p.advice, 2);
    case H2MI_BUF_INSTANCE: return pick(p.instance, index, d_ptr_out, count_out);
    case H2MI_BUF_PERM_Z: return form(p.z, 0);
    case H2MI_BUF_PERM_Z_POLY: return form(p.z, 1);
    case H2MI_BUF_PERM_Z_COSET: return form(p.z, 2);
    case H2MI_BUF_LOOKUP_PERMUTED_INPUT: return index < p.pk->cs.n_lookups ? pick(p.lk[index].a.value, 0, d_ptr_out, count_out) : H2MI_ERANGE;
    case H2MI_BUF_LOOKUP_PERMUTED_TABLE: return index < p.pk->cs.n_lookups ? pick(p.lk[index].s.value, 0, d_ptr_out, count_out) : H2MI_ERANGE;
    case H2MI_BUF_LOOKUP_Z: return index < p.pk->cs.n_lookups ? pick(p.lk[index].z.value, 0, d_ptr_out, count_out) : H2MI_ERANGE;
    case H2MI_BUF_RANDOM_POLY: return pick(p.random_poly, index, d_ptr_out, count_out);
    case H2MI_BUF_H: return pick(p.h, index, d_ptr_out, count_out);
    case H2MI_BUF_H_POLY: return pick(p.h_poly, index, d_ptr_out, count_out);
    case H2MI_BUF_SHPLONK_H: return pick(p.h_x, index, d_ptr_out, count_out);
    case H2MI_BUF_SHPLONK_H2: return pick(p.h2_x, index, d_ptr_out, count_out);
    default: return h2mi_prover_pk_buffer(p.pk, kind, index, d_ptr_out, count_out);
  }
}

int h2mi_prover_pk_buffer(h2mi_pk_t pk, uint32_t kind, uint32_t index, void** d_ptr_out, size_t* count_out) {
  if (!alive(g_live_pks, pk)) return H2MI_EHANDLE;
  if (!d_ptr_out) return H2MI_EINVAL;
  switch (kind) {
    case H2MI_PKBUF_FIXED: return pick(pk->fixed_values, index, d_ptr_out, count_out);
    case H2MI_PKBUF_FIXED_POLY: return pick(pk->fixed_polys, index, d_ptr_out, count_out);
    case H2MI_PKBUF_FIXED_COSET: return pick(pk->fixed_cosets, index, d_ptr_out, count_out);
    case H2MI_PKBUF_SIGMA: return pick(pk->sigma_values, index, d_ptr_out, count_out);
    case H2MI_PKBUF_SIGMA_POLY: return pick(pk->sigma_polys, index, d_ptr_out, count_out);
    case H2MI_PKBUF_SIGMA_COSET: return pick(pk->sigma_cosets, index, d_ptr_out, count_out);
    case H2MI_PKBUF_L0_COSET: return pick(pk->l0, index, d_ptr_out, count_out);
    case H2MI_PKBUF_L_LAST_COSET: return pick(pk->l_last, index, d_ptr_out, count_out);
    case H2MI_PKBUF_L_ACTIVE_COSET: return pick(pk->l_active, index, d_ptr_out, count_out);
    default: return H2MI_EINVAL;
  }
}

}  // extern "C"
