// h2mi_flex.hpp — C++17 host layer for the halo2-lib builders the reference proves through `scaffold::prove`
// (src/scaffold.rs:246-366: GateWithInstanceCircuitBuilder / RangeWithInstanceCircuitBuilder, :379-485): the constraint
// systems halo2-base configures for ONE gate advice column, its Context (cell layout of load_witness / mul / add /
// mul_add / range_check), keygen and create_proof with every vector resident in HBM, over the C ABI (h2mi.h).
//
// The same restatement as the Python host (halo2-scaffold_amd/flex.py — see its header for what is recalled from the
// un-vendored halo2-base and what the reference itself shows) and the oracle (oracle/flex.py): the three produce
// identical proof bytes (tests/test_gpu_flex.py).  Closures: examples/halo2_lib.rs:14-60 and examples/range.rs:10-34.
//
//   Gate builder:  fixed 0 constants, 1 q_enable;                         degree 3, permutation sets of one, 2 h pieces
//   Range builder: fixed 0 table, 1 constants, 2 q_lookup, 3 q_enable;    the lookup of q_lookup * a makes the degree 5:
//                  one permutation set of three, 4 h pieces, extended domain 4n
// rng stand-in as in h2mi_plonk.hpp, plus streams seed + 4 (blinding rows of the permuted lookup columns) and seed + 5
// (of the lookup product).
#pragma once
#include <array>

#include "h2mi_plonk.hpp"

namespace h2mi {
namespace flex {

using arithmetic::DeviceVec;
using plonk::Dev;
using plonk::fr_zero;

enum Kind : uint32_t { ADVICE = 0, FIXED = 1, INSTANCE = 2 };
struct Col {
  Kind kind;
  uint32_t index;
};
struct CellRef {
  Kind kind;
  uint32_t col, row;
};
typedef std::pair<uint32_t, int32_t> Query;  // (column, rotation)

struct FlexGateCS {
  bool lookup;
  uint32_t n_fixed;
  int col_table = -1, col_const, col_qlookup = -1, col_q;
  std::vector<Col> perm_columns;
  std::vector<Query> advice_queries, fixed_queries;
  uint32_t degree, blinding_factors = 6, chunk;
  explicit FlexGateCS(bool with_lookup) : lookup(with_lookup) {
    if (lookup) {
      col_table = 0; col_const = 1; col_qlookup = 2; col_q = 3;
      n_fixed = 4;
      fixed_queries = {{1, 0}, {0, 0}, {2, 0}, {3, 0}};
    } else {
      col_const = 0; col_q = 1;
      n_fixed = 2;
      fixed_queries = {{0, 0}, {1, 0}};
    }
    perm_columns = {{FIXED, (uint32_t)col_const}, {ADVICE, 0}, {INSTANCE, 0}};
    advice_queries = {{0, 0}, {0, 1}, {0, 2}, {0, 3}};
    degree = lookup ? 5 : 3;
    chunk = degree - 2;
  }
};

struct Assignment {
  const FlexGateCS* cs;
  std::vector<Fr> advice;                     // the gate advice column, rows 0 ..
  std::vector<std::map<uint32_t, Fr>> fixed;  // sparse cells per fixed column (the table column: `table_values`)
  std::vector<Fr> instance;                   // public inputs
  std::vector<std::pair<CellRef, CellRef>> copies;
  std::vector<uint64_t> table_values;
  explicit Assignment(const FlexGateCS& c) : cs(&c), fixed(c.n_fixed) {}
};

// halo2-base `Context` on one advice column
class Context {
 public:
  enum What { WITNESS, CONSTANT, EXISTING };
  struct Item {
    What what;
    Fr value;      // WITNESS / CONSTANT
    uint32_t cell; // EXISTING
  };
  static Item witness(const Fr& v) { return {WITNESS, v, 0}; }
  static Item constant(const Fr& v) { return {CONSTANT, v, 0}; }
  static Item constant(uint64_t v) { return {CONSTANT, fr::from_u64(v), 0}; }
  static Item existing(uint32_t c) { return {EXISTING, fr_zero(), c}; }

  explicit Context(Assignment& a) : asg_(a) {}
  const Fr& value(uint32_t cell) const { return asg_.advice[cell]; }
  uint32_t load_witness(const Fr& v) {
    asg_.advice.push_back(v);
    return (uint32_t)asg_.advice.size() - 1;
  }
  uint32_t assign_region_last(const std::vector<Item>& items, const std::vector<uint32_t>& gate_offsets) {
    const uint32_t base = (uint32_t)asg_.advice.size();
    for (const Item& it : items) {
      const uint32_t row = (uint32_t)asg_.advice.size();
      if (it.what == EXISTING) {
        asg_.advice.push_back(asg_.advice[it.cell]);
        asg_.copies.push_back({{ADVICE, 0, row}, {ADVICE, 0, it.cell}});
      } else {
        asg_.advice.push_back(it.value);
        if (it.what == CONSTANT) const_cells_.push_back({row, it.value});
      }
    }
    for (uint32_t off : gate_offsets) asg_.fixed[asg_.cs->col_q][base + off] = fr::ONE;
    return (uint32_t)asg_.advice.size() - 1;
  }
  // GateInstructions
  uint32_t mul(uint32_t a, uint32_t b) { return assign_region_last({constant(0), existing(a), existing(b), witness(fr::mul(value(a), value(b)))}, {0}); }
  uint32_t add(uint32_t a, uint32_t b) { return assign_region_last({existing(a), existing(b), constant(1), witness(fr::add(value(a), value(b)))}, {0}); }
  uint32_t add_constant(uint32_t a, const Fr& c) { return assign_region_last({existing(a), constant(c), constant(1), witness(fr::add(value(a), c))}, {0}); }
  uint32_t mul_add_constant(uint32_t a, uint32_t b, const Fr& c) {
    return assign_region_last({constant(c), existing(a), existing(b), witness(fr::add(fr::mul(value(a), value(b)), c))}, {0});
  }
  // RangeInstructions::range_check(a, range_bits) for range_bits <= 64
  void range_check(uint32_t a, uint32_t range_bits, uint32_t lookup_bits) {
    const Fr canon = plonk::to_canonical(value(a));
    if (range_bits > 64 || canon.l[1] || canon.l[2] || canon.l[3] || (range_bits < 64 && (canon.l[0] >> range_bits)))
      throw Error(H2MI_EINVAL, "range_check: witness out of range");
    const uint64_t x = canon.l[0];
    const uint32_t num_limbs = (range_bits + lookup_bits - 1) / lookup_bits;
    const uint64_t mask = (1ULL << lookup_bits) - 1;
    std::vector<uint64_t> limbs(num_limbs);
    for (uint32_t i = 0; i < num_limbs; i++) limbs[i] = (x >> (lookup_bits * i)) & mask;
    std::vector<uint32_t> rows = {load_witness(fr::from_u64(limbs[0]))};  // bases[0] = 1: the first limb is the first accumulator
    uint64_t acc = limbs[0];
    uint32_t acc_row = rows[0];
    for (uint32_t i = 1; i < num_limbs; i++) {  // [acc, limb_i, 2^(b i), acc'] sharing the accumulator cell
      const uint32_t base = (uint32_t)asg_.advice.size() - 1;
      acc += limbs[i] << (lookup_bits * i);
      rows.push_back(load_witness(fr::from_u64(limbs[i])));
      const Fr pw = fr::from_u64(1ULL << (lookup_bits * i));
      const_cells_.push_back({load_witness(pw), pw});
      acc_row = load_witness(fr::from_u64(acc));
      asg_.fixed[asg_.cs->col_q][base] = fr::ONE;
    }
    asg_.copies.push_back({{ADVICE, 0, a}, {ADVICE, 0, acc_row}});  // ctx.constrain_equal(&a, &acc)
    for (uint32_t r : rows) lookup_cells_.push_back(r);
    const uint32_t rem = range_bits % lookup_bits;
    if (rem == 1) {  // a one-bit top limb: assert_bit, | 0 | x | x | x |
      assign_region_last({constant(0), existing(rows.back()), existing(rows.back()), existing(rows.back())}, {0});
    } else if (rem) {  // the top limb times 2^(lookup_bits - rem) must be in the table too
      lookup_cells_.push_back(assign_region_last({constant(0), existing(rows.back()), constant(1ULL << (lookup_bits - rem)),
                                                  witness(fr::from_u64(limbs.back() << (lookup_bits - rem)))}, {0}));
    }
  }
  void finish(const std::vector<uint32_t>& public_rows) {
    const FlexGateCS& cs = *asg_.cs;
    std::vector<Fr> consts;  // one fixed cell per distinct value, in order of first use
    for (const auto& rc : const_cells_) {
      size_t idx = std::find(consts.begin(), consts.end(), rc.second) - consts.begin();
      if (idx == consts.size()) {
        consts.push_back(rc.second);
        asg_.fixed[cs.col_const][(uint32_t)idx] = rc.second;
      }
      asg_.copies.push_back({{ADVICE, 0, rc.first}, {FIXED, (uint32_t)cs.col_const, (uint32_t)idx}});
    }
    for (uint32_t r : lookup_cells_) asg_.fixed[cs.col_qlookup][r] = fr::ONE;
    for (size_t i = 0; i < public_rows.size(); i++) {  // layouter.constrain_instance(cell, instance, i)
      asg_.instance.push_back(asg_.advice[public_rows[i]]);
      asg_.copies.push_back({{ADVICE, 0, public_rows[i]}, {INSTANCE, 0, (uint32_t)i}});
    }
  }

 private:
  Assignment& asg_;
  std::vector<std::pair<uint32_t, Fr>> const_cells_;
  std::vector<uint32_t> lookup_cells_;
};

// reference examples/halo2_lib.rs:14-60 `some_algorithm_in_zk`: x^2 + 72 three ways; make_public = [x, out]
inline Assignment halo2_lib_closure(const FlexGateCS& cs, const Fr& x) {
  Assignment asg(cs);
  Context ctx(asg);
  const Fr c = fr::from_u64(72);
  const uint32_t xc = ctx.load_witness(x);
  const uint32_t x_sq = ctx.mul(xc, xc);
  const uint32_t out = ctx.add_constant(x_sq, c);
  ctx.assign_region_last({Context::constant(c), Context::existing(xc), Context::existing(xc), Context::witness(fr::add(fr::mul(x, x), c))}, {0});
  ctx.mul_add_constant(xc, xc, c);
  ctx.finish({xc, out});
  return asg;
}
// reference examples/range.rs:10-34: make_public = [x]; range_check(x, 64); x + x.  Table: 0 .. 2^LOOKUP_BITS - 1
inline Assignment range_closure(const FlexGateCS& cs, uint64_t x, uint32_t lookup_bits) {
  Assignment asg(cs);
  Context ctx(asg);
  const uint32_t xc = ctx.load_witness(fr::from_u64(x));
  ctx.range_check(xc, 64, lookup_bits);
  ctx.add(xc, xc);
  ctx.finish({xc});
  asg.table_values.resize((size_t)1 << lookup_bits);
  for (size_t i = 0; i < asg.table_values.size(); i++) asg.table_values[i] = i;
  return asg;
}

// ---- reference examples/poseidon.rs:15-36 `hash_two`: T = 3, RATE = 2, R_F = 8, R_P = 57 ---------------------------------
// Parameters from the Grain LFSR of the Poseidon reference script (pinned in the test-suite by circomlib's published
// constants); the permutation laid out from its definition on FlexGate cells; sponge convention [RECALL snark-verifier]:
// state (2^64, 0, 0), inputs added into state[1..], a short / trailing empty chunk adds 1 at the next free position,
// squeeze returns state[1].  See halo2-scaffold_amd/poseidon.py.
namespace poseidon {
constexpr uint32_t T = 3, RATE = 2, R_F = 8, R_P = 57;
class Grain {
 public:
  Grain(uint32_t t, uint32_t r_f, uint32_t r_p) {
    const std::pair<uint64_t, int> fields[] = {{1, 2}, {0, 4}, {254, 12}, {t, 12}, {r_f, 10}, {r_p, 10}, {(1u << 30) - 1, 30}};
    for (const auto& f : fields)
      for (int i = f.second - 1; i >= 0; i--) bits_.push_back((f.first >> i) & 1);
    for (int i = 0; i < 160; i++) clock();
  }
  Fr next_field_element(bool reject) {  // 254 bits, big-endian
    for (;;) {
      uint64_t l[4] = {0, 0, 0, 0};
      for (int i = 0; i < 254; i++) {
        for (int j = 3; j > 0; j--) l[j] = (l[j] << 1) | (l[j - 1] >> 63);
        l[0] = (l[0] << 1) | next_bit();
      }
      bool lt = false;
      for (int j = 3; j >= 0; j--) {
        if (l[j] != fr::MODULUS[j]) { lt = l[j] < fr::MODULUS[j]; break; }
      }
      if (reject && !lt) continue;
      Fr raw{{l[0], l[1], l[2], l[3]}};
      return fr::mul(raw, fr::R2);  // Montgomery form (reduces values >= r as well)
    }
  }

 private:
  uint64_t clock() {
    const uint64_t b = bits_[pos_ + 62] ^ bits_[pos_ + 51] ^ bits_[pos_ + 38] ^ bits_[pos_ + 23] ^ bits_[pos_ + 13] ^ bits_[pos_];
    bits_.push_back((uint8_t)b);
    pos_++;
    return b;
  }
  uint64_t next_bit() {
    for (;;) {
      const uint64_t first = clock(), second = clock();
      if (first) return second;
    }
  }
  std::vector<uint8_t> bits_;
  size_t pos_ = 0;
};
struct Spec {
  std::vector<std::array<Fr, T>> constants;
  std::array<std::array<Fr, T>, T> mds;
};
inline const Spec& spec() {
  static const Spec s = [] {
    Spec sp;
    Grain g(T, R_F, R_P);
    for (uint32_t r = 0; r < R_F + R_P; r++) {
      std::array<Fr, T> row;
      for (uint32_t i = 0; i < T; i++) row[i] = g.next_field_element(true);
      sp.constants.push_back(row);
    }
    for (;;) {
      std::array<Fr, T> xs, ys;
      for (uint32_t i = 0; i < T; i++) xs[i] = g.next_field_element(false);
      for (uint32_t i = 0; i < T; i++) ys[i] = g.next_field_element(false);
      bool ok = true;
      std::vector<Fr> all(xs.begin(), xs.end());
      all.insert(all.end(), ys.begin(), ys.end());
      for (size_t i = 0; i < all.size(); i++)
        for (size_t j = i + 1; j < all.size(); j++) ok = ok && !(all[i] == all[j]);
      for (uint32_t i = 0; i < T; i++)
        for (uint32_t j = 0; j < T; j++) ok = ok && !(fr::add(xs[i], ys[j]) == fr_zero());
      if (!ok) continue;
      for (uint32_t i = 0; i < T; i++)
        for (uint32_t j = 0; j < T; j++) sp.mds[i][j] = fr::invert(fr::add(xs[i], ys[j]));
      return sp;
    }
  }();
  return s;
}
class Chip {
 public:
  explicit Chip(Context& ctx) : ctx_(ctx) {
    Fr two64 = fr::from_u64(1ULL << 32);
    two64 = fr::mul(two64, two64);
    state_ = {ctx.assign_region_last({Context::constant(two64)}, {}), ctx.assign_region_last({Context::constant(0)}, {}),
              ctx.assign_region_last({Context::constant(0)}, {})};
  }
  void update(const std::vector<uint32_t>& cells) { buf_.insert(buf_.end(), cells.begin(), cells.end()); }
  uint32_t squeeze() {
    std::vector<std::vector<uint32_t>> chunks;
    for (size_t i = 0; i < buf_.size(); i += RATE) chunks.push_back(std::vector<uint32_t>(buf_.begin() + i, buf_.begin() + std::min(buf_.size(), i + RATE)));
    if (buf_.size() % RATE == 0) chunks.push_back({});
    buf_.clear();
    for (const auto& chunk : chunks) {
      for (size_t i = 0; i < chunk.size(); i++) state_[1 + i] = ctx_.add(state_[1 + i], chunk[i]);
      if (chunk.size() < RATE) state_[1 + chunk.size()] = ctx_.add_constant(state_[1 + chunk.size()], fr::ONE);
      permute();
    }
    return state_[1];
  }

 private:
  uint32_t inner_product_const(const std::array<uint32_t, T>& cells, const std::array<Fr, T>& coeffs) {
    std::vector<Context::Item> items = {Context::constant(0)};
    std::vector<uint32_t> gates;
    Fr acc = fr_zero();
    for (uint32_t k = 0; k < T; k++) {
      acc = fr::add(acc, fr::mul(ctx_.value(cells[k]), coeffs[k]));
      items.push_back(Context::existing(cells[k]));
      items.push_back(Context::constant(coeffs[k]));
      items.push_back(Context::witness(acc));
      gates.push_back(3 * k);
    }
    return ctx_.assign_region_last(items, gates);
  }
  void permute() {
    const Spec& sp = spec();
    const uint32_t half = R_F / 2;
    for (uint32_t rnd = 0; rnd < R_F + R_P; rnd++) {
      std::array<uint32_t, T> s;
      for (uint32_t i = 0; i < T; i++) s[i] = ctx_.add_constant(state_[i], sp.constants[rnd][i]);
      const uint32_t lanes = (rnd < half || rnd >= half + R_P) ? T : 1;
      for (uint32_t i = 0; i < lanes; i++) {
        const uint32_t x2 = ctx_.mul(s[i], s[i]), x4 = ctx_.mul(x2, x2);
        s[i] = ctx_.mul(x4, s[i]);
      }
      for (uint32_t i = 0; i < T; i++) state_[i] = inner_product_const(s, sp.mds[i]);
    }
  }
  Context& ctx_;
  std::array<uint32_t, T> state_;
  std::vector<uint32_t> buf_;
};
}  // namespace poseidon
inline Assignment poseidon_hash_two_closure(const FlexGateCS& cs, const Fr& x, const Fr& y) {
  Assignment asg(cs);
  Context ctx(asg);
  const uint32_t xc = ctx.load_witness(x), yc = ctx.load_witness(y);
  poseidon::Chip chip(ctx);
  chip.update({xc, yc});
  const uint32_t out = chip.squeeze();
  ctx.finish({xc, yc, out});
  return asg;
}

// ---- keys -----------------------------------------------------------------------------------------------------------
struct FlexKeys {
  FlexGateCS cs;
  poly::EvaluationDomain domain;
  uint32_t u;  // usable rows
  plonk::VerifyingKey vk;
  std::vector<Dev> fixed_values, fixed_polys, fixed_cosets, sigma_values, sigma_polys, sigma_cosets;
  Dev l0, l_last, l_active;
  Dev table_sorted, table_sorted_mont, table_mult, active_rows;
  uint32_t n_unique = 0, n_active = 0;
  FlexKeys(const FlexGateCS& c, uint32_t k) : cs(c), domain(c.degree, k), u(((uint32_t)1 << k) - (c.blinding_factors + 1)) {}
};

inline std::unique_ptr<FlexKeys> keygen(const poly::kzg::ParamsKZG& params, const FlexGateCS& cs, const Assignment& asg) {
  using namespace plonk::detail;
  std::unique_ptr<FlexKeys> pk(new FlexKeys(cs, params.k()));
  const poly::EvaluationDomain& dom = pk->domain;
  const size_t n = params.n();
  const uint32_t u = pk->u;
  // fixed columns
  for (uint32_t c = 0; c < cs.n_fixed; c++) {
    Dev d = zeros(n);
    if ((int)c == cs.col_table) {
      if (asg.table_values.size() > u) throw Error(H2MI_ERANGE, "lookup table larger than the usable rows (LOOKUP_BITS must be below DEGREE)");
      std::vector<Fr> tv(asg.table_values.size());
      for (size_t i = 0; i < tv.size(); i++) tv[i] = fr::from_u64(asg.table_values[i]);
      if (!tv.empty()) check(h2mi_memcpy_h2d(d->p, tv.data(), tv.size() * 32), "table column");
    } else {
      for (const auto& kv : asg.fixed[c]) patch(*d, kv.first, kv.second);
    }
    pk->fixed_values.push_back(std::move(d));
  }
  // sigma columns from the copy constraints (Assembly::copy over constants, advice and instance cells alike)
  const uint32_t m = (uint32_t)cs.perm_columns.size();
  auto perm_index = [&](Kind kind, uint32_t col) {
    for (uint32_t j = 0; j < m; j++)
      if (cs.perm_columns[j].kind == kind && cs.perm_columns[j].index == col) return j;
    throw Error(H2MI_EINVAL, "copy constraint on a column without equality enabled");
  };
  plonk::PermutationAssembly asm_;
  for (const auto& c : asg.copies)
    asm_.copy(plonk::Cell(perm_index(c.first.kind, c.first.col), c.first.row), plonk::Cell(perm_index(c.second.kind, c.second.col), c.second.row));
  {
    DeviceVec omega_pows(n);
    check(h2mi_fr_powers_dev(omega_pows.p, n, dom.get_omega().l, nullptr), "powers");
    const Fr delta = plonk::fr_delta();
    for (uint32_t j = 0; j < m; j++) {
      Dev d(new DeviceVec(n));
      const void* ptrs[1] = {omega_pows.p};
      Fr sc = fr::pow_u64(delta, j);
      check(h2mi_fr_lincomb_dev(ptrs, sc.l, 1, n, d->p, nullptr), "identity permutation");
      pk->sigma_values.push_back(std::move(d));
    }
    std::vector<uint32_t> pos;
    for (const auto& kv : asm_.mapping()) {
      if (kv.first == kv.second) continue;
      patch(*pk->sigma_values[kv.first.first], kv.first.second, fr::mul(fr::pow_u64(delta, kv.second.first), fr::pow_u64(dom.get_omega(), kv.second.second)));
      if (kv.first.second < u) pos.push_back((kv.first.first / cs.chunk) * u + kv.first.second);
    }
    check(h2mi_sync(), "sync");
    std::sort(pos.begin(), pos.end());
    pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
    pk->n_active = (uint32_t)pos.size();
    pk->active_rows.reset(new DeviceVec(pos.size() / 8 + 1));
    if (!pos.empty()) check(h2mi_memcpy_h2d(pk->active_rows->p, pos.data(), pos.size() * 4), "active rows");
  }
  // verifying key
  {
    std::vector<const void*> fc, sc;
    for (auto& d : pk->fixed_values) fc.push_back(d->p);
    for (auto& d : pk->sigma_values) sc.push_back(d->p);
    pk->vk.k = params.k();
    pk->vk.cs_degree = cs.degree;
    pk->vk.fixed_commitments = commit_points(params.g_lagrange_handle(), fc, n);
    pk->vk.permutation_commitments = commit_points(params.g_lagrange_handle(), sc, n);
    pk->vk.compute_transcript_repr();
  }
  for (auto& col : pk->fixed_values) {
    Dev p, e;
    to_poly_and_coset(dom, *col, p, e);
    pk->fixed_polys.push_back(std::move(p));
    pk->fixed_cosets.push_back(std::move(e));
  }
  for (auto& col : pk->sigma_values) {
    Dev p, e;
    to_poly_and_coset(dom, *col, p, e);
    pk->sigma_polys.push_back(std::move(p));
    pk->sigma_cosets.push_back(std::move(e));
  }
  {
    Dev l0 = zeros(n), ll = zeros(n), la(new DeviceVec(n)), unused;
    patch(*l0, 0, fr::ONE);
    patch(*ll, u, fr::ONE);
    check(h2mi_fr_fill_dev(la->p, n, fr::ONE.l, nullptr), "fill");
    check(h2mi_memset_zero((char*)la->p + (size_t)u * 32, (n - u) * 32), "zero");
    to_poly_and_coset(dom, *l0, unused, pk->l0);
    to_poly_and_coset(dom, *ll, unused, pk->l_last);
    to_poly_and_coset(dom, *la, unused, pk->l_active);
    check(h2mi_sync(), "sync");
  }
  if (cs.lookup) {  // the table's distinct values in ascending order with their multiplicities over the usable rows
    std::map<uint64_t, uint32_t> counts;
    for (uint64_t v : asg.table_values) counts[v]++;
    counts[0] += u - (uint32_t)asg.table_values.size();
    std::vector<uint64_t> canon;
    std::vector<Fr> mont;
    std::vector<uint32_t> mult;
    for (const auto& kv : counts) {
      canon.insert(canon.end(), {kv.first, 0, 0, 0});
      mont.push_back(fr::from_u64(kv.first));
      mult.push_back(kv.second);
    }
    pk->n_unique = (uint32_t)mult.size();
    pk->table_sorted.reset(new DeviceVec(mult.size()));
    pk->table_sorted_mont.reset(new DeviceVec(mult.size()));
    pk->table_mult.reset(new DeviceVec(mult.size() / 8 + 1));
    check(h2mi_memcpy_h2d(pk->table_sorted->p, canon.data(), canon.size() * 8), "table");
    check(h2mi_memcpy_h2d(pk->table_sorted_mont->p, mont.data(), mont.size() * 32), "table");
    check(h2mi_memcpy_h2d(pk->table_mult->p, mult.data(), mult.size() * 4), "table");
  }
  return pk;
}

// ---- create_proof ---------------------------------------------------------------------------------------------------
struct FlexWorkspace {  // device buffers of one create_proof, handed out in request order and kept for the next proof
  std::vector<Dev> pool, shplonk_q, shplonk_s;
  size_t cursor = 0;
  Dev points, nx, tmp, h_x, l_x, h2_x;
  std::unique_ptr<plonk::ShplonkLanes> lanes;
  h2mi_stream_t side = nullptr;
  FlexWorkspace(const FlexWorkspace&) = delete;
  FlexWorkspace& operator=(const FlexWorkspace&) = delete;
  explicit FlexWorkspace(const FlexKeys& pk) {
    const size_t n = (size_t)1 << pk.domain.k();
    auto vec = [&](size_t cnt) { return Dev(new DeviceVec(cnt)); };
    points = vec(24);  // 8 x 96 B
    nx = vec(n); tmp = vec(n); h_x = vec(n); l_x = vec(n); h2_x = vec(n);
    for (int i = 0; i < 6; i++) shplonk_q.push_back(vec(n));
    for (int i = 0; i < 6; i++) shplonk_s.push_back(vec(n));
    lanes.reset(new plonk::ShplonkLanes(n));
    check(h2mi_stream_create(&side), "stream_create");
  }
  ~FlexWorkspace() {
    if (side) h2mi_stream_destroy(side);
  }
  DeviceVec& take(size_t count) {
    if (cursor == pool.size()) pool.push_back(Dev(new DeviceVec(count)));
    DeviceVec& d = *pool[cursor++];
    if (d.n != count) throw Error(H2MI_EINVAL, "workspace reused with another proving key");
    return d;
  }
};

inline void create_proof(const poly::kzg::ParamsKZG& params, const FlexKeys& pk, const Assignment& asg, uint64_t seed, transcript::Blake2bWrite& tr,
                         FlexWorkspace* workspace = nullptr) {
  using namespace plonk;
  using namespace plonk::detail;
  std::unique_ptr<FlexWorkspace> own;
  if (!workspace) {
    own.reset(new FlexWorkspace(pk));
    workspace = own.get();
  }
  FlexWorkspace& ws = *workspace;
  ws.cursor = 0;
  const FlexGateCS& cs = pk.cs;
  const poly::EvaluationDomain& d = pk.domain;
  const size_t n = params.n(), ext = d.extended_len();
  const uint32_t bf = cs.blinding_factors, u = pk.u;
  const Fr omega = d.get_omega(), omega_inv = d.get_omega_inv();
  DeviceVec& points = *ws.points;
  auto write_points = [&](size_t k) {
    std::vector<G1> jac(k);
    check(h2mi_memcpy_d2h(jac.data(), points.p, k * 96), "d2h");  // joins the MSM pipeline
    for (const G1Affine& a : normalize_host_batch(jac)) tr.write_point(a);
  };
  auto commit = [&](bool lagrange, const void* col, size_t slot) {
    check(h2mi_msm_bn254_g1_dev(lagrange ? params.g_lagrange_handle() : params.g_handle(), col, n, (char*)points.p + 96 * slot, nullptr), "commit");
  };
  struct Forms {
    DeviceVec *poly, *coset;
  };
  auto forms = [&](const DeviceVec& col, h2mi_stream_t stream) {
    Forms f{&ws.take(n), &ws.take(ext)};
    to_poly_and_coset_into(d, col, *f.poly, *f.coset, stream);
    return f;
  };

  tr.common_scalar(pk.vk.transcript_repr);
  for (const Fr& v : asg.instance) tr.common_scalar(v);  // KZG: public inputs are hashed as scalars, not committed
  DeviceVec& instance = ws.take(n);
  check(h2mi_memset_zero(instance.p, n * 32), "zero");
  if (!asg.instance.empty()) check(h2mi_memcpy_h2d_async(instance.p, asg.instance.data(), asg.instance.size() * 32), "instance");
  // the advice column + blinding rows
  if (asg.advice.size() > u) throw Error(H2MI_ERANGE, "assignment reaches into the blinding rows");
  DeviceVec& advice = ws.take(n);
  check(h2mi_memset_zero(advice.p, n * 32), "zero");
  check(h2mi_memcpy_h2d(advice.p, asg.advice.data(), asg.advice.size() * 32), "advice cells");
  std::vector<Fr> blind = uniform_fr(seed + 1, bf + 1);
  check(h2mi_memcpy_h2d_async((char*)advice.p + (size_t)u * 32, blind.data(), (bf + 1) * 32), "blinding rows");
  // the one advice column: a lone commitment read back next — in order on one stream, nothing deferred
  check(h2mi_msm_bn254_g1_inorder_dev(params.g_lagrange_handle(), advice.p, n, points.p, nullptr), "commit");
  check(h2mi_msm_flush(), "flush");  // the bucket reductions start now, not when the host reaches the join
  // coefficient / extended forms that wait for no challenge: on the side stream, beside the transcript round trips
  check(h2mi_stream_wait(ws.side, nullptr), "stream_wait");
  Forms advice_f = forms(advice, ws.side);
  Forms instance_f{nullptr, nullptr};
  if (asg.instance.size() <= 16) {  // a handful of public inputs: sum_r v_r * (l_0's coset rotated by r rows), no transform
    instance_f.coset = &ws.take(ext);
    check(h2mi_plonk_instance_coset_dev(pk.l0->p, d.k(), d.extended_k(), (const uint64_t*)asg.instance.data(), asg.instance.size(), instance_f.coset->p,
                                        ws.side),
          "instance coset");
  } else {
    instance_f = forms(instance, ws.side);
  }
  write_points(1);
  (void)tr.squeeze_challenge();  // theta
  // lookup: input expression rows q_lookup * a, permuted input / table columns
  DeviceVec *lk_input = nullptr, *a_perm = nullptr, *s_perm = nullptr, *lz = nullptr;
  Forms ap_f{nullptr, nullptr}, sp_f{nullptr, nullptr}, lz_f{nullptr, nullptr};
  if (cs.lookup) {
    lk_input = &ws.take(n);
    check(h2mi_fr_mul_dev(pk.fixed_values[cs.col_qlookup]->p, advice.p, n, lk_input->p, nullptr), "lookup input");
    a_perm = &ws.take(n);
    s_perm = &ws.take(n);
    uint64_t missing = 0;
    check(h2mi_plonk_lookup_permute_dev(lk_input->p, pk.table_sorted->p, pk.table_sorted_mont->p, pk.table_mult->p, pk.n_unique, d.k(), u, a_perm->p,
                                        s_perm->p, &missing, nullptr), "lookup_permute");
    if (missing) throw Error(H2MI_EINVAL, "lookup input not in the table (ConstraintSystemFailure)");
    std::vector<Fr> lb = uniform_fr(seed + 4, 2 * (bf + 1));
    {
      std::vector<void*> cells;
      for (uint32_t r = 0; r <= bf; r++) cells.push_back((char*)a_perm->p + (size_t)(u + r) * 32);
      for (uint32_t r = 0; r <= bf; r++) cells.push_back((char*)s_perm->p + (size_t)(u + r) * 32);
      check(h2mi_fr_patch_cells_dev(cells.data(), (const uint64_t*)lb.data(), cells.size(), nullptr), "blinding rows");
    }
    {  // the permuted pair is all this phase commits and is read back next
      const void* cols[2] = {a_perm->p, s_perm->p};
      check(h2mi_msm_bn254_g1_phase_dev(params.g_lagrange_handle(), cols, 2, n, points.p, H2MI_MSM_INORDER, nullptr), "commit");
    }
    check(h2mi_msm_flush(), "flush");
    check(h2mi_stream_wait(ws.side, nullptr), "stream_wait");
    ap_f = forms(*a_perm, ws.side);
    sp_f = forms(*s_perm, ws.side);
    write_points(2);
  }
  const Fr beta = tr.squeeze_challenge(), gamma = tr.squeeze_challenge();
  // the random polynomial's dense commitment is queued before the grand products (its point is written after theirs)
  const uint32_t m = (uint32_t)cs.perm_columns.size(), n_sets = (m + cs.chunk - 1) / cs.chunk;
  DeviceVec& random_poly = ws.take(n);
  check(h2mi_fr_random_dev(random_poly.p, n, seed + 3, 0, nullptr), "random_poly");
  commit(false, random_poly.p, n_sets + (cs.lookup ? 1 : 0));
  // permutation argument over the copy constraints' support
  const Fr delta = fr_delta();
  auto value_col = [&](const Col& c) -> const void* {
    return c.kind == ADVICE ? advice.p : c.kind == INSTANCE ? instance.p : pk.fixed_values[c.index]->p;
  };
  std::vector<const void*> vals, sigs;
  std::vector<void*> zptr;
  std::vector<Fr> bd;
  std::vector<DeviceVec*> zs;
  for (uint32_t j = 0; j < m; j++) {
    vals.push_back(value_col(cs.perm_columns[j]));
    sigs.push_back(pk.sigma_values[j]->p);
    bd.push_back(fr::mul(beta, fr::pow_u64(delta, j)));
  }
  for (uint32_t s = 0; s < n_sets; s++) {
    zs.push_back(&ws.take(n));
    zptr.push_back(zs.back()->p);
  }
  check(h2mi_plonk_permutation_products_sparse_dev(vals.data(), sigs.data(), m, cs.chunk, d.k(), u, beta.l, gamma.l, (const uint64_t*)bd.data(), omega.l,
                                                   pk.active_rows->p, pk.n_active, zptr.data(), nullptr), "permutation_products");
  std::vector<Fr> zblind = uniform_fr(seed + 2, (size_t)n_sets * bf);
  {  // every grand product's blinding rows from one launch's arguments (h2mi_fr_patch_cells_dev)
    std::vector<void*> cells;
    for (uint32_t s = 0; s < n_sets; s++)
      for (uint32_t r = 0; r < bf; r++) cells.push_back((char*)zs[s]->p + (size_t)(u + 1 + r) * 32);
    check(h2mi_fr_patch_cells_dev(cells.data(), (const uint64_t*)zblind.data(), cells.size(), nullptr), "z blinding rows");
  }
  // the grand products' coefficient / extended forms: side stream, behind the columns and AHEAD of their commitments'
  // partition kernels (which a dense accumulation in flight starves for milliseconds at DEGREE 22)
  check(h2mi_stream_wait(ws.side, nullptr), "stream_wait");
  std::vector<Forms> z_f;
  for (uint32_t s = 0; s < n_sets; s++) z_f.push_back(forms(*zs[s], ws.side));
  size_t slot = 0;
  {  // the grand products' commitments in one call: constant but for the copy constraints, so batched at every size
    std::vector<const void*> cols;
    for (uint32_t s = 0; s < n_sets; s++) cols.push_back(zs[s]->p);
    check(h2mi_msm_bn254_g1_batch_sparse_dev(params.g_lagrange_handle(), cols.data(), cols.size(), n, (char*)points.p + 96 * slot, nullptr), "commit");
    slot += n_sets;
  }
  std::vector<Fr> lzblind;
  if (cs.lookup) {
    lz = &ws.take(n);
    check(h2mi_plonk_lookup_product_dev(lk_input->p, pk.fixed_values[cs.col_table]->p, a_perm->p, s_perm->p, d.k(), u, beta.l, gamma.l, lz->p, nullptr),
          "lookup_product");
    lzblind = uniform_fr(seed + 5, bf);
    check(h2mi_memcpy_h2d_async((char*)lz->p + (size_t)(u + 1) * 32, lzblind.data(), bf * 32), "lookup z blinding rows");
    check(h2mi_stream_wait(ws.side, nullptr), "stream_wait");
    lz_f = forms(*lz, ws.side);
    commit(true, lz->p, slot++);
  }
  slot++;  // the random polynomial's slot
  check(h2mi_msm_flush(), "flush");
  write_points(slot);
  const Fr y = tr.squeeze_challenge();
  // joined AFTER the read-back: the copy runs on the library stream, and a join in front of it made the transcript wait for every
  // transform of the side stream instead of the bucket reductions only
  check(h2mi_stream_wait(nullptr, ws.side), "stream_wait");
  // quotient
  DeviceVec& h = ws.take(ext);
  {
    auto coset_col = [&](const Col& c) -> const void* {
      return c.kind == ADVICE ? advice_f.coset->p : c.kind == INSTANCE ? instance_f.coset->p : pk.fixed_cosets[c.index]->p;
    };
    h2mi_range_cosets rc;
    std::memset(&rc, 0, sizeof(rc));
    rc.a = advice_f.coset->p;
    rc.q = pk.fixed_cosets[cs.col_q]->p;
    for (uint32_t j = 0; j < m; j++) {
      rc.perm_value[j] = coset_col(cs.perm_columns[j]);
      rc.perm_sigma[j] = pk.sigma_cosets[j]->p;
    }
    for (uint32_t s = 0; s < n_sets; s++) rc.perm_z[s] = z_f[s].coset->p;
    rc.l0 = pk.l0->p;
    rc.l_last = pk.l_last->p;
    rc.l_active = pk.l_active->p;
    rc.n_perm = m;
    rc.chunk_len = cs.chunk;
    rc.has_lookup = cs.lookup ? 1 : 0;
    if (cs.lookup) {
      rc.lookup_selector = pk.fixed_cosets[cs.col_qlookup]->p;
      rc.table = pk.fixed_cosets[cs.col_table]->p;
      rc.lookup_permuted_input = ap_f.coset->p;
      rc.lookup_permuted_table = sp_f.coset->p;
      rc.lookup_z = lz_f.coset->p;
    }
    const Fr& zeta = d.get_g_coset();
    const std::vector<Fr>& t_inv = d.t_inv();  // (X^n - 1)^-1 on the coset: cached in the domain
    check(h2mi_plonk_evaluate_h_range_dev(&rc, d.k(), d.extended_k(), bf, beta.l, gamma.l, y.l, delta.l, zeta.l, d.get_extended_omega().l,
                                          (const uint64_t*)t_inv.data(), h.p, nullptr), "evaluate_h");
    check(h2mi_ntt_bn254_fr_dev(h.p, d.extended_k(), d.get_extended_omega_inv().l, nullptr, nullptr, nullptr), "extended_to_coeff");
    check(h2mi_fr_scale_powers_dev(h.p, ext, d.get_g_coset_inv().l, d.get_extended_ifft_divisor().l, nullptr), "distribute_powers_zeta");
  }
  const uint32_t pieces = cs.degree - 1;
  {
    std::vector<const void*> cols;
    for (uint32_t i = 0; i < pieces; i++) cols.push_back((char*)h.p + (size_t)i * n * 32);
    check(h2mi_msm_bn254_g1_phase_dev(params.g_handle(), cols.data(), cols.size(), n, points.p, H2MI_MSM_INORDER, nullptr), "commit");
  }
  write_points(pieces);
  const Fr x = tr.squeeze_challenge();
  const Fr xn = fr::pow_u64(x, n);
  auto rot = [&](int64_t r) { return fr::mul(x, pow_signed(omega, omega_inv, r)); };
  const Fr x_next = rot(1), x_last = rot(-(int64_t)(bf + 1)), x_inv = rot(-1);
  DeviceVec& h_poly = ws.take(n);
  {
    std::vector<const void*> ptrs;
    std::vector<Fr> sc;
    Fr p = fr::ONE;
    for (uint32_t i = 0; i < pieces; i++) {
      ptrs.push_back((char*)h.p + (size_t)i * n * 32);
      sc.push_back(p);
      p = fr::mul(p, xn);
    }
    check(h2mi_fr_lincomb_dev(ptrs.data(), (const uint64_t*)sc.data(), pieces, n, h_poly.p, nullptr), "h_poly");
  }
  struct Q {
    const DeviceVec* poly;
    Fr point;
  };
  std::vector<Q> written;
  for (const Query& q : cs.advice_queries) written.push_back({advice_f.poly, rot(q.second)});
  for (const Query& q : cs.fixed_queries) written.push_back({pk.fixed_polys[q.first].get(), rot(q.second)});
  written.push_back({&random_poly, x});
  for (auto& sp : pk.sigma_polys) written.push_back({sp.get(), x});
  for (uint32_t i = 0; i < n_sets; i++) {
    written.push_back({z_f[i].poly, x});
    written.push_back({z_f[i].poly, x_next});
    if (i + 1 < n_sets) written.push_back({z_f[i].poly, x_last});
  }
  if (cs.lookup) {
    written.push_back({lz_f.poly, x});
    written.push_back({lz_f.poly, x_next});
    written.push_back({ap_f.poly, x});
    written.push_back({ap_f.poly, x_inv});
    written.push_back({sp_f.poly, x});
  }
  std::vector<Q> todo = written;
  todo.push_back({&h_poly, x});
  // one launch per distinct point over the distinct polynomials opened there
  std::vector<Fr> distinct;
  for (const Q& q : todo)
    if (!contains(distinct, q.point)) distinct.push_back(q.point);
  struct Slot {
    const DeviceVec* poly;
    Fr point;
    size_t slot;
  };
  std::vector<Slot> slots;
  DeviceVec& evals = ws.take(todo.size() + 8);
  {  // every evaluation in one call (h2mi_fr_eval_polys_multi_dev), grouped by distinct point
    std::vector<const void*> polys;
    std::vector<size_t> counts;
    for (const Fr& pt : distinct) {
      const size_t first = slots.size();
      for (const Q& q : todo) {
        if (!(q.point == pt)) continue;
        bool seen = false;
        for (size_t i = first; i < slots.size(); i++) seen = seen || slots[i].poly == q.poly;
        if (seen) continue;
        slots.push_back({q.poly, pt, slots.size()});
        polys.push_back(q.poly->p);
      }
      counts.push_back(slots.size() - first);
    }
    check(h2mi_fr_eval_polys_multi_dev(polys.data(), counts.data(), (const uint64_t*)distinct.data(), distinct.size(), n, evals.p, nullptr), "eval");
  }
  std::vector<Fr> ev(slots.size());
  check(h2mi_memcpy_d2h(ev.data(), evals.p, slots.size() * 32), "d2h");
  auto value_of = [&](const DeviceVec* poly, const Fr& pt) {
    for (const Slot& s : slots)
      if (s.poly == poly && s.point == pt) return ev[s.slot];
    throw Error(H2MI_EINVAL, "query without an evaluation");
  };
  for (const Q& q : written) tr.write_scalar(value_of(q.poly, q.point));
  std::vector<ProverQuery> queries;
  auto q = [&](const DeviceVec* poly, const Fr& pt) { queries.push_back({poly, pt, value_of(poly, pt)}); };
  for (const Query& aq : cs.advice_queries) q(advice_f.poly, rot(aq.second));
  for (uint32_t i = 0; i < n_sets; i++) {
    q(z_f[i].poly, x);
    q(z_f[i].poly, x_next);
  }
  for (uint32_t i = n_sets - 1; i-- > 0;) q(z_f[i].poly, x_last);
  if (cs.lookup) {
    q(lz_f.poly, x);
    q(ap_f.poly, x);
    q(sp_f.poly, x);
    q(ap_f.poly, x_inv);
    q(lz_f.poly, x_next);
  }
  for (const Query& fq_ : cs.fixed_queries) q(pk.fixed_polys[fq_.first].get(), rot(fq_.second));
  for (auto& sp : pk.sigma_polys) q(sp.get(), x);
  q(&h_poly, x);
  q(&random_poly, x);
  ShplonkScratch scratch{ws.nx.get(), ws.tmp.get(), ws.h_x.get(), ws.l_x.get(), ws.h2_x.get(), &ws.shplonk_q, &ws.shplonk_s, ws.lanes->lanes()};
  shplonk_create_proof(n, tr, queries, [&](DeviceVec& poly) {
    // a lone commitment, read back at once: in order on one stream, nothing deferred
    check(h2mi_msm_bn254_g1_inorder_dev(params.g_handle(), poly.p, n, points.p, nullptr), "commit");
    write_points(1);
  }, scratch);
  check(h2mi_sync(), "sync");
}

}  // namespace flex
}  // namespace h2mi
