// h2mi_flex.hpp — C++17 caller of the library's prover (h2mi_prover.h) for the halo2-lib builders the reference proves through
// `scaffold::prove` (src/scaffold.rs:246-366: GateWithInstanceCircuitBuilder / RangeWithInstanceCircuitBuilder, :379-485): the
// constraint systems halo2-base configures for ONE gate advice column as data (h2mi_constraint_system), its Context (cell layout
// of load_witness / mul / add / mul_add / range_check — witness generation, the caller's job in a Rust fork too), and keygen /
// create_proof as thin calls: fixed cells + copy constraints into h2mi_prover_keygen, witness cells + public inputs into the seven
// phase calls of plonk::drive_proof.
//
// The cell layout is the same restatement as the Python host's (halo2-scaffold_amd/flex.py — see its header for what is recalled
// from the un-vendored halo2-base and what the reference itself shows) and the oracle's (oracle/flex.py): identical proof bytes
// (tests/test_gpu_flex.py).  Closures: examples/halo2_lib.rs:14-60 and examples/range.rs:10-34.
//
//   Gate builder:  fixed 0 constants, 1 q_enable;                         degree 3, permutation sets of one, 2 h pieces
//   Range builder: fixed 0 table, 1 constants, 2 q_lookup, 3 q_enable;    the lookup of q_lookup * a makes the degree 5:
//                  one permutation set of three, 4 h pieces, extended domain 4n
// rng stand-in: `seed`, handed to the library (h2mi_prover.h).
#pragma once
#include <array>

#include "h2mi_plonk.hpp"

namespace h2mi {
namespace flex {

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
  // the same constraint system as the numbers create_proof reads off it (h2mi_prover.h)
  h2mi_constraint_system abi(uint32_t k) const {
    h2mi_constraint_system cs;
    std::memset(&cs, 0, sizeof(cs));
    cs.k = k;
    cs.n_advice = 1;
    cs.n_fixed = n_fixed;
    cs.n_instance = 1;
    cs.degree = degree;
    cs.blinding_factors = blinding_factors;
    cs.gates = H2MI_GATES_FLEX_VERTICAL;
    cs.n_gates = 1;
    cs.gate_advice[0] = 0;
    cs.gate_selector[0] = (uint32_t)col_q;
    cs.n_perm = (uint32_t)perm_columns.size();
    for (size_t j = 0; j < perm_columns.size(); j++) cs.perm_columns[j] = {(uint32_t)perm_columns[j].kind, perm_columns[j].index};
    if (lookup) {
      cs.n_lookups = 1;
      cs.lookups[0].input = {H2MI_COL_ADVICE, 0};
      cs.lookups[0].selector_fixed = col_qlookup;
      cs.lookups[0].table_fixed = (uint32_t)col_table;
    }
    cs.n_advice_queries = (uint32_t)advice_queries.size();
    for (size_t i = 0; i < advice_queries.size(); i++) cs.advice_queries[i] = {advice_queries[i].first, advice_queries[i].second};
    cs.n_fixed_queries = (uint32_t)fixed_queries.size();
    for (size_t i = 0; i < fixed_queries.size(); i++) cs.fixed_queries[i] = {fixed_queries[i].first, fixed_queries[i].second};
    return cs;
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
  plonk::VerifyingKey vk;
  plonk::PkHandle pk;
  explicit FlexKeys(const FlexGateCS& c) : cs(c) {}
};

// keygen_vk + keygen_pk (src/scaffold.rs:284,287) from the cells a run of the closure assigns: the fixed columns (the table
// column dense, the others sparse) and the copy constraints, columns renumbered into the permutation argument's order
inline std::unique_ptr<FlexKeys> keygen(const poly::kzg::ParamsKZG& params, const FlexGateCS& cs, const Assignment& asg) {
  std::unique_ptr<FlexKeys> keys(new FlexKeys(cs));
  plonk::KeygenInput in;
  for (uint32_t c = 0; c < cs.n_fixed; c++) {
    if ((int)c == cs.col_table) {
      plonk::ColumnCells tv;
      for (uint64_t v : asg.table_values) tv.values.push_back(fr::from_u64(v));
      in.fixed.push_back(std::move(tv));  // rows 0 .. size - 1
    } else {
      in.fixed.push_back(plonk::ColumnCells(asg.fixed[c]));
    }
  }
  const uint32_t m = (uint32_t)cs.perm_columns.size();
  auto perm_index = [&](Kind kind, uint32_t col) {
    for (uint32_t j = 0; j < m; j++)
      if (cs.perm_columns[j].kind == kind && cs.perm_columns[j].index == col) return j;
    throw Error(H2MI_EINVAL, "copy constraint on a column without equality enabled");
  };
  for (const auto& c : asg.copies)
    in.copies.insert(in.copies.end(), {perm_index(c.first.kind, c.first.col), c.first.row, perm_index(c.second.kind, c.second.col), c.second.row});
  plonk::run_keygen(cs.abi(params.k()), params, in, 0, keys->pk, keys->vk);
  return keys;
}

// ---- create_proof ---------------------------------------------------------------------------------------------------
struct FlexWorkspace : plonk::ProverWorkspace {  // the prover's device buffers, kept for the next proof against the same key
  FlexWorkspace(const poly::kzg::ParamsKZG& params, const FlexKeys& pk) : plonk::ProverWorkspace(params, pk.pk) {}
};

inline void create_proof(const poly::kzg::ParamsKZG& params, const FlexKeys& pk, const Assignment& asg, uint64_t seed, transcript::Blake2bWrite& tr,
                         FlexWorkspace* workspace = nullptr) {
  std::unique_ptr<FlexWorkspace> own;
  if (!workspace) {
    own.reset(new FlexWorkspace(params, pk));
    workspace = own.get();
  }
  tr.common_scalar(pk.vk.transcript_repr);
  for (const Fr& v : asg.instance) tr.common_scalar(v);  // KZG: public inputs are hashed as scalars, not committed
  const std::vector<h2mi_column_cells> advice = {{nullptr, (const uint64_t*)asg.advice.data(), asg.advice.size(), 0}};  // rows 0 .. size - 1
  plonk::drive_proof(*workspace, advice, asg.instance, seed, tr);
}

}  // namespace flex
}  // namespace h2mi
