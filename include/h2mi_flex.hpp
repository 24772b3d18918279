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
//   several gate columns (`configure`, round 5 in C++; round 4 in the Python host): one selector per gate column, lookup-advice
//                  columns instead of q_lookup: degree 4 with lookups, 3 without
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
  uint32_t num_advice = 1, num_lookup_advice = 0;  // gate columns; lookup-advice columns (only with several gate columns)
  uint32_t num_fixed = 1;                          // constants columns: the distinct constants are dealt out round-robin over them
  uint32_t k = 0, minimum_rows = 9;                // the multi-column layout needs the row budget 2^k - minimum_rows
  uint32_t n_advice = 1, n_fixed;
  int col_table = -1, col_const, col_qlookup = -1;
  std::vector<uint32_t> col_qs;  // the gate selectors, one per gate column
  std::vector<Col> perm_columns;
  std::vector<Query> advice_queries, fixed_queries;
  uint32_t degree, blinding_factors = 6, chunk;
  // one gate column: what every example of the reference configures at its DEGREE (see halo2-scaffold_amd/flex.py FlexGateCS)
  explicit FlexGateCS(bool with_lookup) : lookup(with_lookup) {
    if (lookup) {
      col_table = 0; col_const = 1; col_qlookup = 2;
      col_qs = {3};
      n_fixed = 4;
      fixed_queries = {{1, 0}, {0, 0}, {2, 0}, {3, 0}};
    } else {
      col_const = 0;
      col_qs = {1};
      n_fixed = 2;
      fixed_queries = {{0, 0}, {1, 0}};
    }
    perm_columns = {{FIXED, (uint32_t)col_const}, {ADVICE, 0}, {INSTANCE, 0}};
    advice_queries = {{0, 0}, {0, 1}, {0, 2}, {0, 3}};
    degree = lookup ? 5 : 3;
    chunk = degree - 2;
  }
  // several gate columns (what `builder.config(k, Some(minimum_rows))`, src/scaffold.rs:268, configures when the cells overflow
  // 2^k - minimum_rows rows [RECALL halo2-base 0.3]): per gate column an advice column with its own selector and vertical gate; the
  // Range builder then looks up dedicated lookup-advice columns instead of q_lookup * a: degree 4, permutation sets of two
  // constants_columns: config's ceil(distinct constants / 2^k), allocated where the single constants column was
  FlexGateCS(bool with_lookup, uint32_t gate_columns, uint32_t lookup_columns, uint32_t k_, uint32_t minimum_rows_ = 9, uint32_t constants_columns = 1)
      : FlexGateCS(with_lookup) {
    k = k_;
    minimum_rows = minimum_rows_;
    if (gate_columns <= 1) return;
    num_fixed = std::max(1u, constants_columns);
    const uint32_t A = gate_columns, Lc = lookup_columns;
    if (A > H2MI_MAX_GATES || Lc > H2MI_MAX_LOOKUPS || (Lc >= 1 && !lookup))  // Range builder, nothing looked up: no lookup-advice column
      throw Error(H2MI_EINVAL, "FlexGateCS: up to 32 gate columns and 8 lookup-advice columns (none for the Gate builder)");
    num_advice = A;
    num_lookup_advice = Lc;
    n_advice = A + Lc;
    col_table = lookup ? 0 : -1;
    col_const = lookup ? 1 : 0;
    col_qlookup = -1;
    col_qs.clear();
    for (uint32_t j = 0; j < A; j++) col_qs.push_back((uint32_t)col_const + num_fixed + j);
    n_fixed = (uint32_t)col_const + num_fixed + A;
    fixed_queries.clear();
    for (uint32_t c = 0; c < num_fixed; c++) fixed_queries.push_back({(uint32_t)col_const + c, 0});
    if (Lc) fixed_queries.push_back({(uint32_t)col_table, 0});  // the table column is queried by the lookup arguments only
    for (uint32_t q : col_qs) fixed_queries.push_back({q, 0});
    perm_columns.clear();
    for (uint32_t c = 0; c < num_fixed; c++) perm_columns.push_back({FIXED, (uint32_t)col_const + c});
    for (uint32_t j = 0; j < A + Lc; j++) perm_columns.push_back({ADVICE, j});
    perm_columns.push_back({INSTANCE, 0});
    advice_queries.clear();
    for (uint32_t j = 0; j < A; j++)
      for (int32_t r = 0; r < 4; r++) advice_queries.push_back({j, r});
    for (uint32_t l = 0; l < Lc; l++) advice_queries.push_back({A + l, 0});
    degree = Lc ? 4 : 3;
    chunk = degree - 2;
  }
  // the same constraint system as the numbers create_proof reads off it (h2mi_prover.h)
  h2mi_constraint_system abi(uint32_t k_) const {
    h2mi_constraint_system cs;
    std::memset(&cs, 0, sizeof(cs));
    cs.k = k_;
    cs.n_advice = n_advice;
    cs.n_fixed = n_fixed;
    cs.n_instance = 1;
    cs.degree = degree;
    cs.blinding_factors = blinding_factors;
    cs.gates = H2MI_GATES_FLEX_VERTICAL;
    cs.n_gates = (uint32_t)col_qs.size();
    for (size_t g = 0; g < col_qs.size(); g++) {
      cs.gate_advice[g] = (uint32_t)g;
      cs.gate_selector[g] = col_qs[g];
    }
    cs.n_perm = (uint32_t)perm_columns.size();
    for (size_t j = 0; j < perm_columns.size(); j++) cs.perm_columns[j] = {(uint32_t)perm_columns[j].kind, perm_columns[j].index};
    if (lookup && num_advice == 1) {
      cs.n_lookups = 1;
      cs.lookups[0].input = {H2MI_COL_ADVICE, 0};
      cs.lookups[0].selector_fixed = col_qlookup;
      cs.lookups[0].table_fixed = (uint32_t)col_table;
    } else {
      cs.n_lookups = num_lookup_advice;
      for (uint32_t l = 0; l < num_lookup_advice; l++) {
        cs.lookups[l].input = {H2MI_COL_ADVICE, num_advice + l};
        cs.lookups[l].selector_fixed = -1;
        cs.lookups[l].table_fixed = (uint32_t)col_table;
      }
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
  std::vector<std::vector<Fr>> advice;        // per advice column: rows 0 .. (every layout fills its columns from the top)
  std::vector<std::map<uint32_t, Fr>> fixed;  // sparse cells per fixed column (the table column: `table_values`)
  std::vector<Fr> instance;                   // public inputs
  std::vector<std::pair<CellRef, CellRef>> copies;
  std::vector<uint64_t> table_values;
  size_t n_cells = 0, n_lookup_cells = 0;     // what `configure` counts
  explicit Assignment(const FlexGateCS& c) : cs(&c), advice(c.n_advice), fixed(c.n_fixed) {}
};

// halo2-base `Context` [layout restated from memory]: cells are appended in program order to ONE virtual column; Existing(cell)
// re-assigns the value and constrains it equal to the original; Constant(v) cells are tied to one fixed cell per distinct value
// afterwards; `finish` lays the virtual column out over the constraint system's gate columns (the same rules as the Python host's
// Context, halo2-scaffold_amd/flex.py: identical cells, copies and proof bytes).
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
  const Fr& value(uint32_t cell) const { return cells_[cell]; }
  uint32_t load_witness(const Fr& v) {
    cells_.push_back(v);
    return (uint32_t)cells_.size() - 1;
  }
  uint32_t assign_region_last(const std::vector<Item>& items, const std::vector<uint32_t>& gate_offsets) {
    const uint32_t base = (uint32_t)cells_.size();
    for (const Item& it : items) {
      const uint32_t row = (uint32_t)cells_.size();
      if (it.what == EXISTING) {
        cells_.push_back(cells_[it.cell]);
        eqs_.push_back({row, it.cell});
      } else {
        cells_.push_back(it.value);
        if (it.what == CONSTANT) const_cells_.push_back({row, it.value});
      }
    }
    for (uint32_t off : gate_offsets) gates_.push_back(base + off);
    return (uint32_t)cells_.size() - 1;
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
      const uint32_t base = (uint32_t)cells_.size() - 1;
      acc += limbs[i] << (lookup_bits * i);
      rows.push_back(load_witness(fr::from_u64(limbs[i])));
      const Fr pw = fr::from_u64(1ULL << (lookup_bits * i));
      const_cells_.push_back({load_witness(pw), pw});
      acc_row = load_witness(fr::from_u64(acc));
      gates_.push_back(base);
    }
    eqs_.push_back({a, acc_row});  // ctx.constrain_equal(&a, &acc)
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
    asg_.n_cells = cells_.size();
    asg_.n_lookup_cells = lookup_cells_.size();
    const uint32_t A = cs.num_advice, Lc = cs.num_lookup_advice;
    // where each cell of the virtual column lands: (advice column, row) of its first copy
    std::vector<std::pair<uint32_t, uint32_t>> where(cells_.size());
    if (A == 1) {
      asg_.advice[0] = cells_;
      for (uint32_t i = 0; i < cells_.size(); i++) where[i] = {0, i};
      for (uint32_t g : gates_) asg_.fixed[cs.col_qs[0]][g] = fr::ONE;
    } else {
      // assign_all over several gate columns [RECALL halo2-base 0.3 gates/builder.rs]: the cells run down the current column; a cell
      // that lands on the column's last row, or that starts a gate which no longer fits, is assigned a second time at row 0 of the
      // next column and tied to its first copy, and a gate starting there is enabled on the new column
      const uint32_t max_rows = ((uint32_t)1 << cs.k) - cs.minimum_rows;
      std::vector<bool> gate_at(cells_.size() + 1, false);
      for (uint32_t g : gates_) gate_at[g] = true;
      uint32_t col = 0, row = 0;
      for (uint32_t i = 0; i < cells_.size(); i++) {
        asg_.advice[col].push_back(cells_[i]);
        where[i] = {col, row};
        const bool q = gate_at[i];
        if ((q && row + 4 > max_rows) || row >= max_rows - 1) {
          if (col + 1 >= A) throw Error(H2MI_ERANGE, "NOT ENOUGH ADVICE COLUMNS");
          asg_.copies.push_back({{ADVICE, col + 1, 0}, {ADVICE, col, row}});
          col++;
          row = 0;
          asg_.advice[col].push_back(cells_[i]);
        }
        if (q) asg_.fixed[cs.col_qs[col]][row] = fr::ONE;
        row++;
      }
    }
    auto cell = [&](uint32_t i) { return CellRef{ADVICE, where[i].first, where[i].second}; };
    if (A > 1) {  // the cells to look up are copied into the lookup-advice columns
      const uint32_t max_rows = ((uint32_t)1 << cs.k) - cs.minimum_rows;
      uint32_t lcol = 0, lrow = 0;
      for (uint32_t i : lookup_cells_) {
        if (lrow >= max_rows) { lcol++; lrow = 0; }
        if (lcol >= Lc) throw Error(H2MI_ERANGE, "NOT ENOUGH LOOKUP ADVICE COLUMNS");
        asg_.advice[A + lcol].push_back(cells_[i]);
        asg_.copies.push_back({cell(i), {ADVICE, A + lcol, lrow}});
        lrow++;
      }
    }
    for (const auto& e : eqs_) asg_.copies.push_back({cell(e.first), cell(e.second)});
    std::vector<Fr> consts;  // one fixed cell per distinct value, in order of first use
    for (const auto& rc : const_cells_) {
      size_t idx = std::find(consts.begin(), consts.end(), rc.second) - consts.begin();
      const uint32_t ccol = (uint32_t)cs.col_const + (uint32_t)(idx % cs.num_fixed), crow = (uint32_t)(idx / cs.num_fixed);
      if (idx == consts.size()) {
        consts.push_back(rc.second);
        asg_.fixed[ccol][crow] = rc.second;
      }
      asg_.copies.push_back({cell(rc.first), {FIXED, ccol, crow}});
    }
    if (A == 1 && cs.lookup)
      for (uint32_t r : lookup_cells_) asg_.fixed[cs.col_qlookup][r] = fr::ONE;
    for (size_t i = 0; i < public_rows.size(); i++) {  // layouter.constrain_instance(cell, instance, i)
      asg_.instance.push_back(cells_[public_rows[i]]);
      asg_.copies.push_back({cell(public_rows[i]), {INSTANCE, 0, (uint32_t)i}});
    }
  }

 private:
  Assignment& asg_;
  std::vector<Fr> cells_;
  std::vector<uint32_t> gates_;                      // cells (virtual-column indices) with the gate enabled
  std::vector<std::pair<uint32_t, uint32_t>> eqs_;   // (new cell, source cell) equalities in call order
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
// RangeConfig::load_lookup_table (src/scaffold.rs:462): the Range builder assigns 0 .. 2^LOOKUP_BITS - 1 to the table column whatever the
// closure does — also for one that looks nothing up (the reference takes the Range builder whenever LOOKUP_BITS is set)
inline void load_lookup_table(Assignment& asg, uint32_t lookup_bits) {
  asg.table_values.resize((size_t)1 << lookup_bits);
  for (size_t i = 0; i < asg.table_values.size(); i++) asg.table_values[i] = i;
}
// reference examples/range.rs:10-34: make_public = [x]; range_check(x, 64); x + x.  Table: 0 .. 2^LOOKUP_BITS - 1
// count > 1: the same body for x, x + step, x + 2 step, ... (mod 2^64) in one context, every value public: fills several gate and
// lookup-advice columns while the limb bases, shared by all the checks, still fit the one constants column
inline Assignment range_closure(const FlexGateCS& cs, uint64_t x, uint32_t lookup_bits, uint32_t count = 1) {
  Assignment asg(cs);
  Context ctx(asg);
  std::vector<uint32_t> pub;
  for (uint32_t i = 0; i < count; i++) {
    const uint32_t xc = ctx.load_witness(fr::from_u64(x + (uint64_t)i * 0x9E3779B97F4A7C15ull));
    ctx.range_check(xc, 64, lookup_bits);
    ctx.add(xc, xc);
    pub.push_back(xc);
  }
  ctx.finish(pub);
  load_lookup_table(asg, lookup_bits);
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

// scaffold::mock (src/scaffold.rs:39-93: MockProver::run(k, &circuit, instances).assert_satisfied()) for these constraint systems, on
// the host: no cell beyond the usable rows (MockProver::run's NotEnoughRowsAvailable; the rule h2mi_prover_keygen applies), every
// enabled row satisfies its column's vertical gate, every copy constraint joins equal cells, every looked-up cell is a table
// value.  Throws Error(H2MI_EUNSAT / H2MI_ERANGE) naming the first violation — what the reference's users run before `prove`.
inline void mock(const Assignment& asg, uint32_t k) {
  const FlexGateCS& cs = *asg.cs;
  const uint64_t u = ((uint64_t)1 << k) - (cs.blinding_factors + 1);
  auto rows_ok = [&](const char* what, uint64_t last_row) {
    if (last_row >= u) throw Error(H2MI_ERANGE, std::string("mock: NotEnoughRowsAvailable: ") + what + " reaches beyond the usable rows");
  };
  for (const auto& col : asg.advice)
    if (!col.empty()) rows_ok("an advice column", col.size() - 1);
  for (const auto& col : asg.fixed)
    if (!col.empty()) rows_ok("a fixed column", col.rbegin()->first);
  if (cs.lookup && !asg.table_values.empty()) rows_ok("the lookup table", asg.table_values.size() - 1);
  auto adv = [&](uint32_t c, uint64_t r) { return r < asg.advice[c].size() ? asg.advice[c][r] : fr_zero(); };
  for (size_t j = 0; j < cs.col_qs.size(); j++)
    for (const auto& on : asg.fixed[cs.col_qs[j]]) {
      const uint64_t r = on.first;
      const Fr lhs = fr::add(adv((uint32_t)j, r), fr::mul(adv((uint32_t)j, r + 1), adv((uint32_t)j, r + 2)));
      if (!(fr::mul(on.second, fr::sub(lhs, adv((uint32_t)j, r + 3))) == fr_zero()))
        throw Error(H2MI_EUNSAT, "mock: gate not satisfied at row " + std::to_string(r) + " of gate column " + std::to_string(j));
    }
  auto value = [&](const CellRef& c) -> Fr {
    if (c.kind == ADVICE) return adv(c.col, c.row);
    if (c.kind == INSTANCE) return c.row < asg.instance.size() ? asg.instance[c.row] : fr_zero();
    const auto it = asg.fixed[c.col].find(c.row);
    return it == asg.fixed[c.col].end() ? fr_zero() : it->second;
  };
  for (const auto& cp : asg.copies) {
    rows_ok("a copy constraint", std::max(cp.first.row, cp.second.row));
    if (!(value(cp.first) == value(cp.second))) throw Error(H2MI_EUNSAT, "mock: a copy constraint joins unequal cells");
  }
  if (cs.lookup) {
    auto in_table = [&](const Fr& v) {
      const Fr c = plonk::to_canonical(v);
      return c.l[1] == 0 && c.l[2] == 0 && c.l[3] == 0 && (c.l[0] == 0 || std::find(asg.table_values.begin(), asg.table_values.end(), c.l[0]) != asg.table_values.end());
    };
    if (cs.num_advice == 1) {
      for (const auto& on : asg.fixed[cs.col_qlookup])
        if (!in_table(adv(0, on.first))) throw Error(H2MI_EUNSAT, "mock: lookup not satisfied at row " + std::to_string(on.first));
    } else {
      for (uint32_t l = 0; l < cs.num_lookup_advice; l++)
        for (const Fr& v : asg.advice[cs.num_advice + l])
          if (!in_table(v)) throw Error(H2MI_EUNSAT, "mock: lookup not satisfied in lookup-advice column " + std::to_string(l));
    }
  }
}

// GateThreadBuilder::config (src/scaffold.rs:268 `builder.config(k, Some(minimum_rows))`): run the closure once on the one-column
// constraint system to count its cells and cells to look up, and take ceil(count / (2^k - minimum_rows)) columns of each kind
template <class Closure>
inline FlexGateCS configure(bool lookup, uint32_t k, Closure closure, uint32_t minimum_rows = 9) {
  const FlexGateCS probe(lookup);
  const Assignment asg = closure(probe);
  const size_t max_rows = ((size_t)1 << k) - minimum_rows;
  const uint32_t num_advice = (uint32_t)std::max<size_t>(1, (asg.n_cells + max_rows - 1) / max_rows);
  if (num_advice == 1) return FlexGateCS(lookup, 1, 0, k, minimum_rows);
  const uint32_t num_lookup = lookup ? (uint32_t)((asg.n_lookup_cells + max_rows - 1) / max_rows) : 0;
  const size_t total_fixed = asg.fixed[probe.col_const].size();  // distinct constants: `(total_fixed + (1 << k) - 1) >> k` columns
  return FlexGateCS(lookup, num_advice, num_lookup, k, minimum_rows, (uint32_t)std::max<size_t>(1, (total_fixed + ((size_t)1 << k) - 1) >> k));
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
  std::vector<h2mi_column_cells> advice;  // every column dense from row 0
  for (const std::vector<Fr>& col : asg.advice) advice.push_back({nullptr, (const uint64_t*)col.data(), col.size(), 0});
  plonk::drive_proof(*workspace, advice, asg.instance, seed, tr);
}

}  // namespace flex
}  // namespace h2mi
