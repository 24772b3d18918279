// AddressSanitizer / UndefinedBehaviorSanitizer run of the C++ host's circuit-building code (include/h2mi_flex.hpp: Context, the
// break-point layout over several gate / lookup-advice / constants columns, configure; include/h2mi_plonk.hpp: StandardPlonk's
// synthesize and its constraint system): everything a caller runs on the CPU before the first call into the library.  No GPU call
// is made (the library is linked because the headers reference it).  Built and run by
// tests/test_host.py::test_cpp_host_layout_clean_under_sanitizers.  Test infrastructure, not product.
#include <cstdio>

#include "../../include/h2mi_flex.hpp"

using namespace h2mi;

int main() {
  size_t laid_out = 0, refused = 0, cells = 0, mocked = 0, rows_refused = 0;
  for (uint32_t k = 4; k <= 9; k++)
    for (uint32_t bits = 1; bits <= 8 && bits < k; bits++)
      for (uint32_t count : {1u, 2u, 5u, 12u}) {
        const uint64_t x = 0xDEADBEEFCAFE1234ull * (k + 3 * bits + count);
        auto closure = [&](const flex::FlexGateCS& c) { return flex::range_closure(c, x, bits, count); };
        try {
          const flex::FlexGateCS cs = flex::configure(true, k, closure);
          const flex::Assignment asg = closure(cs);
          try {
            flex::mock(asg, k);
            mocked++;
          } catch (const Error& e) {  // only the row rule may refuse a layout the closure produced itself
            if (e.code != H2MI_ERANGE) throw;
            rows_refused++;
          }
          const h2mi_constraint_system abi = cs.abi(k);
          cells += asg.n_cells + abi.n_perm + abi.n_advice_queries;
          laid_out++;
        } catch (const Error&) {  // NOT ENOUGH ADVICE COLUMNS, or more columns than the prover ABI holds: as in halo2-base
          refused++;
        }
      }
  for (uint32_t k = 4; k <= 12; k++) {
    auto hl = [&](const flex::FlexGateCS& c) { return flex::halo2_lib_closure(c, fr::from_u64(k + 7)); };
    auto ps = [&](const flex::FlexGateCS& c) { return flex::poseidon_hash_two_closure(c, fr::from_u64(k), fr::from_u64(k + 1)); };
    try {
      const flex::FlexGateCS cs = flex::configure(false, k, hl);
      cells += hl(cs).n_cells;
      laid_out++;
    } catch (const Error&) {
      refused++;
    }
    try {
      const flex::FlexGateCS cs = flex::configure(false, k, ps);
      cells += ps(cs).n_cells;
      laid_out++;
    } catch (const Error&) {
      refused++;
    }
  }
  {  // a broken witness is caught: one cell changed under an enabled gate
    const flex::FlexGateCS cs(false);
    flex::Assignment asg = flex::halo2_lib_closure(cs, fr::from_u64(12));
    flex::mock(asg, 6);
    asg.advice[0][3] = fr::add(asg.advice[0][3], fr::ONE);
    bool caught = false;
    try {
      flex::mock(asg, 6);
    } catch (const Error& e) {
      caught = e.code == H2MI_EUNSAT;
    }
    if (!caught) return 2;
  }
  {  // explicit column counts, two constants columns
    const flex::FlexGateCS cs(true, 5, 2, 5, 9, 2);
    cells += flex::range_closure(cs, 0xDEADBEEFCAFE1234ull, 2).n_cells;
    laid_out++;
  }
  {
    plonk::StandardPlonk circuit;
    circuit.x = fr::from_u64(0xC0FFEE);
    const plonk::Synthesis syn = circuit.synthesize();
    const h2mi_constraint_system abi = plonk::StandardPlonk::constraint_system(5);
    cells += syn.copies.size() + abi.n_perm;
  }
  std::printf("sanitize_flex: done (%zu layouts, %zu refused, %zu mocked, %zu beyond the usable rows, %zu)\n", laid_out, refused, mocked, rows_refused, cells);
  return laid_out > 40 && mocked > 40 ? 0 : 1;
}
