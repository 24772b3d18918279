// C++ counterpart of the reference's halo2-lib examples proved through scaffold::prove (src/scaffold.rs:246-366) over
// include/h2mi_flex.hpp: examples/halo2_lib.rs (x^2 + 72 through the Gate builder) and examples/range.rs
// (range_check(x, 64) through the Range builder with a LOOKUP_BITS table) — keygen, then create_proof with every vector in
// HBM; public inputs and proof bytes printed.  The proof bytes are checked by the test-suite against the oracle engine
// and the Python host (tests/test_gpu_flex.py).
//
// Usage: [DEGREE=k] [LOOKUP_BITS=b] [MINIMUM_ROWS=r] halo2_lib <halo2_lib | range | poseidon> [k [lookup_bits [x [srs_secret_hex [seed [count [num_advice num_lookup_advice [num_fixed]]]]]]]]
//        (poseidon hashes x and x + 1: examples/poseidon.rs `hash_two`; count: range checks in one context, see flex::range_closure;
//        num_advice / num_lookup_advice: the column counts set by hand instead of taken from builder.config)
//        (the reference reads DEGREE and LOOKUP_BITS from the environment and draws x and the rng from OsRng)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>

#include "../include/h2mi_flex.hpp"

using namespace h2mi;
using Clock = std::chrono::steady_clock;

struct Timer {
  const char* name;
  Clock::time_point t0;
  explicit Timer(const char* n) : name(n), t0(Clock::now()) { std::printf("Start:   %s\n", name); }
  ~Timer() { std::printf("End:     %s ...%.3fms\n", name, std::chrono::duration<double, std::milli>(Clock::now() - t0).count()); }
};
static Fr fr_from_hex(std::string h) {
  if (h.rfind("0x", 0) == 0) h = h.substr(2);
  while (h.size() < 64) h = "0" + h;
  if (h.size() > 64) h = h.substr(h.size() - 64);
  Fr raw;
  for (int i = 0; i < 4; i++) raw.l[i] = std::stoull(h.substr(64 - 16 * (i + 1), 16), nullptr, 16);
  return fr::mul(raw, fr::R2);
}
static std::string hex(const std::vector<uint8_t>& b) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (uint8_t c : b) {
    s.push_back(d[c >> 4]);
    s.push_back(d[c & 15]);
  }
  return s;
}

int main(int argc, char** argv) {
  const std::string shape = argc > 1 ? argv[1] : "halo2_lib";
  // DEGREE, LOOKUP_BITS and MINIMUM_ROWS are read from the environment as the reference does (README: `DEGREE=<k> LOOKUP_BITS=8 cargo run
  // --example range`; src/scaffold.rs:44-50); the arguments override them
  const uint32_t k = argc > 2 ? (uint32_t)std::atoi(argv[2]) : std::getenv("DEGREE") ? (uint32_t)std::atoi(std::getenv("DEGREE")) : 10;
  const uint32_t minimum_rows = std::getenv("MINIMUM_ROWS") ? (uint32_t)std::atoi(std::getenv("MINIMUM_ROWS")) : 9;
  const uint32_t lookup_bits = std::getenv("LOOKUP_BITS") ? (uint32_t)std::atoi(std::getenv("LOOKUP_BITS")) : argc > 3 ? (uint32_t)std::atoi(argv[3]) : 8;
  const uint64_t x = argc > 4 ? std::stoull(argv[4], nullptr, 0) : 12;
  const Fr s = fr_from_hex(argc > 5 ? argv[5] : "5ec2e7");
  const uint64_t seed = argc > 6 ? std::stoull(argv[6]) : 11;
  const uint32_t count = argc > 7 ? (uint32_t)std::max(1, std::atoi(argv[7])) : 1;
  const uint32_t set_advice = argc > 9 ? (uint32_t)std::atoi(argv[8]) : 0, set_lookup = argc > 9 ? (uint32_t)std::atoi(argv[9]) : 0;
  const uint32_t set_fixed = argc > 10 ? (uint32_t)std::max(1, std::atoi(argv[10])) : 1;
  // the reference takes the Range builder whenever LOOKUP_BITS is set, whatever the closure (src/scaffold.rs:44-48): so does this
  const char* env_bits = std::getenv("LOOKUP_BITS");
  const bool lookup = shape == "range" || env_bits != nullptr;
  if (shape != "range" && shape != "halo2_lib" && shape != "poseidon") {
    std::fprintf(stderr, "usage: halo2_lib <halo2_lib | range | poseidon> [k [lookup_bits [x [srs_secret_hex [seed]]]]]\n");
    return 1;
  }
  try {
    init();
    auto params = [&] { Timer t("Generating params"); return poly::kzg::ParamsKZG::setup(k, s); }();
    auto run = [&](const flex::FlexGateCS& c, uint64_t v) {
      if (shape == "range") return flex::range_closure(c, v, lookup_bits, count);
      flex::Assignment a = shape == "poseidon" ? flex::poseidon_hash_two_closure(c, fr::from_u64(v), fr::from_u64(v + 1)) : flex::halo2_lib_closure(c, fr::from_u64(v));
      if (lookup) flex::load_lookup_table(a, lookup_bits);  // the Range builder loads its table whatever the closure looks up
      return a;
    };
    // `builder.config(k, Some(minimum_rows))` (src/scaffold.rs:268): more than one gate column when the closure's cells overflow 2^k rows
    const flex::FlexGateCS cs = set_advice > 1 ? flex::FlexGateCS(lookup, set_advice, set_lookup, k, minimum_rows, set_fixed)
                                               : flex::configure(lookup, k, [&](const flex::FlexGateCS& c) { return run(c, x); }, minimum_rows);
    if (cs.num_advice > 1) std::printf("columns %u gate + %u lookup-advice\n", cs.num_advice, cs.num_lookup_advice);
    auto closure = [&](uint64_t v) { return run(cs, v); };
    // keygen: the reference runs the closure once on dummy inputs to fix the circuit's shape
    auto pk = [&] { Timer t("Generating verifying and proving key"); return flex::keygen(params, cs, closure(0)); }();
    flex::Assignment asg = closure(x);
    flex::mock(asg, k);  // scaffold::mock (the reference's examples run it first: MockProver::run(..).assert_satisfied())
    std::vector<uint8_t> proof;
    flex::FlexWorkspace ws(params, *pk);
    for (int run = 0; run < 3; run++) {
      Timer t(run ? "Creating proof" : "Creating proof (first call: builds the domain's tables)");
      auto transcript = transcript::Blake2bWrite::init();
      flex::create_proof(params, *pk, asg, seed, transcript, &ws);
      proof = transcript.finalize();
    }
    if (const char* np = std::getenv("H2MI_PROOFS")) {  // steady state: N more proofs through the same workspace (bench.py reads the line)
      const int count = std::max(1, std::atoi(np));  // 0 or a non-number: one proof, never a division by zero
      check(h2mi_sync(), "sync");
      ws.time_phases = std::getenv("H2MI_PHASES") != nullptr;  // untraced host-side phase clock (ProverWorkspace::phase_us)
      const auto t0 = Clock::now();
      for (int i = 0; i < count; i++) {
        auto transcript = transcript::Blake2bWrite::init();
        flex::create_proof(params, *pk, asg, seed + 1 + (uint64_t)i, transcript, &ws);
        transcript.finalize();
      }
      std::printf("steady_ms_per_proof %.4f over %d proofs\n", std::chrono::duration<double, std::milli>(Clock::now() - t0).count() / count, count);
      if (ws.time_phases)
        std::printf("phase_us advice %.1f lookups %.1f products %.1f quotient %.1f evaluations %.1f shplonk_1 %.1f shplonk_2 %.1f\n", ws.phase_us[0] / count,
                    ws.phase_us[1] / count, ws.phase_us[2] / count, ws.phase_us[3] / count, ws.phase_us[4] / count, ws.phase_us[5] / count, ws.phase_us[6] / count);
    }
    std::printf("vk %s\n", hex(pk->vk.to_bytes()).c_str());
    for (const Fr& v : asg.instance) {
      const Fr c = plonk::to_canonical(v);
      std::printf("instance %016llx%016llx%016llx%016llx\n", (unsigned long long)c.l[3], (unsigned long long)c.l[2], (unsigned long long)c.l[1],
                  (unsigned long long)c.l[0]);
    }
    std::printf("proof %s\n", hex(proof).c_str());
    std::printf("proof_bytes %zu\n", proof.size());
    h2mi_shutdown();
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "halo2_lib: %s\n", e.what());
    return 2;
  }
}
