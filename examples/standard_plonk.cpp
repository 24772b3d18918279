// C++ counterpart of the reference's examples/standard_plonk.rs (lines 25-65) over include/h2mi.hpp + h2mi_plonk.hpp — a thin
// caller of the library's prover (h2mi_prover.h: keygen and seven phase calls per proof; the transcript stays here):
// the same flow with the same names —
//     ParamsKZG::setup(k, ..); keygen_vk; keygen_pk; "Creating proof": Blake2bWrite::init, create_proof, finalize
// — on the reference's StandardPlonk circuit (src/circuits/standard_plonk.rs), data-true: real witness and copy
// constraints, challenges from the Blake2b transcript, 992 proof bytes.  Spans are timed like the reference's ark-std
// timers.  The reference then calls verify_proof; no verifier is part of the product (it is not on the hot path): the
// proof bytes printed here are checked by the test-suite against the oracle prover / verifier and the golden proofs.
//
// Usage: standard_plonk [k [srs_secret_hex [witness_hex [seed]]]]     (the reference hard-codes k = 5 and draws the rest
//        from OsRng; here they are arguments so that runs are comparable)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>

#include "../include/h2mi_plonk.hpp"

using namespace h2mi;
using Clock = std::chrono::steady_clock;

struct Timer {
  const char* name;
  Clock::time_point t0;
  explicit Timer(const char* n) : name(n), t0(Clock::now()) { std::printf("Start:   %s\n", name); }
  ~Timer() {
    double ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
    std::printf("End:     %s ...%.3fms\n", name, ms);
  }
};

static Fr fr_from_hex(std::string h) {  // canonical integer (< 2^256, reduced mod r by the Montgomery conversions) -> Fr
  if (h.rfind("0x", 0) == 0) h = h.substr(2);
  while (h.size() < 64) h = "0" + h;
  if (h.size() > 64) h = h.substr(h.size() - 64);
  Fr raw;
  for (int i = 0; i < 4; i++) raw.l[i] = std::stoull(h.substr(64 - 16 * (i + 1), 16), nullptr, 16);
  // raw may exceed r: (raw R^-1) * R^2 * ... two Montgomery products bring it to raw * R mod r
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
  const uint32_t k = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 5;  // `let k = 5;`
  const Fr s = fr_from_hex(argc > 2 ? argv[2] : "5ec2e7");
  const Fr x = fr_from_hex(argc > 3 ? argv[3] : "c0ffee");
  const uint64_t seed = argc > 4 ? std::stoull(argv[4]) : 11;
  try {
    init();
    // let params = ParamsKZG::<Bn256>::setup(k, OsRng);
    auto params = [&] { Timer t("Generating params"); return poly::kzg::ParamsKZG::setup(k, s); }();
    // let circuit = StandardPlonk { x: Value::unknown() }; keygen_vk; keygen_pk
    plonk::StandardPlonk keygen_circuit;
    plonk::VerifyingKey vk = [&] { Timer t("Generating verifying key"); return plonk::keygen_vk(params, keygen_circuit); }();
    auto pk = [&] { Timer t("Generating proving key"); return plonk::keygen_pk(params, vk, keygen_circuit); }();
    // let circuit = StandardPlonk { x: Value::known(..) };
    plonk::StandardPlonk circuit(x);
    std::vector<uint8_t> proof;
    plonk::ProverWorkspace ws(params, *pk);  // the prover's device buffers, kept across proofs
    for (int run = 0; run < 3; run++) {  // later runs show the steady state (tables and plans are cached)
      Timer t(run ? "Creating proof" : "Creating proof (first call: builds the domain's tables)");
      auto transcript = transcript::Blake2bWrite::init();
      plonk::create_proof(params, *pk, circuit, seed, transcript, &ws);
      proof = transcript.finalize();
    }
    if (const char* np = std::getenv("H2MI_PROOFS")) {  // steady state: N more proofs through the same workspace (bench.py reads the line)
      const int count = std::max(1, std::atoi(np));  // 0 or a non-number: one proof, never a division by zero
      check(h2mi_sync(), "sync");
      ws.time_phases = std::getenv("H2MI_PHASES") != nullptr;  // untraced host-side phase clock (ProverWorkspace::phase_us)
      const auto t0 = Clock::now();
      for (int i = 0; i < count; i++) {
        auto transcript = transcript::Blake2bWrite::init();
        plonk::create_proof(params, *pk, circuit, seed + 1 + (uint64_t)i, transcript, &ws);
        transcript.finalize();
      }
      std::printf("steady_ms_per_proof %.4f over %d proofs\n", std::chrono::duration<double, std::milli>(Clock::now() - t0).count() / count, count);
      if (ws.time_phases)
        std::printf("phase_us advice %.1f z_random %.1f h_pieces %.1f evaluations %.1f shplonk_1 %.1f shplonk_2 %.1f\n", ws.phase_us[0] / count,
                    (ws.phase_us[1] + ws.phase_us[2]) / count, ws.phase_us[3] / count, ws.phase_us[4] / count, ws.phase_us[5] / count, ws.phase_us[6] / count);
    }
    std::printf("vk %s\n", hex(vk.to_bytes()).c_str());
    std::printf("proof %s\n", hex(proof).c_str());
    std::printf("proof_bytes %zu\n", proof.size());
    h2mi_shutdown();
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "standard_plonk: %s\n", e.what());
    return 2;
  }
}
