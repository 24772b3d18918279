// C++ counterpart of the reference's examples/standard_plonk.rs (lines 25-65) over include/h2mi.hpp:
// same flow — setup the SRS for 2^k rows, "keygen" (commit the fixed / permutation columns and bring them
// to coefficient and extended form), then "Creating proof": the commitments and transforms create_proof
// issues for the StandardPlonk circuit (reference src/circuits/standard_plonk.rs:29-112), timed like the
// reference's ark-std spans.  The advice columns are the circuit's real witness (x, x, x^2 / x, x, x^2+72 in
// rows 1-2, zeros elsewhere, random blinding rows at the end); permutation products, the random polynomial
// and h(X) are synthetic dense vectors, because gate evaluation and the transcript are out of scope
// (DESIGN.md).  It is NOT a prover: instead of verify_proof it self-checks the KZG identity
// commit(f; g) == commit_lagrange(NTT f; g_lagrange) and f(s)*G on every run.
//
// Usage: standard_plonk [k]      (the reference hard-codes k = 5)
#include <chrono>
#include <cstdio>
#include <random>

#include "../include/h2mi.hpp"

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

static Fr random_fr(std::mt19937_64& rng) {  // uniform Montgomery representative = uniform field element
  for (;;) {
    Fr a = {{rng(), rng(), rng(), rng() & 0x3fffffffffffffffULL}};
    for (int i = 3; i >= 0; i--) {
      if (a.l[i] < fr::MODULUS[i]) return a;
      if (a.l[i] > fr::MODULUS[i]) break;
    }
  }
}
static std::vector<Fr> random_vec(size_t n, std::mt19937_64& rng) {
  std::vector<Fr> v(n);
  for (auto& x : v) x = random_fr(rng);
  return v;
}

int main(int argc, char** argv) {
  const uint32_t k = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 5;  // `let k = 5;`
  const size_t n = (size_t)1 << k;
  const uint32_t cs_degree = 3, blinding_rows = 5;
  try {
    init();
    std::mt19937_64 rng(0x48324D49);  // the reference uses OsRng everywhere; a seed makes runs comparable
    const Fr zero = {{0, 0, 0, 0}};

    // let params = ParamsKZG::<Bn256>::setup(k, OsRng);
    const Fr s = random_fr(rng);
    auto params = [&] { Timer t("Generating params"); return poly::kzg::ParamsKZG::setup(k, s); }();
    poly::EvaluationDomain domain(cs_degree, k);

    // keygen_vk / keygen_pk: 5 fixed columns (q_a, q_b, q_c, q_ab, constant) + 3 permutation sigmas
    std::vector<std::vector<Fr>> fixed(5, std::vector<Fr>(n, zero));
    const Fr one = fr::ONE, minus_one = fr::neg(fr::ONE);
    if (n > 2) {
      fixed[2][1] = minus_one; fixed[3][1] = one;                                   // row 1: q_c = -1, q_ab = 1
      fixed[2][2] = minus_one; fixed[3][2] = one; fixed[4][2] = fr::from_u64(72);  // row 2: + constant 72
    }
    std::vector<std::vector<Fr>> sigma(3);
    for (auto& sg : sigma) sg = random_vec(n, rng);
    std::vector<G1> vk_commitments;
    {
      Timer t("Generating verifying key");
      for (auto& c : fixed) vk_commitments.push_back(params.commit_lagrange(c));
      for (auto& c : sigma) vk_commitments.push_back(params.commit_lagrange(c));
    }
    {
      Timer t("Generating proving key");
      for (auto& c : fixed) (void)domain.coeff_to_extended(domain.lagrange_to_coeff(c));
      for (auto& c : sigma) (void)domain.coeff_to_extended(domain.lagrange_to_coeff(c));
    }

    // the witness of StandardPlonk { x }: a, b, c columns
    const Fr x = random_fr(rng);
    std::vector<std::vector<Fr>> advice(3, std::vector<Fr>(n, zero));
    if (n > 2) {
      const Fr xx = fr::mul(x, x);
      const Fr xx72 = fr::add(xx, fr::from_u64(72));
      advice[0][0] = x;
      advice[0][1] = x; advice[1][1] = x; advice[2][1] = xx;
      advice[0][2] = x; advice[1][2] = x; advice[2][2] = xx72;
    }
    for (auto& col : advice)  // blinding factors in the last rows (create_proof step 2)
      for (size_t r = n > blinding_rows ? n - blinding_rows : 0; r < n; r++) col[r] = random_fr(rng);

    std::vector<G1> proof_points;
    {
      Timer t("Creating proof");
      // advice commitments
      for (auto& col : advice) proof_points.push_back(params.commit_lagrange(col));
      // permutation products: commit, to coefficients, to the extended domain
      std::vector<std::vector<Fr>> zs(3);
      for (auto& z : zs) {
        z = random_vec(n, rng);
        proof_points.push_back(params.commit_lagrange(z));
        (void)domain.coeff_to_extended(domain.lagrange_to_coeff(z));
      }
      // vanishing argument: random polynomial
      proof_points.push_back(params.commit(random_vec(n, rng)));
      // advice to coefficient / extended form
      std::vector<std::vector<Fr>> advice_coeff;
      for (auto& col : advice) {
        advice_coeff.push_back(domain.lagrange_to_coeff(col));
        (void)domain.coeff_to_extended(advice_coeff.back());
      }
      // h(X): extended -> coefficients, (degree - 1) pieces
      std::vector<Fr> h = domain.extended_to_coeff(random_vec(domain.extended_len(), rng));
      for (uint32_t piece = 0; piece < cs_degree - 1; piece++)
        proof_points.push_back(params.commit(std::vector<Fr>(h.begin() + piece * n, h.begin() + (piece + 1) * n)));
      // evaluations at the challenge x and the two SHPLONK commitments
      const Fr xc = random_fr(rng);
      std::vector<Fr> evals;
      for (auto& c : advice_coeff) evals.push_back(arithmetic::eval_polynomial(c, xc));
      proof_points.push_back(params.commit(advice_coeff[0]));
      std::vector<Fr> q = arithmetic::kate_division(advice_coeff[0], xc);
      proof_points.push_back(params.commit(q));
      // (X - xc) q(X) + f(xc) == f(X): spot-check at a second point
      const Fr zc = random_fr(rng);
      Fr lhs = fr::add(fr::mul(fr::sub(zc, xc), arithmetic::eval_polynomial(q, zc)), evals[0]);
      if (!(lhs == arithmetic::eval_polynomial(advice_coeff[0], zc))) {
        std::printf("kate_division self-check FAILED\n");
        return 1;
      }
    }
    std::printf("proof replay: %zu commitments\n", proof_points.size());

    // self-check instead of verify_proof: commit(f; g) == commit_lagrange(NTT f; g_lagrange) == f(s) G
    {
      Timer t("verify");
      std::vector<Fr> f = random_vec(n, rng);
      G1 c1 = params.commit(f), c2 = params.commit_lagrange(domain.coeff_to_lagrange(f));
      Fr fs = arithmetic::eval_polynomial(f, s);
      std::vector<Fr> one_scalar = {fs};
      std::vector<G1Affine> g0 = {params.get_g()[0]};
      G1 c3 = arithmetic::best_multiexp(one_scalar, g0);
      std::vector<G1Affine> a = batch_normalize({c1, c2, c3});
      bool ok = std::memcmp(&a[0], &a[1], 64) == 0 && std::memcmp(&a[0], &a[2], 64) == 0 && !c1.is_identity();
      if (!ok) {
        std::printf("self-check FAILED\n");
        return 1;
      }
    }
    std::printf("MSM / NTT self-check passed (commit == commit_lagrange == f(s)*G) at k = %u\n", k);
    h2mi_shutdown();
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "h2mi error %d: %s\n", e.code, e.what());
    return 2;
  }
}
