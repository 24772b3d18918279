// Host-side test shim for the prover's host helpers (csrc/h2mi_hostmath.hpp: what csrc/h2mi_prover.cpp runs on the CPU around its
// kernels): permutation/keygen.rs Assembly, G1::batch_normalize of a phase's points, the counter-based blinding stream.
// tests/test_host.py drives it against the oracle without a GPU.  Test infrastructure, not product.
#include <algorithm>
#include <array>

#include "../../halo2-scaffold_amd/csrc/h2mi_hostmath.hpp"

extern "C" {
// copies: n x 4 (left column, left row, right column, right row) in call order; out: the non-identity entries of the mapping as
// (column, row, column', row') sorted by (column, row); returns their number (out may hold up to 2 n entries)
size_t h2t_assembly(const uint32_t* copies, size_t n, uint32_t* out) {
  h2mi::plonk::PermutationAssembly a;
  for (size_t i = 0; i < n; i++)
    a.copy(h2mi::plonk::Cell(copies[4 * i], copies[4 * i + 1]), h2mi::plonk::Cell(copies[4 * i + 2], copies[4 * i + 3]));
  std::vector<std::array<uint32_t, 4>> rows;
  a.for_each([&](const h2mi::plonk::Cell& from, const h2mi::plonk::Cell& to) {
    if (from != to) rows.push_back({from.first, from.second, to.first, to.second});
  });
  std::sort(rows.begin(), rows.end());
  for (size_t i = 0; i < rows.size(); i++) std::copy(rows[i].begin(), rows[i].end(), out + 4 * i);
  return rows.size();
}
void h2t_uniform_fr(uint64_t seed, size_t count, uint64_t start, uint64_t* out) {
  const std::vector<h2mi::Fr> v = h2mi::plonk::uniform_fr(seed, count, start);
  std::memcpy(out, v.data(), count * 32);
}
// k Jacobian points (12 limbs each) -> k affine points (8 limbs each), identity -> (0, 0)
void h2t_normalize(const uint64_t* jac, size_t k, uint64_t* out) {
  std::vector<h2mi::G1> pts(k);
  std::memcpy(pts.data(), jac, k * 96);
  const std::vector<h2mi::G1Affine> aff = h2mi::plonk::normalize_host_batch(pts);
  std::memcpy(out, aff.data(), k * 64);
}
// the keyed blinding stream (ChaCha20 block per scalar, Fr::from_u512): what h2mi_prover_set_rng_key draws from on the host side
void h2t_chacha_fr(const uint8_t* key, uint64_t stream, size_t count, uint64_t start, uint64_t* out) {
  const std::vector<h2mi::Fr> v = h2mi::plonk::chacha_fr(key, stream, count, start);
  std::memcpy(out, v.data(), count * 32);
}
// canonical order of Fr (BTreeSet<Fr> in ProverSHPLONK): 1 when a < b
int h2t_canonical_less(const uint64_t* a, const uint64_t* b) {
  h2mi::Fr x, y;
  std::memcpy(x.l, a, 32);
  std::memcpy(y.l, b, 32);
  return h2mi::plonk::canonical_less(x, y) ? 1 : 0;
}
}
