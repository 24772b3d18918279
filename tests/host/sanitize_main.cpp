// AddressSanitizer / UndefinedBehaviorSanitizer run of the host-compilable layers (GPU sanitizers are not available on the pool: the
// CPU build is where they run): the 29-bit-limb field and curve code of csrc/f29.cuh / g1_29.cuh (the same headers the kernels
// compile), the division-step inversion, and the prover's host helpers (csrc/h2mi_hostmath.hpp: permutation assembly, batch
// normalisation, the blinding streams), driven through the test shims with seeded pseudo-random operands.  Results are not judged
// here (tests/test_f29_host.py and tests/test_host.py compare them with the oracle): the run only has to finish without a report.
// Built and run by tests/test_host.py::test_host_layers_clean_under_sanitizers.  Test infrastructure, not product.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "f29_host.cpp"
#include "inv_host.cpp"
#include "prover_host.cpp"

static uint64_t state = 0x9E3779B97F4A7C15ull;
static uint64_t next() {
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int main() {
  const size_t n = 256;
  // field layer: canonical 256-bit words below 2^253 (any value below the moduli), both fields, every multiplication mode
  std::vector<uint32_t> a(8 * n), b(8 * n), out(9 * n + 64);
  for (auto* v : {&a, &b})
    for (size_t i = 0; i < n; i++) {
      for (int w = 0; w < 8; w++) (*v)[8 * i + w] = (uint32_t)next();
      (*v)[8 * i + 7] &= 0x1FFFFFFFu;
    }
  for (int field = 0; field < 2; field++)
    for (int mode = 0; mode < 3; mode++) f29t_mul(field, mode, a.data(), b.data(), out.data(), n);
  // curve layer: a chain and a tree of mixed additions over multiples of the generator, with signs, doubling and cancellation cases
  {
    // the generator (1, 2) in Montgomery form: x = R mod q, y = 2 R mod q (one conditional subtraction)
    using h2mi::plonk::fq::MODULUS;
    uint64_t g[8];
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
      g[i] = h2mi::plonk::fq::ONE.l[i];
      c += (unsigned __int128)g[i] * 2;
      g[4 + i] = (uint64_t)c;
      c >>= 64;
    }
    bool ge = true;
    for (int i = 3; i >= 0; i--)
      if (g[4 + i] != MODULUS[i]) { ge = g[4 + i] > MODULUS[i]; break; }
    if (ge) {
      unsigned __int128 br = 0;
      for (int i = 0; i < 4; i++) {
        const unsigned __int128 d = (unsigned __int128)g[4 + i] - MODULUS[i] - (uint64_t)br;
        g[4 + i] = (uint64_t)d;
        br = (d >> 64) & 1;
      }
    }
    {  // and through the host normaliser (Jacobian (x, y, 1) -> affine), which is part of what this run covers
      uint64_t jac[12], aff[8];
      for (int i = 0; i < 8; i++) jac[i] = g[i];
      for (int i = 0; i < 4; i++) jac[8 + i] = h2mi::plonk::fq::ONE.l[i];
      h2t_normalize(jac, 1, aff);
      for (int i = 0; i < 8; i++)
        if (aff[i] != g[i]) { std::printf("normalize changed an affine point\n"); return 1; }
    }
    std::vector<uint32_t> pts(16 * n);
    std::vector<uint8_t> signs(n);
    for (size_t i = 0; i < n; i++) {
      for (int w = 0; w < 8; w++) {  // the same point every time: additions, doublings (equal signs) and cancellations (opposite)
        pts[16 * i + 2 * w] = (uint32_t)g[w];
        pts[16 * i + 2 * w + 1] = (uint32_t)(g[w] >> 32);
      }
      signs[i] = (uint8_t)(next() & 1);
    }
    std::vector<uint32_t> xyzz(36);
    for (int tree = 0; tree < 2; tree++) f29t_madd_chain(pts.data(), signs.data(), n, xyzz.data(), tree);
  }
  // inversion by division steps and by Fermat, both fields; zero included
  {
    std::vector<uint64_t> in(4 * n), o(4 * n);
    std::vector<uint8_t> ok(n);
    for (size_t i = 0; i < n; i++) {
      for (int w = 0; w < 4; w++) in[4 * i + w] = i ? next() : 0;
      in[4 * i + 3] &= 0x1FFFFFFFFFFFFFFFull;
    }
    for (int field = 0; field < 2; field++) {
      h2t_inv_plain(field, in.data(), o.data(), ok.data(), n);
      h2t_inv_plain32(field, in.data(), o.data(), ok.data(), n);
      for (int fermat = 0; fermat < 2; fermat++) h2t_inv_mont(field, fermat, in.data(), o.data(), n);
    }
  }
  // prover helpers: assembly over random copies (self-copies, repeated pairs, long cycles), blinding streams, canonical order
  {
    const size_t m = 2000;
    std::vector<uint32_t> copies(4 * m), mapping(8 * m + 8);
    for (size_t i = 0; i < m; i++) {
      copies[4 * i] = (uint32_t)(next() % 5);
      copies[4 * i + 1] = (uint32_t)(next() % 97);
      copies[4 * i + 2] = (uint32_t)(next() % 5);
      copies[4 * i + 3] = (uint32_t)(next() % 97);
    }
    h2t_assembly(copies.data(), m, mapping.data());
    std::vector<uint64_t> fr(4 * 300);
    h2t_uniform_fr(7, 300, 5, fr.data());
    uint8_t key[32];
    for (auto& kb : key) kb = (uint8_t)next();
    h2t_chacha_fr(key, 42, 300, 1, fr.data());
    int less = 0;
    for (size_t i = 0; i + 1 < 300; i++) less += h2t_canonical_less(&fr[4 * i], &fr[4 * i + 4]);
    std::printf("sanitize_main: done (%d)\n", less);
  }
  return 0;
}
