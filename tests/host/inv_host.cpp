// Host-side test shim for include/h2mi.hpp's single-element inversion (detail::inv_mod_odd): tests/test_host.py drives it
// against Python's pow(x, -1, p).  Test infrastructure, not product.
#include "../../halo2-scaffold_amd/csrc/h2mi_hostmath.hpp"
#include "../../halo2-scaffold_amd/csrc/inv_divsteps.cuh"

extern "C" {
// field 0 = Fq, 1 = Fr; plain integers in, plain integers out; ok[i] = 1 when the division steps converged
void h2t_inv_plain(int field, const uint64_t* in, uint64_t* out, uint8_t* ok, size_t n) {
  const uint64_t* mod = field ? h2mi::fr::MODULUS : h2mi::plonk::fq::MODULUS;
  for (size_t i = 0; i < n; i++) ok[i] = h2mi::detail::inv_mod_odd(in + 4 * i, mod, out + 4 * i) ? 1 : 0;
}
// the device's 32-bit form (csrc/inv_divsteps.cuh), compiled for the host: plain integers as 8 x 32-bit words
void h2t_inv_plain32(int field, const uint64_t* in, uint64_t* out, uint8_t* ok, size_t n) {
  const uint64_t* mod = field ? h2mi::fr::MODULUS : h2mi::plonk::fq::MODULUS;
  const uint64_t inv64 = field ? h2mi::fr::INV : h2mi::plonk::fq::INV;  // -p^-1 mod 2^64: its low word is -p^-1 mod 2^32
  for (size_t i = 0; i < n; i++)
    ok[i] = h2::inv_divsteps_256((const uint32_t*)(in + 4 * i), (const uint32_t*)mod, (uint32_t)inv64, (uint32_t*)(out + 4 * i)) ? 1 : 0;
}
// Montgomery forms in and out, through the public helpers; fermat != 0 takes the exponentiation (the definition)
void h2t_inv_mont(int field, int fermat, const uint64_t* in, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    if (field) {
      h2mi::Fr a;
      std::memcpy(a.l, in + 4 * i, 32);
      const h2mi::Fr r = fermat ? h2mi::fr::invert_fermat(a) : h2mi::fr::invert(a);
      std::memcpy(out + 4 * i, r.l, 32);
    } else {
      h2mi::plonk::fq::E a;
      std::memcpy(a.l, in + 4 * i, 32);
      const h2mi::plonk::fq::E r = fermat ? h2mi::plonk::fq::invert_fermat(a) : h2mi::plonk::fq::invert(a);
      std::memcpy(out + 4 * i, r.l, 32);
    }
  }
}
}
