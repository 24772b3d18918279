// Internal interface between the scalar-field translation units of libh2mi.so (h2mi_ntt.hip: transforms, power-table cache,
// scratch vectors; h2mi_poly.hip: the opening argument's polynomial helpers; h2mi_plonk.hip: grand products and quotient kernels).
// Not part of the ABI.
#pragma once
#include "f29.cuh"
#include "fp.cuh"
#include "h2mi_internal.h"

namespace h2 {

using Fr = FrP;
using F9 = Fr29;

// ---- device: 256-bit words <-> the lazy 29-bit-limb layer, and base^e from a split power table (Montgomery-2^261 entries) ----------
__device__ __forceinline__ f29 load_unpack(const fe* p) {
  fe x = fe_load(p);
  return f29_unpack(x.v);
}
__device__ __forceinline__ void pack_store(fe* p, const f29& a_lt2p) {
  fe o;
  f29_pack(f29_reduce_canonical<F9>(a_lt2p), o.v);
  fe_store(p, o);
}
__device__ __forceinline__ f29 pow2tab(const fe* lo, const fe* hi, uint32_t h, uint32_t e) {  // Mont261
  return f29_mul<F9>(load_unpack(&hi[e >> h]), load_unpack(&lo[e & ((1u << h) - 1)]));
}
__device__ __forceinline__ f29 powtab(const fe* lo, const fe* hi, uint32_t h, uint32_t full, uint32_t e) {
  return full ? load_unpack(&lo[e]) : pow2tab(lo, hi, h, e);
}

// ---- host: a few field elements per call --------------------------------------------------------------------------------------------
inline fe host_fe(const uint64_t w[4]) {
  fe r;
  memcpy(r.v, w, 32);
  return r;
}

// host-side Fr arithmetic for the handful of per-call constants of evaluate_h (HConsts): Montgomery-2^256 words in,
// canonical words out, through the same f29 layer the device uses (plain C++ there)
inline fe h_canon(const f29& x_lt2p) {
  fe o;
  f29_pack(f29_reduce_canonical<F9>(x_lt2p), o.v);
  return o;
}
inline fe h_mul256(const fe& a, const fe& b) {  // (a 2^256, b 2^256) -> a b 2^256
  return h_canon(f29_mul<F9>(f29_from_mont256<F9>(a.v), f29_unpack(b.v)));
}
inline fe h_level(const fe& a, int level) {  // a 2^256 -> a 2^256 2^(-5 level)
  fe r = a;
  const fe up = h_canon(f29_mul<F9>(f29_const<F9>(F9::ONE), f29_const<F9>(F9::ONE)));   // 2^261: times it = * 2^5 in the 2^256 domain
  // 2^-5 in the 2^256 domain = the memory word 2^251 = mul(2^256, 2^256) in f29 (divides by 2^261)
  const fe down = h_canon(f29_mul<F9>(f29_const<F9>(F9::TO256), f29_const<F9>(F9::TO256)));
  for (int i = 0; i < (level < 0 ? -level : level); i++) r = h_mul256(r, level < 0 ? up : down);
  return r;
}
// ---- the power-table cache and the per-stream scratch vectors (owned by h2mi_ntt.hip) -----------------------------------------------
struct PowTab {  // base^i split as hi/lo for i < 2^log_n; full: lo holds every power (h = log_n, hi = {1})
  fe* lo = nullptr;
  fe* hi = nullptr;
  uint32_t h = 0;
  bool full = false;
  size_t bytes = 0;
  Built built;            // build kernels' completion: consumers on other streams wait for it
  uint64_t last_use = 0;  // call epoch of the last user (entries of the running call are never evicted)
};
// one per ABI call that uses cached tables: entries handed out during the call are pinned until the next one
void tables_new_call();
struct CallScope {
  CallScope() { tables_new_call(); }
};
int get_powtab(const uint64_t base[4], uint32_t log_n, hipStream_t s, PowTab* out, bool full = false);
int get_powtabs(const uint64_t* bases /* m x 4 */, size_t m, uint32_t log_n, hipStream_t s, PowTab* out);
// scratch of at least `elems` field elements for the call running on stream s (callers hold the library mutex); release_tmp orders
// a later user of the same scratch on another stream behind this call's kernels
int ensure_tmp(size_t elems, hipStream_t s);
int release_tmp(hipStream_t s);
fe* tmp_base();

}  // namespace h2
