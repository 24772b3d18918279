// libh2mi.so — group-valued radix-2 FFT over BN254 G1 (SURVEY.md 8a row a3 for G = G1, 8f-4).
//
// halo2_proofs::arithmetic::best_fft is generic over `Group`; besides Fr it is instantiated for G1 exactly once on the
// reference's path: ParamsKZG::setup (reference examples/standard_plonk.rs:29; gen_srs through src/scaffold.rs:119,174,271)
// turns the monomial SRS g[i] = s^i G into the Lagrange one by `best_fft(&mut g_lagrange_projective, root.invert(), k)` and a
// scaling by n^-1 [poly/kzg/commitment.rs, restated from memory].  A butterfly's twiddle multiplication is a SCALAR
// multiplication of a point by a 254-bit field element (Group::group_scale), so the transform costs (n / 2) log n of those — a
// keygen-time cost of minutes on the CPU at DEGREE 20-22, and the only part of setup that has no secret-free shortcut.
//
// Here: the points are lifted once to XYZZ coordinates on the lazy 29-bit-limb layer (g1_29.cuh: the MSM's own formulas, complete:
// identity operands, P + P, P - P) into a scratch vector in bit-reversed order, log n rounds of in-place decimation-in-time
// butterflies follow — one thread per butterfly: t = [w] b by double-and-add over w's canonical bits, (u + t, u - t) — and the
// last kernel applies the optional scale (one more scalar multiplication per point, as the crate does) and normalises to affine.
// Integer-multiply bound like the MSM accumulation (no MFMA: these are not dense contractions); 2 x 144 B of HBM traffic per
// butterfly against ~380 point operations.
#include "g1.cuh"
#include "g1_29.cuh"
#include "h2mi_fr_tables.h"

namespace h2 {

constexpr uint32_t G1FFT_POINT_WORDS = 36;  // X, Y, ZZ, ZZZ: nine 29-bit limbs each
struct Scalar256 {
  uint32_t w[8];  // canonical integer below r, little-endian words
};

__device__ __forceinline__ xyzz29 g1fft_load(const uint32_t* p) {
  xyzz29 r;
#pragma unroll
  for (int l = 0; l < 9; l++) {
    r.x.v[l] = p[l];
    r.y.v[l] = p[9 + l];
    r.zz.v[l] = p[18 + l];
    r.zzz.v[l] = p[27 + l];
  }
  return r;
}
__device__ __forceinline__ void g1fft_store(uint32_t* p, const xyzz29& a) {
#pragma unroll
  for (int l = 0; l < 9; l++) {
    p[l] = a.x.v[l];
    p[9 + l] = a.y.v[l];
    p[18 + l] = a.zz.v[l];
    p[27 + l] = a.zzz.v[l];
  }
}
__device__ __forceinline__ xyzz29 g1fft_neg(const xyzz29& a) {
  xyzz29 r = a;
  if (!xyzz29_is_identity(a)) r.y = f29_normalize(f29_sub(f29_zero(), a.y, Fq29::K4));  // 4p - Y < 4: the formulas' invariant
  return r;
}
// [k] b, k a canonical 254-bit integer: left-to-right double-and-add (complete formulas: b may be the identity).  The scalar is
// shifted through its top bit (static indices only: a bit index into the word array would put it in scratch memory); doubling the
// identity returns at once, so leading zero bits cost nothing.
__device__ xyzz29 g1fft_scale(const xyzz29& b, const Scalar256& k) {
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = k.w[j];
  xyzz29 acc = xyzz29_identity();
  for (int i = 0; i < 256; i++) {
    const bool bit = (w[7] >> 31) != 0;
#pragma unroll
    for (int j = 7; j > 0; j--) w[j] = (w[j] << 1) | (w[j - 1] >> 31);
    w[0] <<= 1;
    acc = xyzz29_dbl(acc);
    if (bit) xyzz29_add(acc, b);
  }
  return acc;
}

// scratch[bitrev(i)] = lift(in[i]): affine Montgomery-2^256 (the ABI's G1Affine, (0, 0) = identity) -> XYZZ, 29-bit limbs
__global__ void __launch_bounds__(256) k_g1fft_load(const uint8_t* in, uint32_t* scratch, uint32_t log_n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  const affine a = affine_load(in + (size_t)i * 64);
  xyzz29 p = xyzz29_identity();
  if (!affine_is_identity(a)) {
    p.x = f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(a.x.v));
    p.y = f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(a.y.v));
    p.zz = f29_const<Fq29>(Fq29::ONE);
    p.zzz = f29_const<Fq29>(Fq29::ONE);
  }
  const uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0;
  g1fft_store(scratch + (size_t)r * G1FFT_POINT_WORDS, p);
}

// round with half = 2^log_half: butterfly (blk, i) on a[blk * 2 half + i], a[.. + half] with twiddle omega^(i n / (2 half))
__global__ void __launch_bounds__(256) k_g1fft_round(uint32_t* a, uint32_t log_n, uint32_t log_half, const fe* wlo, const fe* whi, uint32_t wh) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (1u << (log_n - 1))) return;
  const uint32_t i = t & ((1u << log_half) - 1), blk = t >> log_half;
  uint32_t* left = a + ((size_t)(blk << (log_half + 1)) + i) * G1FFT_POINT_WORDS;
  uint32_t* right = left + ((size_t)G1FFT_POINT_WORDS << log_half);
  xyzz29 u = g1fft_load(left);
  xyzz29 b = g1fft_load(right);
  if (i != 0) {  // twiddle one: nothing to multiply by (every butterfly of the first round)
    // omega^e as a plain integer: the table entries are Montgomery-2^261 words, and a multiplication by the integer 1 divides by 2^261
    const uint32_t e = i << (log_n - 1 - log_half);
    f29 one = f29_zero();
    one.v[0] = 1;
    fe w;
    f29_pack(f29_reduce_canonical<F9>(f29_mul<F9>(pow2tab(wlo, whi, wh, e), one)), w.v);
    Scalar256 k;
#pragma unroll
    for (int j = 0; j < 8; j++) k.w[j] = w.v[j];
    b = g1fft_scale(b, k);
  }
  xyzz29 s = u;
  xyzz29_add(s, b);
  xyzz29_add(u, g1fft_neg(b));
  g1fft_store(left, s);
  g1fft_store(right, u);
}

// out[i] = normalise([scale] a[i]) as affine Montgomery-2^256
__global__ void __launch_bounds__(256) k_g1fft_store(const uint32_t* a, uint32_t n, Scalar256 scale, int has_scale, uint8_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xyzz29 p = g1fft_load(a + (size_t)i * G1FFT_POINT_WORDS);
  if (has_scale) p = g1fft_scale(p, scale);
  affine o;
  if (xyzz29_is_identity(p)) {
    o.x = fe_zero();
    o.y = fe_zero();
  } else {
    f29 x, y;
    xyzz29_to_affine(p, x, y);
    f29_to_mont256<Fq29>(x, o.x.v);
    f29_to_mont256<Fq29>(y, o.y.v);
  }
  affine_store(out + (size_t)i * 64, o);
}

}  // namespace h2

using namespace h2;

extern "C" {

int h2mi_fft_bn254_g1_dev(const void* d_affine_in, void* d_affine_out, uint32_t log_n, const uint64_t omega[4], const uint64_t* post_scale,
                          h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_affine_in || !d_affine_out || !omega) return H2MI_EINVAL;
  if (log_n > 26) return H2MI_ERANGE;  // 2^26 points: 9.7 GB of scratch; the SRS of the largest circuit the window tables allow
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  CallScope scope_;
  hipStream_t s = pick_stream(stream);
  const uint32_t n = 1u << log_n;
  PowTab pw;
  if (log_n) {
    int rc = get_powtab(omega, log_n, s, &pw);  // omega^e, e < n / 2 <= 2^log_n
    if (rc) return rc;
  }
  DevMem scratch;
  hipError_t e = scratch.alloc((size_t)n * G1FFT_POINT_WORDS * 4);
  if (e == hipErrorOutOfMemory) return H2MI_ENOMEM;
  H2_HIP(e);
  H2_LAUNCH("k_g1fft_load", k_g1fft_load, ceil_div_u32(n, 256), 256, 0, s, (const uint8_t*)d_affine_in, scratch.as<uint32_t>(), log_n);
  for (uint32_t r = 0; r < log_n; r++)
    H2_LAUNCH("k_g1fft_round", k_g1fft_round, ceil_div_u32(n / 2, 256), 256, 0, s, scratch.as<uint32_t>(), log_n, r, (const fe*)pw.lo, (const fe*)pw.hi,
              pw.h);
  Scalar256 sc;
  memset(&sc, 0, sizeof(sc));
  if (post_scale) {  // Montgomery -> canonical on the host: one Montgomery product with the integer 1 through the same f29 layer
    f29 one = f29_zero();
    one.v[0] = 1;
    fe m = host_fe(post_scale), c;
    f29_pack(f29_reduce_canonical<F9>(f29_mul<F9>(f29_from_mont256<F9>(m.v), one)), c.v);
    memcpy(sc.w, c.v, 32);
  }
  H2_LAUNCH("k_g1fft_store", k_g1fft_store, ceil_div_u32(n, 256), 256, 0, s, scratch.as<uint32_t>(), n, sc, post_scale ? 1 : 0, (uint8_t*)d_affine_out);
  H2_HIP(hipStreamSynchronize(s));  // the scratch vector is released on return
  return H2MI_OK;
}

}  // extern "C"
