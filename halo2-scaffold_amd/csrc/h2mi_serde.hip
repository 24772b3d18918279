// libh2mi.so — wire encodings of BN254 field elements and G1 points.
//
// What create_proof writes into the transcript / what ParamsKZG::{read,write} keep on disk, as restated
// in SURVEY.md 8a-0 [RECALL halo2curves 0.3.x — the crate is not available here; every convention below
// is collected in this header comment so that a maintainer can check it against the real crate in one place]:
//   Fr::to_repr / Fq::to_bytes  32 bytes, little-endian, canonical (NOT Montgomery), value < modulus
//   G1Affine::to_bytes          32 bytes: x.to_bytes() with flags in the two spare top bits of byte 31:
//                               bit 6 = y is odd (lsb of canonical y), bit 7 = point at infinity (x = 0)
//   from_bytes                  rejects x >= q, x^3 + 3 a non-residue, and infinity flag with x != 0
// The square root is y = (x^3 + 3)^((q+1)/4) (q = 3 mod 4).  Device-side because an SRS file holds 2^(k+1)
// compressed points: at k = 20 two million 254-bit exponentiations (a minute of one CPU core, milliseconds here).
#include "g1.cuh"
#include "h2mi_internal.h"

namespace h2 {

constexpr uint32_t FLAG_SIGN = 0x40u, FLAG_INF = 0x80u;  // in byte 31
__device__ __constant__ const uint32_t SQRT_EXP[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u,
                                                      0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};  // (q + 1) / 4
__device__ __constant__ const uint32_t B3_MONT[8] = {0x50ad28d7u, 0x7a17caa9u, 0xe15521b9u, 0x1f6ac17au,
                                                     0x696bd284u, 0x334bea4eu, 0xce179d8eu, 0x2a1f6744u};  // 3 * R mod q

template <class F>
__device__ __forceinline__ bool fe_lt_mod(const fe& a) {  // a < MOD as 256-bit integers
  uint32_t borrow;
  raw_sub_mod<F>(a, borrow);
  return borrow != 0;
}

// Montgomery limbs -> 32 canonical little-endian bytes (field 0 = Fq, 1 = Fr)
__global__ void __launch_bounds__(256) k_fe_to_repr(const fe* in, size_t n, fe* out, int field) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe a = fe_load(&in[i]);
  fe_store(&out[i], field ? fe_from_mont<FrP>(a) : fe_from_mont<FqP>(a));
}
// 32 canonical bytes -> Montgomery limbs; counts encodings >= modulus (their output is zero)
__global__ void __launch_bounds__(256) k_fe_from_repr(const fe* in, size_t n, fe* out, int field, unsigned long long* invalid) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe a = fe_load(&in[i]);
  const bool ok = field ? fe_lt_mod<FrP>(a) : fe_lt_mod<FqP>(a);
  fe r = fe_zero();
  if (ok) r = field ? fe_to_mont<FrP>(a) : fe_to_mont<FqP>(a);
  else atomicAdd(invalid, 1ull);
  fe_store(&out[i], r);
}

__global__ void __launch_bounds__(256) k_g1_compress(const uint8_t* aff, size_t n, fe* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine p = affine_load(aff + i * 64);
  fe r = fe_zero();
  if (affine_is_identity(p)) {
    r.v[7] = FLAG_INF << 24;
  } else {
    r = fe_from_mont<FqP>(p.x);
    fe y = fe_from_mont<FqP>(p.y);
    if (y.v[0] & 1u) r.v[7] |= FLAG_SIGN << 24;
  }
  fe_store(&out[i], r);
}

__global__ void __launch_bounds__(256) k_g1_decompress(const fe* in, size_t n, uint8_t* aff, unsigned long long* invalid) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x = fe_load(&in[i]);
  const uint32_t flags = x.v[7] >> 24;
  x.v[7] &= 0x3FFFFFFFu;
  affine p;
  p.x = fe_zero();
  p.y = fe_zero();
  bool ok = fe_lt_mod<FqP>(x);
  if (flags & FLAG_INF) {
    ok = ok && fe_is_zero(x) && !(flags & FLAG_SIGN);
  } else if (ok) {
    fe xm = fe_to_mont<FqP>(x);
    fe rhs = fe_add<FqP>(fe_mul<FqP>(fe_sqr<FqP>(xm), xm), fe_const<FqP>(B3_MONT));
    fe y = fe_one<FqP>();
    for (int w = 7; w >= 0; w--) {
      const uint32_t e = SQRT_EXP[w];
      for (int b = 31; b >= 0; b--) {
        y = fe_sqr<FqP>(y);
        if ((e >> b) & 1u) y = fe_mul<FqP>(y, rhs);
      }
    }
    ok = fe_eq(fe_sqr<FqP>(y), rhs);
    if (ok) {
      const uint32_t odd = fe_from_mont<FqP>(y).v[0] & 1u;
      if (odd != ((flags & FLAG_SIGN) ? 1u : 0u)) y = fe_neg<FqP>(y);
      p.x = xm;
      p.y = y;
    }
  }
  if (!ok) atomicAdd(invalid, 1ull);
  affine_store(aff + i * 64, p);
}

// shared driver: out-of-place kernel over n items, optional invalid counter read back (synchronises)
template <class Launch>
static int run_counted(hipStream_t s, uint64_t* invalid_out, Launch&& launch) {
  DevMem cnt;
  if (invalid_out) {
    H2_HIP(cnt.alloc(8));
    H2_HIP(hipMemsetAsync(cnt.p, 0, 8, s));
  }
  int rc = launch(cnt.as<unsigned long long>());
  if (rc) return rc;
  if (invalid_out) {
    unsigned long long v = 0;
    H2_HIP(hipMemcpyAsync(&v, cnt.p, 8, hipMemcpyDeviceToHost, s));
    H2_HIP(hipStreamSynchronize(s));
    *invalid_out = v;
  }
  return H2MI_OK;
}

}  // namespace h2

using namespace h2;

extern "C" {

int h2mi_fe_to_repr_dev(int field, const void* d_in, size_t n, void* d_out32, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_in || !d_out32 || n == 0 || (field != 0 && field != 1)) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  H2_LAUNCH("k_fe_to_repr", k_fe_to_repr, ceil_div_u32(n, 256), 256, 0, s, (const fe*)d_in, n, (fe*)d_out32, field);
  return H2MI_OK;
}

int h2mi_fe_from_repr_dev(int field, const void* d_in32, size_t n, void* d_out, uint64_t* invalid_out) {
  H2_REQUIRE_INIT();
  if (!d_in32 || !d_out || !invalid_out || n == 0 || (field != 0 && field != 1)) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  return run_counted(s, invalid_out, [&](unsigned long long* cnt) {
    H2_LAUNCH("k_fe_from_repr", k_fe_from_repr, ceil_div_u32(n, 256), 256, 0, s, (const fe*)d_in32, n, (fe*)d_out, field, cnt);
    return H2MI_OK;
  });
}

int h2mi_g1_compress_dev(const void* d_affine, size_t n, void* d_out32, h2mi_stream_t stream) {
  H2_REQUIRE_INIT();
  if (!d_affine || !d_out32 || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = pick_stream(stream);
  H2_LAUNCH("k_g1_compress", k_g1_compress, ceil_div_u32(n, 256), 256, 0, s, (const uint8_t*)d_affine, n, (fe*)d_out32);
  return H2MI_OK;
}

int h2mi_g1_decompress_dev(const void* d_in32, size_t n, void* d_affine_out, uint64_t* invalid_out) {
  H2_REQUIRE_INIT();
  if (!d_in32 || !d_affine_out || !invalid_out || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  return run_counted(s, invalid_out, [&](unsigned long long* cnt) {
    H2_LAUNCH("k_g1_decompress", k_g1_decompress, ceil_div_u32(n, 256), 256, 0, s, (const fe*)d_in32, n, (uint8_t*)d_affine_out, cnt);
    return H2MI_OK;
  });
}

// host-pointer forms (synchronous)
int h2mi_g1_compress(const uint64_t* affine, size_t n, uint8_t* out32) {
  H2_REQUIRE_INIT();
  if (!affine || !out32 || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  DevMem din, dout;
  H2_HIP(din.alloc(n * 64));
  H2_HIP(dout.alloc(n * 32));
  H2_HIP(hipMemcpyAsync(din.p, affine, n * 64, hipMemcpyHostToDevice, s));
  H2_LAUNCH("k_g1_compress", k_g1_compress, ceil_div_u32(n, 256), 256, 0, s, din.as<uint8_t>(), n, dout.as<fe>());
  H2_HIP(hipMemcpyAsync(out32, dout.p, n * 32, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  return H2MI_OK;
}

int h2mi_g1_decompress(const uint8_t* in32, size_t n, uint64_t* affine_out, uint64_t* invalid_out) {
  H2_REQUIRE_INIT();
  if (!in32 || !affine_out || !invalid_out || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  DevMem din, dout;
  H2_HIP(din.alloc(n * 32));
  H2_HIP(dout.alloc(n * 64));
  H2_HIP(hipMemcpyAsync(din.p, in32, n * 32, hipMemcpyHostToDevice, s));
  int rc = run_counted(s, invalid_out, [&](unsigned long long* cnt) {
    H2_LAUNCH("k_g1_decompress", k_g1_decompress, ceil_div_u32(n, 256), 256, 0, s, din.as<fe>(), n, dout.as<uint8_t>(), cnt);
    return H2MI_OK;
  });
  if (rc) return rc;
  H2_HIP(hipMemcpyAsync(affine_out, dout.p, n * 64, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  return H2MI_OK;
}

}  // extern "C"
