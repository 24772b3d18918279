// libh2mi_hooks.so = the product's objects + this file: elementwise device arithmetic exposed for the parity tests
// (tests/test_gpu_parity.py: every field operation, the XYZZ point formulas and the lane-cooperative quad operations against the
// oracle).  NOT part of the product: libh2mi.so neither contains nor exports these (round-4 VERDICT: debug hooks shipped in the
// product library).  The kernels below instantiate the same device functions (fp.cuh, g1.cuh, g1_29_quad.cuh) the product's kernels
// inline; the hooks library carries its own copy of the library state and is initialised separately (h2mi_init).
#include "g1.cuh"
#include "g1_29.cuh"
#include "g1_29_quad.cuh"
#include "h2mi_hooks.h"
#include "h2mi_internal.h"

namespace h2 {

// ---- elementwise field kernels (test hooks) -------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_dbg_field(int op, const fe* a, const fe* b, fe* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x = fe_load(&a[i]);
  fe y = b ? fe_load(&b[i]) : fe_zero();
  fe r;
  switch (op) {
    case 0: r = fe_mul<F>(x, y); break;
    case 1: r = fe_add<F>(x, y); break;
    case 2: r = fe_sub<F>(x, y); break;
    case 3: r = fe_sqr<F>(x); break;
    case 4: r = fe_inv<F>(x); break;
    case 5: r = fe_from_mont<F>(x); break;
    case 6: r = fe_to_mont<F>(x); break;
    case 7: r = fe_neg<F>(x); break;
    case 9: r = fe_inv_ds<F>(x); break;
    case 10: r = fe_inv_gcd<F>(x); break;
    default: r = fe_dbl<F>(x); break;
  }
  fe_store(&out[i], r);
}

__global__ void __launch_bounds__(256) k_dbg_g1(int op, const uint8_t* p, const uint8_t* q, uint8_t* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine P = affine_load(p + i * 64);
  xyzz acc;
  if (op == 0) {
    affine Q = affine_load(q + i * 64);
    acc = xyzz_from_affine(P);
    xyzz_madd(acc, Q);
  } else if (op == 1) {
    acc = xyzz_from_affine(P);
    acc = xyzz_dbl(acc);
  } else {
    affine Q = affine_load(q + i * 64);
    acc = xyzz_from_affine(P);
    // make the second operand a non-trivial XYZZ representative: (2Q) - Q computed as 2Q + (-Q)
    xyzz b = xyzz_dbl(xyzz_from_affine(Q));
    affine nq = Q;
    nq.y = fe_neg<Fq>(Q.y);
    xyzz_madd(b, nq);
    xyzz_add(acc, b);
  }
  jac_store(out + i * 96, xyzz_to_jac(acc));
}

// debug hook for the lane-cooperative point operations (g1_29_quad.cuh): four lanes per element.
// op 0: (P) + (Q) with both operands brought to non-trivial XYZZ representatives; op 1: 2 (P).
__global__ void __launch_bounds__(256) k_dbg_quad(int op, const uint8_t* pp, const uint8_t* qq, uint8_t* out, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (i >= n) return;  // n is padded by the host so that whole quads share the decision
  auto lift = [](const uint8_t* src) {  // affine Mont256 -> XYZZ with ZZ != 1: (2P) + (-P)
    affine a = affine_load(src);
    if (affine_is_identity(a)) return xyzz29_identity();
    f29 x = f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(a.x.v));
    f29 y = f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(a.y.v));
    xyzz29 r = xyzz29_dbl_affine(x, y);
    xyzz29_madd(r, x, f29_sub(f29_zero(), y, Fq29::K2));
    return r;
  };
  xyzz29 a = lift(pp + i * 64);
  xyzz29 r = op == 0 ? xyzz29_add_quad(a, lift(qq + i * 64)) : xyzz29_dbl_quad(a);
  if ((threadIdx.x & 3u) != (uint32_t)(i & 3u)) return;  // one lane of the quad writes (a different one per element)
  jac j;
  if (xyzz29_is_identity(r)) {
    j.x = fe_zero(); j.y = fe_one<Fq>(); j.z = fe_zero();
  } else {
    f29_to_mont256<Fq29>(f29_mul<Fq29>(r.x, r.zz), j.x.v);
    f29_to_mont256<Fq29>(f29_mul<Fq29>(r.y, r.zzz), j.y.v);
    f29_to_mont256<Fq29>(r.zz, j.z.v);
  }
  jac_store(out + i * 96, j);
}

}  // namespace h2

using namespace h2;

extern "C" {

int h2mi_dbg_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  H2_REQUIRE_INIT();
  if (!a || !out || n == 0) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  DevMem da, db, dout;
  H2_HIP(da.alloc(n * 32));
  H2_HIP(dout.alloc(n * 32));
  H2_HIP(hipMemcpyAsync(da.p, a, n * 32, hipMemcpyHostToDevice, s));
  if (b) {
    H2_HIP(db.alloc(n * 32));
    H2_HIP(hipMemcpyAsync(db.p, b, n * 32, hipMemcpyHostToDevice, s));
  }
  uint32_t grid = ceil_div_u32(n, 256);
  if (field == 0) {
    H2_LAUNCH("k_dbg_field_fq", k_dbg_field<FqP>, grid, 256, 0, s, op, da.as<fe>(), db.as<fe>(), dout.as<fe>(), n);
  } else {
    H2_LAUNCH("k_dbg_field_fr", k_dbg_field<FrP>, grid, 256, 0, s, op, da.as<fe>(), db.as<fe>(), dout.as<fe>(), n);
  }
  H2_HIP(hipMemcpyAsync(out, dout.p, n * 32, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  return H2MI_OK;
}


int h2mi_dbg_g1_op(int op, const uint64_t* p, const uint64_t* q, uint64_t* out_jac, size_t n) {
  H2_REQUIRE_INIT();
  if (!p || !out_jac || n == 0 || (op != 1 && !q)) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  DevMem dp, dq, dout;
  H2_HIP(dp.alloc(n * 64));
  H2_HIP(dout.alloc(n * 96));
  H2_HIP(hipMemcpyAsync(dp.p, p, n * 64, hipMemcpyHostToDevice, s));
  if (q) {
    H2_HIP(dq.alloc(n * 64));
    H2_HIP(hipMemcpyAsync(dq.p, q, n * 64, hipMemcpyHostToDevice, s));
  }
  H2_LAUNCH("k_dbg_g1", k_dbg_g1, ceil_div_u32(n, 256), 256, 0, s, op, dp.as<uint8_t>(), dq.as<uint8_t>(), dout.as<uint8_t>(), n);
  H2_HIP(hipMemcpyAsync(out_jac, dout.p, n * 96, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  return H2MI_OK;
}


int h2mi_dbg_g1_quad_op(int op, const uint64_t* p, const uint64_t* q, uint64_t* out_jac, size_t n) {
  H2_REQUIRE_INIT();
  if (!p || !out_jac || n == 0 || (op != 0 && op != 1) || (op == 0 && !q)) return H2MI_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(ctx().mu);
  hipStream_t s = ctx().stream;
  DevMem dp, dq, dout;
  H2_HIP(dp.alloc(n * 64));
  H2_HIP(dout.alloc(n * 96));
  H2_HIP(hipMemcpyAsync(dp.p, p, n * 64, hipMemcpyHostToDevice, s));
  if (q) {
    H2_HIP(dq.alloc(n * 64));
    H2_HIP(hipMemcpyAsync(dq.p, q, n * 64, hipMemcpyHostToDevice, s));
  }
  H2_LAUNCH("k_dbg_quad", k_dbg_quad, ceil_div_u32(n * 4, 256), 256, 0, s, op, dp.as<uint8_t>(), dq.as<uint8_t>(), dout.as<uint8_t>(), n);
  H2_HIP(hipMemcpyAsync(out_jac, dout.p, n * 96, hipMemcpyDeviceToHost, s));
  H2_HIP(hipStreamSynchronize(s));
  return H2MI_OK;
}

}  // extern "C"
