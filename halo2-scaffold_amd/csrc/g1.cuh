// BN254 G1 (y^2 = x^3 + 3 over Fq) group law for gfx950.
//
// Replaces (on the device) halo2curves::bn256::{G1Affine, G1} — reference call site
// src/scaffold.rs:14, SURVEY.md 8a row a7.  Layouts at the C-ABI are the crate's:
//   G1Affine = {x, y} 64 B Montgomery, identity encoded as (0, 0)
//   G1       = {x, y, z} 96 B Jacobian, identity z = 0
// Internally bucket sums are kept in extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2; identity ZZ = 0): mixed addition costs 8M + 2S instead of Jacobian's 7M + 4S and
// needs no field inversion.  All special cases (identity operands, P + P, P + (-P)) are handled, so
// the group law is complete.
#pragma once
#include "fp.cuh"

namespace h2 {

using Fq = FqP;

struct affine {
  fe x, y;
};
struct xyzz {
  fe x, y, zz, zzz;
};
struct jac {
  fe x, y, z;
};

__device__ __forceinline__ bool affine_is_identity(const affine& p) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= p.x.v[i] | p.y.v[i];
  return o == 0;
}
__device__ __forceinline__ affine affine_load(const void* p) {
  affine a;
  a.x = fe_load(p);
  a.y = fe_load(reinterpret_cast<const char*>(p) + 32);
  return a;
}
__device__ __forceinline__ void affine_store(void* p, const affine& a) {
  fe_store(p, a.x);
  fe_store(reinterpret_cast<char*>(p) + 32, a.y);
}
__device__ __forceinline__ xyzz xyzz_identity() {
  xyzz r;
  r.x = fe_zero(); r.y = fe_zero(); r.zz = fe_zero(); r.zzz = fe_zero();
  return r;
}
__device__ __forceinline__ bool xyzz_is_identity(const xyzz& p) { return fe_is_zero(p.zz); }
__device__ __forceinline__ xyzz xyzz_from_affine(const affine& p) {
  xyzz r;
  if (affine_is_identity(p)) return xyzz_identity();
  r.x = p.x; r.y = p.y; r.zz = fe_one<Fq>(); r.zzz = fe_one<Fq>();
  return r;
}
__device__ __forceinline__ xyzz xyzz_load(const void* p) {
  const char* c = reinterpret_cast<const char*>(p);
  xyzz r;
  r.x = fe_load(c); r.y = fe_load(c + 32); r.zz = fe_load(c + 64); r.zzz = fe_load(c + 96);
  return r;
}
__device__ __forceinline__ void xyzz_store(void* p, const xyzz& a) {
  char* c = reinterpret_cast<char*>(p);
  fe_store(c, a.x); fe_store(c + 32, a.y); fe_store(c + 64, a.zz); fe_store(c + 96, a.zzz);
}

// dbl-2008-s-1 specialised to an affine input (ZZ = ZZZ = 1): 2M + 3S... written out.
__device__ __forceinline__ xyzz xyzz_dbl_affine(const affine& p) {
  xyzz r;
  fe u = fe_dbl<Fq>(p.y);            // U = 2Y
  fe v = fe_sqr<Fq>(u);              // V = U^2
  fe w = fe_mul<Fq>(u, v);           // W = U*V
  fe s = fe_mul<Fq>(p.x, v);         // S = X*V
  fe xx = fe_sqr<Fq>(p.x);
  fe m = fe_add<Fq>(fe_dbl<Fq>(xx), xx);  // M = 3X^2 (a = 0)
  r.x = fe_sub<Fq>(fe_sqr<Fq>(m), fe_dbl<Fq>(s));
  r.y = fe_sub<Fq>(fe_mul<Fq>(m, fe_sub<Fq>(s, r.x)), fe_mul<Fq>(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// dbl-2008-s-1, a = 0
__device__ __forceinline__ xyzz xyzz_dbl(const xyzz& p) {
  if (xyzz_is_identity(p)) return p;
  xyzz r;
  fe u = fe_dbl<Fq>(p.y);
  fe v = fe_sqr<Fq>(u);
  fe w = fe_mul<Fq>(u, v);
  fe s = fe_mul<Fq>(p.x, v);
  fe xx = fe_sqr<Fq>(p.x);
  fe m = fe_add<Fq>(fe_dbl<Fq>(xx), xx);
  r.x = fe_sub<Fq>(fe_sqr<Fq>(m), fe_dbl<Fq>(s));
  r.y = fe_sub<Fq>(fe_mul<Fq>(m, fe_sub<Fq>(s, r.x)), fe_mul<Fq>(w, p.y));
  r.zz = fe_mul<Fq>(v, p.zz);
  r.zzz = fe_mul<Fq>(w, p.zzz);
  return r;
}

// acc += p (p affine, possibly negated by the caller): madd-2008-s, 8M + 2S.
__device__ __forceinline__ void xyzz_madd(xyzz& acc, const affine& p) {
  if (affine_is_identity(p)) return;
  if (xyzz_is_identity(acc)) {
    acc.x = p.x; acc.y = p.y; acc.zz = fe_one<Fq>(); acc.zzz = fe_one<Fq>();
    return;
  }
  fe u2 = fe_mul<Fq>(p.x, acc.zz);
  fe s2 = fe_mul<Fq>(p.y, acc.zzz);
  fe pp_ = fe_sub<Fq>(u2, acc.x);   // P
  fe r = fe_sub<Fq>(s2, acc.y);     // R
  if (fe_is_zero(pp_)) {
    if (fe_is_zero(r)) acc = xyzz_dbl_affine(p);  // same point: double it
    else acc = xyzz_identity();                   // opposite points
    return;
  }
  fe pp = fe_sqr<Fq>(pp_);
  fe ppp = fe_mul<Fq>(pp_, pp);
  fe q = fe_mul<Fq>(acc.x, pp);
  fe x3 = fe_sub<Fq>(fe_sub<Fq>(fe_sqr<Fq>(r), ppp), fe_dbl<Fq>(q));
  fe y3 = fe_sub<Fq>(fe_mul<Fq>(r, fe_sub<Fq>(q, x3)), fe_mul<Fq>(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = fe_mul<Fq>(acc.zz, pp);
  acc.zzz = fe_mul<Fq>(acc.zzz, ppp);
}

// a += b (both XYZZ): add-2008-s, 12M + 2S.
__device__ __forceinline__ void xyzz_add(xyzz& a, const xyzz& b) {
  if (xyzz_is_identity(b)) return;
  if (xyzz_is_identity(a)) { a = b; return; }
  fe u1 = fe_mul<Fq>(a.x, b.zz);
  fe u2 = fe_mul<Fq>(b.x, a.zz);
  fe s1 = fe_mul<Fq>(a.y, b.zzz);
  fe s2 = fe_mul<Fq>(b.y, a.zzz);
  fe p = fe_sub<Fq>(u2, u1);
  fe r = fe_sub<Fq>(s2, s1);
  if (fe_is_zero(p)) {
    if (fe_is_zero(r)) a = xyzz_dbl(a);
    else a = xyzz_identity();
    return;
  }
  fe pp = fe_sqr<Fq>(p);
  fe ppp = fe_mul<Fq>(p, pp);
  fe q = fe_mul<Fq>(u1, pp);
  fe x3 = fe_sub<Fq>(fe_sub<Fq>(fe_sqr<Fq>(r), ppp), fe_dbl<Fq>(q));
  fe y3 = fe_sub<Fq>(fe_mul<Fq>(r, fe_sub<Fq>(q, x3)), fe_mul<Fq>(s1, ppp));
  fe zz = fe_mul<Fq>(fe_mul<Fq>(a.zz, b.zz), pp);
  fe zzz = fe_mul<Fq>(fe_mul<Fq>(a.zzz, b.zzz), ppp);
  a.x = x3; a.y = y3; a.zz = zz; a.zzz = zzz;
}

// XYZZ -> Jacobian without inversion: (X*ZZ, Y*ZZZ, ZZ) since ZZ^3 = ZZZ^2.  Identity -> (0, R, 0),
// the crate's G1::identity() = (0, 1, 0).
__device__ __forceinline__ jac xyzz_to_jac(const xyzz& p) {
  jac j;
  if (xyzz_is_identity(p)) {
    j.x = fe_zero(); j.y = fe_one<Fq>(); j.z = fe_zero();
    return j;
  }
  j.x = fe_mul<Fq>(p.x, p.zz);
  j.y = fe_mul<Fq>(p.y, p.zzz);
  j.z = p.zz;
  return j;
}
__device__ __forceinline__ xyzz jac_to_xyzz(const jac& j) {
  xyzz r;
  if (fe_is_zero(j.z)) return xyzz_identity();
  r.x = j.x; r.y = j.y;
  r.zz = fe_sqr<Fq>(j.z);
  r.zzz = fe_mul<Fq>(r.zz, j.z);
  return r;
}
__device__ __forceinline__ void jac_store(void* p, const jac& j) {
  char* c = reinterpret_cast<char*>(p);
  fe_store(c, j.x); fe_store(c + 32, j.y); fe_store(c + 64, j.z);
}
__device__ __forceinline__ jac jac_load(const void* p) {
  const char* c = reinterpret_cast<const char*>(p);
  jac j;
  j.x = fe_load(c); j.y = fe_load(c + 32); j.z = fe_load(c + 64);
  return j;
}

// XYZZ -> affine (one field inversion); identity -> (0, 0)
__device__ __forceinline__ affine xyzz_to_affine(const xyzz& p) {
  affine a;
  if (xyzz_is_identity(p)) { a.x = fe_zero(); a.y = fe_zero(); return a; }
  fe zi = fe_inv<Fq>(p.zzz);  // 1/ZZZ
  a.y = fe_mul<Fq>(p.y, zi);
  // 1/ZZ = ZZ^2/ZZZ^2 (because ZZ^3 = ZZZ^2)  =>  1/ZZ = (ZZ * zi)^2
  fe t = fe_mul<Fq>(p.zz, zi);
  a.x = fe_mul<Fq>(p.x, fe_sqr<Fq>(t));
  return a;
}

}  // namespace h2
