// XYZZ mixed addition on the lazy 29-bit-limb representation (f29.cuh) — the inner loop of the MSM
// bucket accumulation.  Same group law as g1.cuh (madd-2008-s, complete), different field layer.
//
// Coordinates are Montgomery-2^261 values, loosely reduced.  Invariants of an accumulator between
// additions (value bounds in units of p; all limbs normalized):
//     X < 6, Y < 4, ZZ < 1.1, ZZZ < 1.1          ("acc invariant")
// Table points (x2, y2): canonical (< 1), normalized; a negated y2 is 2p - y2 (lazy limbs, < 2).
// With eps = p / 2^261 = 0.0059 and mul(A, B) < 1 + eps*A*B:
//     U2 = x2*ZZ < 1.007            S2 = y2*ZZZ < 1.014
//     P  = U2 - X + 6p < 7.007      R  = S2 - Y + 4p < 5.014          (normalized before squaring)
//     PP = P^2 < 1.29               PPP = P*PP < 1.054                Q = X*PP < 1.046
//     RR = R^2 < 1.149              X3 = RR - PPP - 2Q + 4p < 5.149   (< 6: invariant holds)
//     T  = Q - X3 + 6p < 7.05 (lazy operand)   Y' = 4p - Y < 4 (lazy operand)
//     Y3 = (T*R + Y'*PPP) / 2^261 < 1 + eps * (35.4 + 4.3) = 1.24  (< 4: invariant holds; one reduction, f29_mul2)
//     ZZ3, ZZZ3 < 1.01
#pragma once
#include "f29.cuh"

namespace h2 {

struct affine29 {
  f29 x, y;  // canonical Mont261, normalized; (0,0) = identity
};
struct xyzz29 {
  f29 x, y, zz, zzz;  // zz limbs all zero = identity
};

H2_HD bool affine29_is_identity(const affine29& p) { return f29_limbs_zero(p.x) && f29_limbs_zero(p.y); }
H2_HD bool xyzz29_is_identity(const xyzz29& p) { return f29_limbs_zero(p.zz); }
H2_HD xyzz29 xyzz29_identity() {
  xyzz29 r;
  r.x = f29_zero(); r.y = f29_zero(); r.zz = f29_zero(); r.zzz = f29_zero();
  return r;
}

// 2 * (affine point) — rare path (a bucket receiving the same point twice); everything normalized.
H2_HD xyzz29 xyzz29_dbl_affine(const f29& x, const f29& y_any) {
  using F = Fq29;
  xyzz29 r;
  f29 y = f29_normalize(y_any);
  f29 u = f29_normalize(f29_dbl(y));                           // U = 2Y            (< 4)
  f29 v = f29_sqr<F>(u);                                       // V = U^2           (< 1.1)
  f29 w = f29_mul<F>(u, v);                                    // W = U*V
  f29 s = f29_mul<F>(x, v);                                    // S = X*V
  f29 xx = f29_sqr<F>(x);
  f29 m = f29_normalize(f29_add(f29_dbl(xx), xx));             // M = 3X^2          (< 3.1)
  f29 mm = f29_sqr<F>(m);                                      // (< 1.06)
  r.x = f29_normalize(f29_sub(mm, f29_dbl(s), F::KW4));        // X3 = M^2 - 2S + 4p (< 5.1); 2S limbs < 2^30
  f29 t = f29_sub(s, r.x, F::K6);                              // S - X3 + 6p  (lazy, < 7.1)
  r.y = f29_mul2<F>(t, m, f29_sub(f29_zero(), y, F::K4), w);   // (T*M + (4p - Y)*W) / 2^261  (< 1.2)
  r.zz = v;
  r.zzz = w;
  return r;
}

// acc += (x2, y2);  y2 may be lazy (negated).  Complete: handles acc = identity, P + P, P + (-P).
// The caller skips identity table points.
H2_HD void xyzz29_madd(xyzz29& acc, const f29& x2, const f29& y2) {
  using F = Fq29;
  if (xyzz29_is_identity(acc)) {
    acc.x = x2;
    acc.y = f29_normalize(y2);
    acc.zz = f29_const<F>(F::ONE);
    acc.zzz = f29_const<F>(F::ONE);
    return;
  }
  f29 u2 = f29_mul<F>(x2, acc.zz);
  f29 s2 = f29_mul<F>(y2, acc.zzz);
  f29 p = f29_normalize(f29_sub(u2, acc.x, F::K6));
  f29 r = f29_normalize(f29_sub(s2, acc.y, F::K4));
  f29 pp = f29_sqr<F>(p);
  if (f29_is_zero_mod<F>(pp)) {  // same x: equal or opposite points (rare)
    f29 rr = f29_sqr<F>(r);
    if (f29_is_zero_mod<F>(rr)) acc = xyzz29_dbl_affine(x2, y2);
    else acc = xyzz29_identity();
    return;
  }
  f29 ppp = f29_mul<F>(p, pp);
  f29 q = f29_mul<F>(acc.x, pp);
  f29 rr = f29_sqr<F>(r);
  // X3 = RR - PPP - 2Q + 4p : subtract (PPP + 2Q) (limbs < 3 * 2^29) with the 2^31-biased 4p
  f29 sub3 = f29_add(ppp, f29_dbl(q));
  f29 x3 = f29_normalize(f29_sub(rr, sub3, F::KW4));
  f29 t = f29_sub(q, x3, F::K6);  // lazy operand (limbs < 1.5 * 2^30)
  f29 ny = f29_sub(f29_zero(), acc.y, F::K4);  // 4p - Y (lazy, limbs < 2^30)
  acc.x = x3;
  acc.y = f29_mul2<F>(t, r, ny, ppp);  // Y3 = (T*R + (4p - Y)*PPP) / 2^261: one reduction for both products
  acc.zz = f29_mul<F>(acc.zz, pp);
  acc.zzz = f29_mul<F>(acc.zzz, ppp);
}

// ---- batched-affine pair addition (round 5 experiment: see "pair-affine accumulation" in h2mi_msm.hip) ------------------------------
// Two table points of one bucket are added in AFFINE coordinates, the inversion of x2 - x1 shared by every pair of the launch
// (Montgomery's trick, hierarchically): 5 multiplications + 1 squaring per pair instead of a second mixed addition (8M + 2S).
// d = x2 - x1 + 2p for canonical x's: normalized, 0 < value < 3p, nonzero modulo p whenever x1 != x2
H2_HD f29 affine29_pair_diff(const f29& x1, const f29& x2) { return f29_normalize(f29_sub(x2, x1, Fq29::K2)); }
// (x3, y3) = (x1, s1 y1) + (x2, s2 y2) with dinv = 1 / (x2 - x1) (normalized, < 1.1p); x, y canonical table coordinates, neg = the
// entry's sign bit.  Out: x3 normalized < 5.1p, y3 lazy (limbs < 1.5 * 2^30) < 3.1p — within what xyzz29_madd takes as (x2, y2):
//   dy  = s2 y2 - s1 y1                    < 4p (normalized)
//   lam = dy * dinv                        < 1 + eps * 4 * 1.1
//   x3  = lam^2 - (x1 + x2) + 4p           < 5.02      (the 2^31-biased 4p: the subtrahend is a lazy sum)
//   t   = x1 - x3 + 6p                     < 7.02      (lazy operand, as T in xyzz29_madd)
//   y3  = t * lam - s1 y1                  < 1.05 + 2p (s1 = +: minus the canonical y1 with the 2p bias; s1 = -: plus y1)
H2_HD void affine29_pair_add(const f29& x1, const f29& y1, bool neg1, const f29& x2, const f29& y2, bool neg2, const f29& dinv, f29& x3, f29& y3) {
  using F = Fq29;
  f29 dy;
  if (neg1 == neg2) dy = neg2 ? f29_sub(y1, y2, F::K2) : f29_sub(y2, y1, F::K2);
  else dy = neg2 ? f29_sub(f29_zero(), f29_add(y1, y2), F::KW4) : f29_add(y1, y2);
  const f29 lam = f29_mul<F>(f29_normalize(dy), dinv);
  const f29 ll = f29_sqr<F>(lam);
  x3 = f29_normalize(f29_sub(ll, f29_add(x1, x2), F::KW4));
  const f29 t = f29_sub(x1, x3, F::K6);
  const f29 m = f29_mul<F>(t, lam);
  y3 = neg1 ? f29_add(m, y1) : f29_sub(m, y1, F::K2);
}

// 2 * p (XYZZ, dbl-2008-s-1).  Invariant in / out: X < 6, Y < 4, ZZ, ZZZ < 1.5 (units of p), normalized.
//   U = 2Y < 8    V = U^2 < 1.38    W = U*V < 1.07    S = X*V < 1.05    M = 3X^2 < 3.64
//   X3 = M^2 - 2S + 4p < 5.1    T = S - X3 + 6p < 7.1    Y3 = (M*T + W*(4p - Y)) / 2^261 < 1.2    ZZ3, ZZZ3 < 1.02
H2_HD xyzz29 xyzz29_dbl(const xyzz29& p) {
  using F = Fq29;
  if (xyzz29_is_identity(p)) return p;
  xyzz29 r;
  f29 u = f29_normalize(f29_dbl(p.y));
  f29 v = f29_sqr<F>(u);
  f29 w = f29_mul<F>(u, v);
  f29 s = f29_mul<F>(p.x, v);
  f29 xx = f29_sqr<F>(p.x);
  f29 m = f29_normalize(f29_add(f29_dbl(xx), xx));
  f29 mm = f29_sqr<F>(m);
  r.x = f29_normalize(f29_sub(mm, f29_dbl(s), F::KW4));
  f29 t = f29_sub(s, r.x, F::K6);
  r.y = f29_mul2<F>(t, m, f29_sub(f29_zero(), p.y, F::K4), w);  // Y3 = (T*M + (4p - Y)*W) / 2^261 < 1.2: one reduction
  r.zz = f29_mul<F>(v, p.zz);
  r.zzz = f29_mul<F>(w, p.zzz);
  return r;
}

// a += b (both XYZZ, add-2008-s, complete).  Same invariant as xyzz29_dbl.
//   U1, U2 < 1.06   S1, S2 < 1.04   P = U2 - U1 + 2p < 3.1   R = S2 - S1 + 2p < 3.1
//   PP < 1.06  PPP < 1.02  Q = U1*PP < 1.01  X3 = R^2 - PPP - 2Q + 4p < 5.1
//   T = Q - X3 + 6p < 7.1   Y3 = (R*T + (2p - S1)*PPP) / 2^261 < 1.2
H2_HD void xyzz29_add(xyzz29& a, const xyzz29& b) {
  using F = Fq29;
  if (xyzz29_is_identity(b)) return;
  if (xyzz29_is_identity(a)) { a = b; return; }
  f29 u1 = f29_mul<F>(a.x, b.zz);
  f29 u2 = f29_mul<F>(b.x, a.zz);
  f29 s1 = f29_mul<F>(a.y, b.zzz);
  f29 s2 = f29_mul<F>(b.y, a.zzz);
  f29 p = f29_normalize(f29_sub(u2, u1, F::K2));
  f29 r = f29_normalize(f29_sub(s2, s1, F::K2));
  f29 pp = f29_sqr<F>(p);
  if (f29_is_zero_mod<F>(pp)) {
    f29 rr0 = f29_sqr<F>(r);
    if (f29_is_zero_mod<F>(rr0)) a = xyzz29_dbl(a);
    else a = xyzz29_identity();
    return;
  }
  f29 ppp = f29_mul<F>(p, pp);
  f29 q = f29_mul<F>(u1, pp);
  f29 rr = f29_sqr<F>(r);
  f29 x3 = f29_normalize(f29_sub(rr, f29_add(ppp, f29_dbl(q)), F::KW4));
  f29 t = f29_sub(q, x3, F::K6);
  f29 zz = f29_mul<F>(f29_mul<F>(a.zz, b.zz), pp);
  f29 zzz = f29_mul<F>(f29_mul<F>(a.zzz, b.zzz), ppp);
  a.x = x3;
  a.y = f29_mul2<F>(t, r, f29_sub(f29_zero(), s1, F::K2), ppp);  // Y3 = (T*R + (2p - S1)*PPP) / 2^261 < 1.2: one reduction
  a.zz = zz;
  a.zzz = zzz;
}

// XYZZ -> affine (canonical Montgomery-2^261 limbs): x = X / ZZ, y = Y / ZZZ with one inversion
// (1/ZZ = (ZZ / ZZZ)^2 because ZZ^3 = ZZZ^2).  Identity -> (0, 0).
H2_HD void xyzz29_to_affine(const xyzz29& p, f29& x, f29& y) {
  using F = Fq29;
  if (xyzz29_is_identity(p)) { x = f29_zero(); y = f29_zero(); return; }
  f29 zi = f29_inv<F>(p.zzz);
  y = f29_reduce_canonical<F>(f29_mul<F>(p.y, zi));
  f29 t = f29_mul<F>(p.zz, zi);
  x = f29_reduce_canonical<F>(f29_mul<F>(p.x, f29_sqr<F>(t)));
}

}  // namespace h2
