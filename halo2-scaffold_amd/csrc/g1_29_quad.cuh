// Lane-cooperative XYZZ point operations: the four lanes of a DPP quad compute ONE addition / doubling.
//
// Why: the bucket reduction of an MSM (fold ... final in h2mi_msm.hip) is a chain of ~40 dependent point
// operations over little data; a lone wavefront needs 5-9 us per operation (14 dependent field
// multiplications for an addition, 10 for a doubling), so the chain's latency — not its throughput — is what
// a prover waits for at every transcript join.  The formulas have 4-way parallelism: an addition is four
// ROUNDS of up to four independent multiplications (a doubling three), so a quad does one multiplication
// per lane per round and exchanges the products with DPP quad broadcasts (one v_mov_dpp per limb).
//   add:  1) U1 = X1*ZZ2   U2 = X2*ZZ1   S1 = Y1*ZZZ2   S2 = Y2*ZZZ1        P = U2 - U1, R = S2 - S1
//         2) PP = P^2      RR = R^2      ZZ12 = ZZ1*ZZ2 ZZZ12 = ZZZ1*ZZZ2
//         3) PPP = P*PP    Q = U1*PP     ZZ3 = ZZ12*PP  -                   X3 = RR - PPP - 2Q
//         4) T1 = (Q-X3)*R T2 = S1*PPP   ZZZ3 = ZZZ12*PPP -                 Y3 = T1 - T2
//   dbl:  1) V = U^2 (U = 2Y)  XX = X^2  -  -                               M = 3 XX
//         2) W = U*V       S = X*V       MM = M^2       ZZ3 = V*ZZ          X3 = MM - 2S
//         3) T1 = (S-X3)*M T2 = W*Y      ZZZ3 = W*ZZZ   -                   Y3 = T1 - T2
// Operands and results are REPLICATED in the four lanes (every lane holds both inputs and gets the result),
// so callers keep their one-value-per-lane structure with a lane stride of 4.  Same lazy-reduction choices,
// value bounds and special cases (identity operands, P + P, P + (-P)) as xyzz29_add / xyzz29_dbl in
// g1_29.cuh.  All four lanes of a quad must be active together (callers branch on quad-uniform conditions).
#pragma once
#include "g1_29.cuh"

namespace h2 {

template <int SRC>
__device__ __forceinline__ f29 quad_bcast(const f29& v) {  // every lane of the quad gets lane SRC's value
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.v[i], SRC * 0x55, 0xF, 0xF, true);
  return r;
}
// per-lane choice by role (0..3) among four replicated candidates
__device__ __forceinline__ f29 quad_pick(uint32_t role, const f29& a0, const f29& a1, const f29& a2, const f29& a3) {
  f29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t lo = (role & 1u) ? a1.v[i] : a0.v[i];
    uint32_t hi = (role & 1u) ? a3.v[i] : a2.v[i];
    r.v[i] = (role & 2u) ? hi : lo;
  }
  return r;
}

__device__ __forceinline__ xyzz29 xyzz29_dbl_quad(const xyzz29& p) {
  using F = Fq29;
  const uint32_t role = threadIdx.x & 3u;
  if (!__any(!xyzz29_is_identity(p))) return p;  // no quad of this wavefront holds a point: skip the arithmetic
  f29 u = f29_normalize(f29_dbl(p.y));
  // round 1: V = U^2 | XX = X^2
  f29 a = quad_pick(role, u, p.x, u, u);
  f29 m1 = f29_mul<F>(a, a);
  f29 v = quad_bcast<0>(m1), xx = quad_bcast<1>(m1);
  f29 m = f29_normalize(f29_add(f29_dbl(xx), xx));
  // round 2: W = U*V | S = X*V | MM = M^2 | ZZ3 = V*ZZ
  f29 m2 = f29_mul<F>(quad_pick(role, u, p.x, m, v), quad_pick(role, v, v, m, p.zz));
  f29 w = quad_bcast<0>(m2), s = quad_bcast<1>(m2), mm = quad_bcast<2>(m2);
  xyzz29 r;
  r.zz = quad_bcast<3>(m2);
  r.x = f29_normalize(f29_sub(mm, f29_dbl(s), F::KW4));
  f29 t = f29_sub(s, r.x, F::K6);
  // round 3: T1 = T*M | T2 = W*Y | ZZZ3 = W*ZZZ
  f29 m3 = f29_mul<F>(quad_pick(role, t, w, w, w), quad_pick(role, m, p.y, p.zzz, p.zzz));
  f29 t1 = quad_bcast<0>(m3), t2 = quad_bcast<1>(m3);
  r.zzz = quad_bcast<2>(m3);
  r.y = f29_normalize(f29_sub(t1, t2, F::K2));
  if (xyzz29_is_identity(p)) return p;
  return r;
}

// a + b.  Returns the sum; both operands replicated in the quad.
__device__ __forceinline__ xyzz29 xyzz29_add_quad(const xyzz29& a, const xyzz29& b) {
  using F = Fq29;
  const uint32_t role = threadIdx.x & 3u;
  // reductions over sparse data add mostly identities: when no quad of the wavefront has two real operands
  // the four rounds are skipped (wavefront-uniform branch)
  const bool a_id = xyzz29_is_identity(a), b_id = xyzz29_is_identity(b);
  if (!__any(!(a_id || b_id))) return b_id ? a : b;
  // round 1: U1 = X1*ZZ2 | U2 = X2*ZZ1 | S1 = Y1*ZZZ2 | S2 = Y2*ZZZ1
  f29 m1 = f29_mul<F>(quad_pick(role, a.x, b.x, a.y, b.y), quad_pick(role, b.zz, a.zz, b.zzz, a.zzz));
  f29 u1 = quad_bcast<0>(m1), u2 = quad_bcast<1>(m1), s1 = quad_bcast<2>(m1), s2 = quad_bcast<3>(m1);
  f29 p = f29_normalize(f29_sub(u2, u1, F::K2));
  f29 r = f29_normalize(f29_sub(s2, s1, F::K2));
  // round 2: PP = P^2 | RR = R^2 | ZZ12 = ZZ1*ZZ2 | ZZZ12 = ZZZ1*ZZZ2
  f29 m2 = f29_mul<F>(quad_pick(role, p, r, a.zz, a.zzz), quad_pick(role, p, r, b.zz, b.zzz));
  f29 pp = quad_bcast<0>(m2), rr = quad_bcast<1>(m2), zz12 = quad_bcast<2>(m2), zzz12 = quad_bcast<3>(m2);
  // round 3: PPP = P*PP | Q = U1*PP | ZZ3 = ZZ12*PP
  f29 m3 = f29_mul<F>(quad_pick(role, p, u1, zz12, p), pp);
  f29 ppp = quad_bcast<0>(m3), q = quad_bcast<1>(m3);
  xyzz29 out;
  out.zz = quad_bcast<2>(m3);
  out.x = f29_normalize(f29_sub(rr, f29_add(ppp, f29_dbl(q)), F::KW4));
  f29 t = f29_sub(q, out.x, F::K6);
  // round 4: T1 = T*R | T2 = S1*PPP | ZZZ3 = ZZZ12*PPP
  f29 m4 = f29_mul<F>(quad_pick(role, t, s1, zzz12, s1), quad_pick(role, r, ppp, ppp, ppp));
  f29 t1 = quad_bcast<0>(m4), t2 = quad_bcast<1>(m4);
  out.zzz = quad_bcast<2>(m4);
  out.y = f29_normalize(f29_sub(t1, t2, F::K2));
  // special cases: every flag is computed from replicated values, hence uniform inside the quad
  if (b_id) return a;
  if (a_id) return b;
  if (f29_is_zero_mod<F>(pp)) {  // same x: equal or opposite points (rare)
    if (f29_is_zero_mod<F>(rr)) return xyzz29_dbl_quad(a);
    return xyzz29_identity();
  }
  return out;
}

}  // namespace h2
