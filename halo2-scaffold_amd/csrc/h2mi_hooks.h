/* h2mi_hooks.h — test hooks of libh2mi_hooks.so (the product's objects + csrc/h2mi_hooks.hip); never part of libh2mi.so.
 * Elementwise device arithmetic on host arrays, used by the parity tests only. */
#ifndef H2MI_HOOKS_H
#define H2MI_HOOKS_H
#include "../../include/h2mi.h"
#ifdef __cplusplus
extern "C" {
#endif
int h2mi_dbg_field_op(int field /*0=Fq,1=Fr*/, int op /*0=mul,1=add,2=sub,3=sqr,4=inv (Fermat),5=from_mont,6=to_mont,7=neg,8=dbl,9=inv by division steps,10=inv by binary Euclid*/,
                      const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* op 0: out = P + Q (affine inputs, via XYZZ mixed add); 1: 2P; 2: P + Q via XYZZ full add; output Jacobian (12 limbs each) */
int h2mi_dbg_g1_op(int op, const uint64_t* p_affine, const uint64_t* q_affine, uint64_t* out_jac, size_t n);
/* the lane-cooperative point operations of the bucket reduction (csrc/g1_29_quad.cuh), four lanes per
 * element: op 0 = P[i] + Q[i] (XYZZ + XYZZ), op 1 = 2 P[i]; affine Montgomery in, Jacobian out */
int h2mi_dbg_g1_quad_op(int op, const uint64_t* p, const uint64_t* q_or_null, uint64_t* out_jac, size_t n);
#ifdef __cplusplus
}
#endif
#endif
