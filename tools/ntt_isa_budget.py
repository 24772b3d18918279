#!/usr/bin/env python3
"""Per-section ISA budget of the NTT pass kernels.

Every primitive a pass is made of — global load + limb unpack, table twiddle fetch + unpack, LDS limb-plane get / put with
the bank swizzle, staged-twiddle get + unpack, the Montgomery product, limb-wise add / sub, carry normalisation, canonical
reduction + pack + store — is compiled ALONE for gfx950 (hipcc -S, one __global__ kernel per primitive, operands from and
to global memory so that nothing folds away; the common load / store scaffold is measured by an empty kernel and
subtracted), its instructions are counted by class, and the counts are multiplied by how often one element meets the
primitive in a pass of 2^m points (csrc/h2mi_ntt.hip local_ntt: m / 2 radix-4 rounds, the first with one multiplication
per four elements).  The sum is compared with the whole kernels' static instruction mix and with the measured dynamic count
(profiles/r02_ntt_counters.txt: ~1,800 VALU instructions per element per pass).  Runs anywhere hipcc does (no GPU).
Usage: python tools/ntt_isa_budget.py [m]   (default m = 10: the two passes of a 2^20 transform)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2-scaffold_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

PRIMS = {
    # name: (device code using f29 values a, b read from in[] and writing r to out[])
    "scaffold": "r = f29_add(a, b);",  # two unpacked loads, one limb-wise add, one raw store: subtracted from the others
    "mul": "r = f29_mul<F9>(a, b);",
    "mul_x2": "r = f29_mul<F9>(f29_mul<F9>(a, b), b);",
    "add": "r = f29_add(f29_add(a, b), b);",
    "sub": "r = f29_add(f29_sub(a, b, F9::K2), b);",
    "normalize": "r = f29_normalize(f29_add(a, b));",
    "reduce_canonical_pack": "f29 tt = f29_reduce_canonical<F9>(f29_add(a, b)); uint32_t w[8]; f29_pack(tt, w); r = a; for (int i = 0; i < 8; i++) r.v[i] = w[i];",
    "reduce_loose_pack": "f29 tt = f29_reduce_loose<F9>(f29_add(a, b)); uint32_t w[8]; f29_pack(tt, w); r = a; for (int i = 0; i < 8; i++) r.v[i] = w[i];",
    "unpack": "uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = a.v[i] ^ b.v[i]; r = f29_unpack(w);",
}
SRC = r'''
#include <hip/hip_runtime.h>
#include "f29.cuh"
using namespace h2;
using F9 = Fr29;
__device__ __forceinline__ uint32_t lds_sw(uint32_t i) { return i ^ (((i >> 6) & 3u) * 21u) ^ ((i >> 8) & 3u); }
extern "C" __global__ void __launch_bounds__(512) k_prim(const uint32_t* in, uint32_t* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  f29 a, b, r;
  for (int i = 0; i < 9; i++) { a.v[i] = in[t * 18 + i]; b.v[i] = in[t * 18 + 9 + i]; }
  BODY
  for (int i = 0; i < 9; i++) out[t * 9 + i] = r.v[i];
}
extern "C" __global__ void __launch_bounds__(512) k_lds(const uint32_t* in, uint32_t* out, uint32_t stride) {
  extern __shared__ uint32_t lds[];
  const uint32_t t = threadIdx.x;
  f29 a;
  for (int i = 0; i < 9; i++) a.v[i] = in[t * 9 + i];
  { const uint32_t j = lds_sw(t * 5u + 1u); for (int l = 0; l < 9; l++) lds[l * stride + j] = a.v[l]; }   // lds_put
  __syncthreads();
  f29 r;
  { const uint32_t j = lds_sw(t ^ 37u); for (int l = 0; l < 9; l++) r.v[l] = lds[l * stride + j]; }         // lds_get
  for (int i = 0; i < 9; i++) out[t * 9 + i] = r.v[i];
}
extern "C" __global__ void __launch_bounds__(512) k_twget(const uint32_t* in, uint32_t* out, uint32_t cnt) {
  extern __shared__ uint32_t tw[];
  const uint32_t t = threadIdx.x;
  for (int l = 0; l < 8; l++) tw[l * cnt + t] = in[t * 8 + l];
  __syncthreads();
  uint32_t w[8];
  const uint32_t j = __brev(t) >> 23;
  for (int l = 0; l < 8; l++) w[l] = tw[l * cnt + j];
  f29 r = f29_unpack(w);
  for (int i = 0; i < 9; i++) out[t * 9 + i] = r.v[i];
}
'''
CLASSES = [("mad_u64", r"v_mad_u64_u32"), ("mul_lo/hi", r"v_mul_(lo|hi)_u32"), ("shift64", r"v_(lshr|lshl|ashr)rev_[bi]64|v_lshl_add_u64"),
           ("alignbit/bfe/shift32", r"v_alignbit|v_bfe|v_(lshr|lshl|ashr)rev_b32|v_lshl_or|v_lshl_add_u32|v_and_or|v_bfi"),
           ("and/or/xor", r"v_(and|or|xor)(3)?_b32|v_or3"), ("add/sub", r"v_(add|sub|subrev)(3)?_(u32|co|nc)|v_addc|v_subb|v_add3|v_add_u32|v_sub_u32|v_sub_co|v_add_co"),
           ("cmp/cndmask", r"v_cmp|v_cndmask"), ("mov/other valu", r"v_"), ("ds_read/write", r"ds_"), ("global/buffer", r"global_|buffer_|flat_"),
           ("salu/waitcnt", r"s_")]


def count(asm: str, kernel: str):
    body = asm[asm.index(kernel + ":"):]
    body = body[: body.index("s_endpgm")]
    c = collections.Counter()
    for line in body.splitlines():
        ins = line.strip().split(" ")[0]
        if not ins or ins.startswith((";", ".", "//")) or ins.endswith(":"):
            continue
        for name, pat in CLASSES:
            if re.match(pat, ins):
                c[name] += 1
                break
    return c


def compile_asm(src: str):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "p.hip")
        open(path, "w").write(src)
        out = os.path.join(d, "p.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-I", CSRC, path, "-o", out])
        return open(out).read()


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    cost = {}
    for name, body in PRIMS.items():
        cost[name] = count(compile_asm(SRC.replace("BODY", body)), "k_prim")
    lds_asm = compile_asm(SRC.replace("BODY", "r = a;"))
    cost["lds_put+get"] = count(lds_asm, "k_lds")
    cost["tw_get"] = count(lds_asm, "k_twget")
    valu = lambda c: sum(v for k, v in c.items() if k not in ("ds_read/write", "global/buffer", "salu/waitcnt"))
    sub = lambda a, b: collections.Counter({k: a[k] - b.get(k, 0) for k in a})
    sc = cost["scaffold"]
    one = {
        "mul": sub(cost["mul_x2"], cost["mul"]),  # second product of a chain: no scaffold, no load
        "add": sub(cost["add"], sc), "sub": sub(cost["sub"], sc), "normalize": sub(cost["normalize"], sc),
        "reduce_canonical+pack": sub(cost["reduce_canonical_pack"], sc), "reduce_loose+pack": sub(cost["reduce_loose_pack"], sc),
        "unpack (8 words -> 9 limbs)": sub(cost["unpack"], sc), "lds_put + lds_get (swizzled, 9 planes)": cost["lds_put+get"], "tw_get (8 planes + unpack)": cost["tw_get"],
    }
    rounds = m // 2
    # per ELEMENT of a 2^m-point pass (a radix-4 round handles 4 elements: 4 get + 4 put, 3 tw_get, 5 mul (1 in round 0), 8 add/sub, 4 normalize)
    per_elem = {
        "global load + unpack": (1, one["unpack (8 words -> 9 limbs)"]),
        "LDS fill put + output get": (1, one["lds_put + lds_get (swizzled, 9 planes)"]),
        "butterfly rounds: LDS get + put": (rounds, one["lds_put + lds_get (swizzled, 9 planes)"]),
        "butterfly rounds: staged twiddle get + unpack": ((rounds - 1) * 0.75 + 0.25, one["tw_get (8 planes + unpack)"]),
        "butterfly rounds: Montgomery products": ((rounds - 1) * 1.0 + 0.25, one["mul"]),
        "butterfly rounds: limb-wise add": (rounds * 1.0, one["add"]),
        "butterfly rounds: limb-wise sub (+ k p)": (rounds * 1.0, one["sub"]),
        "butterfly rounds: carry normalisation": (rounds * 1.0, one["normalize"]),
        "inter-pass twiddle: table fetch + unpack": (1, one["unpack (8 words -> 9 limbs)"]),
        "inter-pass twiddle: Montgomery product": (1, one["mul"]),
        "store: canonical reduction + pack": (1, one["reduce_canonical+pack"]),
    }
    print(f"# NTT pass of 2^{m} points on gfx950: instruction budget per ELEMENT per pass (k_ntt_pass_col; the row pass ends with")
    print("# reduce_loose + pack instead of the inter-pass twiddle + canonical reduction)\n")
    print("primitive costs (static instructions, one invocation):")
    for name, c in one.items():
        print(f"  {name:42s} VALU {valu(c):4d}  of which mad_u64 {c.get('mad_u64', 0):4d}  mul_lo {c.get('mul_lo/hi', 0):3d}  shift64 {c.get('shift64', 0):3d}  "
              f"shift32/alignbit {c.get('alignbit/bfe/shift32', 0):3d}  and/or {c.get('and/or/xor', 0):3d}  add/sub {c.get('add/sub', 0):3d}  ds {c.get('ds_read/write', 0):3d}")
    print("\nbudget per element per pass:")
    tot, tot_mad, tot_ds = 0.0, 0.0, 0.0
    rows = []
    for name, (times, c) in per_elem.items():
        v, md, ds = times * valu(c), times * c.get("mad_u64", 0), times * c.get("ds_read/write", 0)
        rows.append((name, times, v, md, ds))
        tot, tot_mad, tot_ds = tot + v, tot_mad + md, tot_ds + ds
    for name, times, v, md, ds in rows:
        print(f"  {name:48s} x{times:5.2f}  VALU {v:7.1f} ({100 * v / tot:4.1f} %)  mad {md:6.1f}  LDS ops {ds:5.1f}")
    print(f"  {'TOTAL':48s}         VALU {tot:7.1f}            mad {tot_mad:6.1f} ({100 * tot_mad / tot:.1f} % of VALU)  LDS ops {tot_ds:5.1f}")
    mul_book = valu(one["mul"]) - one["mul"].get("mad_u64", 0)
    print(f"\n  inside one Montgomery product: {one['mul'].get('mad_u64', 0)} multiply-adds + {mul_book} bookkeeping instructions "
          f"(m_k = acc * p' mod 2^29, the 29-bit column shifts, the masks)")
    print("  measured dynamic count (profiles/r02_ntt_counters.txt, SQ_INSTS_VALU / elements): ~1,800 per element per pass; index arithmetic, loop control and")
    print("  the twiddle staging are what this per-primitive budget leaves out.")


if __name__ == "__main__":
    main()
