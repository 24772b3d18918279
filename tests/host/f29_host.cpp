// Host-side harness for the plain-C++ field/curve layer (halo2-scaffold_amd/csrc/f29.cuh, g1_29.cuh):
// the same code the GPU kernels inline, compiled with g++ so the arithmetic is verified on the CPU
// against the big-integer oracle before it ever runs on a GPU.  Test infrastructure only.
#include <vector>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../halo2-scaffold_amd/csrc/g1_29.cuh"

using namespace h2;

template <class F>
static void mul_words(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n, int mode) {
  for (size_t i = 0; i < n; i++) {
    if (mode == 0) {  // Mont256 x Mont256 -> Mont256 through the internal Mont261 domain
      f29 x = f29_from_mont256<F>(a + 8 * i), y = f29_from_mont256<F>(b + 8 * i);
      f29_to_mont256<F>(f29_mul<F>(x, y), out + 8 * i);
    } else if (mode == 1) {  // NTT butterfly style: data stays Mont256, twiddle is Mont261
      f29 d = f29_unpack(a + 8 * i), w = f29_from_mont256<F>(b + 8 * i);
      f29 r = f29_reduce_canonical<F>(f29_mul<F>(d, w));
      f29_pack(r, out + 8 * i);
    } else if (mode == 2) {  // lazy chain: (a + b) * (a - b + 2p) with un-normalized first operand
      f29 x = f29_from_mont256<F>(a + 8 * i), y = f29_from_mont256<F>(b + 8 * i);
      f29 s = f29_normalize(f29_add(x, y));
      f29 d = f29_sub(x, y, F::K2);
      f29_to_mont256<F>(f29_mul<F>(d, s), out + 8 * i);
    } else if (mode == 4) {  // dedicated squaring of a normalized input
      f29 x = f29_from_mont256<F>(a + 8 * i);
      f29_to_mont256<F>(f29_sqr<F>(x), out + 8 * i);
    } else if (mode == 5) {  // Fermat inversion
      f29 x = f29_from_mont256<F>(a + 8 * i);
      f29_to_mont256<F>(f29_inv<F>(x), out + 8 * i);
    } else {  // pack(unpack(x)) round trip
      f29 x = f29_unpack(a + 8 * i);
      f29_pack(x, out + 8 * i);
    }
  }
}

extern "C" {
void f29t_mul(int field, int mode, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  if (field == 0) mul_words<Fq29>(a, b, out, n, mode);
  else mul_words<Fr29>(a, b, out, n, mode);
}

// f29_reduce_loose on raw normalized 9-limb values (value < 64p): in 9 words, out 9 words per element
void f29t_reduce_loose(int field, const uint32_t* in, uint32_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    f29 x, r;
    for (int k = 0; k < 9; k++) x.v[k] = in[9 * i + k];
    r = field == 0 ? f29_reduce_loose<Fq29>(x) : f29_reduce_loose<Fr29>(x);
    for (int k = 0; k < 9; k++) out[9 * i + k] = r.v[k];
  }
}

// f29_mul2 on raw limb patterns (9 words each): out = (a*b + c*d) / 2^261, normalized limbs
void f29t_mul2_raw(int field, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    f29 x[4], r;
    const uint32_t* src[4] = {a, b, c, d};
    for (int q = 0; q < 4; q++)
      for (int k = 0; k < 9; k++) x[q].v[k] = src[q][9 * i + k];
    r = field == 0 ? f29_mul2<Fq29>(x[0], x[1], x[2], x[3]) : f29_mul2<Fr29>(x[0], x[1], x[2], x[3]);
    for (int k = 0; k < 9; k++) out[9 * i + k] = r.v[k];
  }
}

// f29_mul3 on raw limb patterns (six operands, 9 words each): out = (a*b + c*d + e*f) / 2^261
void f29t_mul3_raw(int field, const uint32_t* ops /* [6][n][9] */, uint32_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    f29 x[6], r;
    for (int q = 0; q < 6; q++)
      for (int k = 0; k < 9; k++) x[q].v[k] = ops[((size_t)q * n + i) * 9 + k];
    r = field == 0 ? f29_mul3<Fq29>(x[0], x[1], x[2], x[3], x[4], x[5]) : f29_mul3<Fr29>(x[0], x[1], x[2], x[3], x[4], x[5]);
    for (int k = 0; k < 9; k++) out[9 * i + k] = r.v[k];
  }
}

// accumulate n affine points (Mont256, 16 words each; (0,0) skipped) with signs[i] != 0 meaning -P_i;
// writes the XYZZ result as 4 x 8 words Mont256 (canonical)
void f29t_madd_chain(const uint32_t* pts, const uint8_t* signs, size_t n, uint32_t* out_xyzz, int tree) {
  xyzz29 acc = xyzz29_identity();
  for (size_t i = 0; i < (tree ? 0 : n); i++) {
    const uint32_t* p = pts + 16 * i;
    bool id = true;
    for (int k = 0; k < 16; k++) id = id && p[k] == 0;
    if (id) continue;
    // table format: canonical Mont261 packed words
    uint32_t xw[8], yw[8];
    f29 x = f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(p));
    f29 y = f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(p + 8));
    f29_pack(x, xw);
    f29_pack(y, yw);
    f29 x2 = f29_unpack(xw), y2 = f29_unpack(yw);
    if (signs[i]) y2 = f29_sub(f29_zero(), y2, Fq29::K2);
    xyzz29_madd(acc, x2, y2);
  }
  if (tree) {
    // exercise the full XYZZ addition / doubling: split the points into `tree` groups, accumulate each
    // with mixed additions, then fold the group sums pairwise (and double-check 2S = S + S)
    xyzz29 groups[16];
    for (int g = 0; g < tree; g++) groups[g] = xyzz29_identity();
    for (size_t i = 0; i < n; i++) {
      const uint32_t* p = pts + 16 * i;
      bool id = true;
      for (int k = 0; k < 16; k++) id = id && p[k] == 0;
      if (id) continue;
      uint32_t xw[8], yw[8];
      f29_pack(f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(p)), xw);
      f29_pack(f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(p + 8)), yw);
      f29 x2 = f29_unpack(xw), y2 = f29_unpack(yw);
      if (signs[i]) y2 = f29_sub(f29_zero(), y2, Fq29::K2);
      xyzz29_madd(groups[i % tree], x2, y2);
    }
    for (int stride = 1; stride < tree; stride *= 2)
      for (int g = 0; g + stride < tree; g += 2 * stride) xyzz29_add(groups[g], groups[g + stride]);
    acc = groups[0];
    if (tree == 16) {  // (S + S) - via add's doubling branch - then + (-2S) computed by dbl ... keep S: S + S - S - S + S
      xyzz29 s2 = acc;
      xyzz29_add(s2, acc);              // doubling branch of add
      xyzz29 d = xyzz29_dbl(acc);       // explicit doubling
      d.y = f29_normalize(f29_sub(f29_zero(), d.y, Fq29::K4));  // -2S
      xyzz29_add(s2, d);                // 2S + (-2S) = identity
      xyzz29_add(s2, acc);              // identity + S = S
      acc = s2;
    }
  }
  if (xyzz29_is_identity(acc)) {
    memset(out_xyzz, 0, 128);
    return;
  }
  f29_to_mont256<Fq29>(acc.x, out_xyzz);
  f29_to_mont256<Fq29>(acc.y, out_xyzz + 8);
  f29_to_mont256<Fq29>(acc.zz, out_xyzz + 16);
  f29_to_mont256<Fq29>(acc.zzz, out_xyzz + 24);
}

// the pair-affine accumulation on the host: consecutive points are added in pairs in AFFINE coordinates (affine29_pair_add, the
// inverses of x2 - x1 by Montgomery's trick over the whole list, as the kernels share them), each pair's sum enters the XYZZ
// accumulator by a mixed addition; a pair with equal x (P + P, P - P) or an identity member goes in as two singles — the routing
// k_msm_pa_forward / k_msm_pa_backward apply.  Same input / output format as f29t_madd_chain.
void f29t_pair_chain(const uint32_t* pts, const uint8_t* signs, size_t n, uint32_t* out_xyzz) {
  std::vector<f29> xs(n), ys(n);
  std::vector<bool> ident(n);
  for (size_t i = 0; i < n; i++) {
    const uint32_t* p = pts + 16 * i;
    bool id = true;
    for (int k = 0; k < 16; k++) id = id && p[k] == 0;
    ident[i] = id;
    uint32_t xw[8], yw[8];
    f29_pack(f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(p)), xw);
    f29_pack(f29_reduce_canonical<Fq29>(f29_from_mont256<Fq29>(p + 8)), yw);
    xs[i] = f29_unpack(xw);
    ys[i] = f29_unpack(yw);
  }
  auto same_x = [&](size_t a, size_t b) {
    bool eq = true;
    for (int k = 0; k < 9; k++) eq = eq && xs[a].v[k] == xs[b].v[k];
    return eq;
  };
  const size_t npairs = n / 2;
  std::vector<bool> valid(npairs);
  std::vector<f29> before(npairs);
  f29 prod = f29_const<Fq29>(Fq29::ONE);
  for (size_t j = 0; j < npairs; j++) {
    valid[j] = !ident[2 * j] && !ident[2 * j + 1] && !same_x(2 * j, 2 * j + 1);
    if (!valid[j]) continue;
    before[j] = prod;
    prod = f29_mul<Fq29>(prod, affine29_pair_diff(xs[2 * j], xs[2 * j + 1]));
  }
  f29 run = f29_inv<Fq29>(prod);
  xyzz29 acc = xyzz29_identity();
  auto single = [&](size_t i) {
    if (ident[i]) return;
    f29 y = ys[i];
    if (signs[i]) y = f29_sub(f29_zero(), y, Fq29::K2);
    xyzz29_madd(acc, xs[i], y);
  };
  if (n & 1) single(n - 1);
  for (size_t j = npairs; j-- > 0;) {
    if (!valid[j]) {
      single(2 * j + 1);
      single(2 * j);
      continue;
    }
    const f29 dinv = f29_mul<Fq29>(run, before[j]);
    run = f29_mul<Fq29>(run, affine29_pair_diff(xs[2 * j], xs[2 * j + 1]));
    f29 x3, y3;
    affine29_pair_add(xs[2 * j], ys[2 * j], signs[2 * j] != 0, xs[2 * j + 1], ys[2 * j + 1], signs[2 * j + 1] != 0, dinv, x3, y3);
    xyzz29_madd(acc, x3, y3);
  }
  if (xyzz29_is_identity(acc)) {
    memset(out_xyzz, 0, 128);
    return;
  }
  f29_to_mont256<Fq29>(acc.x, out_xyzz);
  f29_to_mont256<Fq29>(acc.y, out_xyzz + 8);
  f29_to_mont256<Fq29>(acc.zz, out_xyzz + 16);
  f29_to_mont256<Fq29>(acc.zzz, out_xyzz + 24);
}
}
