"""CPU oracle (pure-Python big integers) for the BN254 MSM + NTT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or
executed by the product path (``halo2-scaffold_amd/``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it,
and only as the checker.

PARITY UNPINNED.  The reference (/root/reference) holds no field, curve, MSM or
NTT code and no golden vectors for this path: the arithmetic lives in the
un-vendored crates ``halo2_proofs`` (privacy-scaling-explorations/halo2 tag
v2023_02_02, reference Cargo.toml:13), ``halo2_proofs`` (axiom-crypto/halo2 branch
axiom/dev via halo2-base branch axiom-dev-0406, reference Cargo.toml:16) and
``halo2curves`` (0.3.x).  No Rust toolchain exists here, so the reference cannot
be run.  This file therefore restates the *published mathematics* those crates
implement and is anchored on

  * the reference's call sites: ``create_proof`` / ``keygen_vk`` / ``keygen_pk`` /
    ``ParamsKZG::setup`` (reference examples/standard_plonk.rs:29,33,34,41-49;
    src/scaffold.rs:119,132,135,174,191-199,271,284,287,322-331),
  * self-derived known-answer tests (tests/test_oracle.py): generator on curve,
    r*G = identity, (r-1)*G = -G, omega orders, NTT(delta)=ones, affine vs Jacobian
    formulas, commit(coeff; g) == commit(evals; g_lagrange).

Definitions restated:
  MSM      sum_i s_i * P_i  in G1 of y^2 = x^3 + 3 over Fq   (best_multiexp)
  NTT      out[i] = sum_j a[j] * omega^(i*j)  over Fr            (best_fft)
  layouts  Fr/Fq = 4 x u64 little-endian limbs in Montgomery form (R = 2^256),
           G1Affine = {x, y} with (0,0) = identity, G1 = Jacobian {x, y, z}, z=0 identity
"""
from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------------------
# constants (all re-derived in tests/test_oracle.py)
# ---------------------------------------------------------------------------
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # base field
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # scalar field
MONT_BITS = 256
MONT_R = 1 << MONT_BITS
FR_S = 28
FR_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R - 1) >> FR_S, R)
# primitive cube root of unity used as the extended-domain coset shift
# (halo2_proofs poly/domain.rs `g_coset = Scalar::ZETA`, restated from memory)
FR_ZETA = pow(FR_GENERATOR, (R - 1) // 3, R)
G1_B = 3
G1_GEN = (1, 2)

SEED = 0x48324D49  # "H2MI" - synthetic-input seed of SURVEY.md 8d


# ---------------------------------------------------------------------------
# Montgomery form and limb packing
# ---------------------------------------------------------------------------
def to_mont(a: int, mod: int) -> int:
    return (a << MONT_BITS) % mod


def from_mont(a: int, mod: int) -> int:
    return a * pow(MONT_R, -1, mod) % mod


def int_to_limbs(a: int):
    return [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def limbs_to_int(l) -> int:
    return int(l[0]) | (int(l[1]) << 64) | (int(l[2]) << 128) | (int(l[3]) << 192)


def pack(values, mod: int | None = None) -> np.ndarray:
    """ints -> (n,4) u64 array; Montgomery-encode when ``mod`` is given."""
    out = np.zeros((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        if mod is not None:
            v = to_mont(v, mod)
        out[i] = int_to_limbs(v)
    return out


def unpack(arr: np.ndarray, mod: int | None = None):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    vals = [limbs_to_int(row) for row in arr]
    if mod is not None:
        rinv = pow(MONT_R, -1, mod)
        vals = [v * rinv % mod for v in vals]
    return vals


def pack_points(points) -> np.ndarray:
    """affine points (None = identity) -> (n,8) u64 Montgomery limbs, identity = (0,0)."""
    out = np.zeros((len(points), 8), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is None:
            continue
        out[i, :4] = int_to_limbs(to_mont(p[0], Q))
        out[i, 4:] = int_to_limbs(to_mont(p[1], Q))
    return out


def unpack_points(arr: np.ndarray):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 8)
    rinv = pow(MONT_R, -1, Q)
    out = []
    for row in arr:
        x = limbs_to_int(row[:4])
        y = limbs_to_int(row[4:])
        if x == 0 and y == 0:
            out.append(None)
        else:
            out.append((x * rinv % Q, y * rinv % Q))
    return out


def unpack_jacobian(arr) -> tuple | None:
    """12 u64 Montgomery limbs (X,Y,Z) -> affine point or None."""
    arr = np.asarray(arr, dtype=np.uint64).reshape(12)
    rinv = pow(MONT_R, -1, Q)
    X = limbs_to_int(arr[0:4]) * rinv % Q
    Y = limbs_to_int(arr[4:8]) * rinv % Q
    Z = limbs_to_int(arr[8:12]) * rinv % Q
    return jac_to_affine((X, Y, Z))


def pack_jacobian(p, z: int = 1) -> np.ndarray:
    """affine point (None = identity) -> 12 u64 Montgomery limbs of the Jacobian representative (x z^2, y z^3, z)."""
    out = np.zeros(12, dtype=np.uint64)
    if p is None:
        out[4:8] = int_to_limbs(to_mont(1, Q))
        return out
    z %= Q
    out[0:4] = int_to_limbs(to_mont(p[0] * z * z % Q, Q))
    out[4:8] = int_to_limbs(to_mont(p[1] * z * z * z % Q, Q))
    out[8:12] = int_to_limbs(to_mont(z, Q))
    return out


# ---------------------------------------------------------------------------
# G1: y^2 = x^3 + 3 over Fq.  Affine points are (x, y) tuples, None = identity.
# ---------------------------------------------------------------------------
def is_on_curve(p) -> bool:
    if p is None:
        return True
    x, y = p
    return (y * y - x * x * x - G1_B) % Q == 0


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q)


def g1_add(p, q_):
    """chord-and-tangent affine addition (textbook)."""
    if p is None:
        return q_
    if q_ is None:
        return p
    x1, y1 = p
    x2, y2 = q_
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    y3 = (lam * (x1 - x3) - y1) % Q
    return (x3, y3)


def g1_double(p):
    return g1_add(p, p)


def g1_mul(k: int, p):
    k %= R
    acc = None
    add = p
    while k:
        if k & 1:
            acc = g1_add(acc, add)
        add = g1_add(add, add)
        k >>= 1
    return acc


# Jacobian formulas (dbl-2009-l / add-2007-bl, a = 0) - second, independent
# formulation used to cross-check the affine one and the GPU's device functions.
JAC_ID = (0, 1, 0)


def jac_from_affine(p):
    return JAC_ID if p is None else (p[0], p[1], 1)


def jac_to_affine(j):
    X, Y, Z = j
    if Z % Q == 0:
        return None
    zi = pow(Z, -1, Q)
    zi2 = zi * zi % Q
    return (X * zi2 % Q, Y * zi2 * zi % Q)


def jac_double(j):
    X, Y, Z = j
    if Z % Q == 0:
        return JAC_ID
    A = X * X % Q
    B = Y * Y % Q
    C = B * B % Q
    D = 2 * ((X + B) * (X + B) - A - C) % Q
    E = 3 * A % Q
    F = E * E % Q
    X3 = (F - 2 * D) % Q
    Y3 = (E * (D - X3) - 8 * C) % Q
    Z3 = 2 * Y * Z % Q
    return (X3, Y3, Z3)


def jac_add(j1, j2):
    X1, Y1, Z1 = j1
    X2, Y2, Z2 = j2
    if Z1 % Q == 0:
        return j2
    if Z2 % Q == 0:
        return j1
    Z1Z1 = Z1 * Z1 % Q
    Z2Z2 = Z2 * Z2 % Q
    U1 = X1 * Z2Z2 % Q
    U2 = X2 * Z1Z1 % Q
    S1 = Y1 * Z2 * Z2Z2 % Q
    S2 = Y2 * Z1 * Z1Z1 % Q
    if U1 == U2:
        if S1 == S2:
            return jac_double(j1)
        return JAC_ID
    H = (U2 - U1) % Q
    I = 4 * H * H % Q
    J = H * I % Q
    r_ = 2 * (S2 - S1) % Q
    V = U1 * I % Q
    X3 = (r_ * r_ - J - 2 * V) % Q
    Y3 = (r_ * (V - X3) - 2 * S1 * J) % Q
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % Q
    return (X3, Y3, Z3)


def jac_mul(k: int, j):
    k %= R
    acc = JAC_ID
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, j)
    return acc


def msm_naive(scalars, points):
    """sum_i s_i * P_i by double-and-add; the *definition* best_multiexp computes."""
    acc = JAC_ID
    for s, p in zip(scalars, points):
        if p is None or s % R == 0:
            continue
        acc = jac_add(acc, jac_mul(s, jac_from_affine(p)))
    return jac_to_affine(acc)


# ---------------------------------------------------------------------------
# NTT over Fr
# ---------------------------------------------------------------------------
def omega_for(log_n: int) -> int:
    """generator of the size-2^log_n domain (EvaluationDomain::new)."""
    assert 0 <= log_n <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - log_n), R)


def dft_naive(a, omega: int):
    n = len(a)
    return [sum(a[j] * pow(omega, i * j, R) for j in range(n)) % R for i in range(n)]


def ntt(a, omega: int):
    """radix-2 recursive NTT, natural order in and out; == dft_naive."""
    n = len(a)
    if n == 1:
        return [a[0] % R]
    w2 = omega * omega % R
    even = ntt(a[0::2], w2)
    odd = ntt(a[1::2], w2)
    out = [0] * n
    w = 1
    h = n // 2
    for i in range(h):
        t = w * odd[i] % R
        out[i] = (even[i] + t) % R
        out[i + h] = (even[i] - t) % R
        w = w * omega % R
    return out


def intt(a, omega: int):
    n = len(a)
    ninv = pow(n, -1, R)
    return [x * ninv % R for x in ntt(a, pow(omega, -1, R))]


def ntt_ext(a, omega: int, pre_base: int | None = None, post_scale: int | None = None):
    """a[i] *= pre_base^i ; NTT ; a[i] *= post_scale  (h2mi_ntt_ext_bn254_fr)."""
    if pre_base is not None:
        w = 1
        b = []
        for x in a:
            b.append(x * w % R)
            w = w * pre_base % R
        a = b
    out = ntt(a, omega)
    if post_scale is not None:
        out = [x * post_scale % R for x in out]
    return out


# EvaluationDomain restatement (halo2_proofs poly/domain.rs, from memory)
class Domain:
    def __init__(self, k: int, j: int):
        """k = log2 rows, j = constraint-system degree (quotient degree j-1)."""
        self.k = k
        self.n = 1 << k
        qd = j - 1
        self.quotient_poly_degree = qd
        ext = k
        while (1 << ext) < (self.n * qd):
            ext += 1
        self.extended_k = ext
        self.omega = omega_for(k)
        self.omega_inv = pow(self.omega, -1, R)
        self.extended_omega = omega_for(ext)
        self.extended_omega_inv = pow(self.extended_omega, -1, R)
        self.g_coset = FR_ZETA
        self.g_coset_inv = FR_ZETA * FR_ZETA % R
        self.ifft_divisor = pow(self.n, -1, R)
        self.extended_ifft_divisor = pow(1 << ext, -1, R)

    def lagrange_to_coeff(self, a):
        return [x * self.ifft_divisor % R for x in ntt(a, self.omega_inv)]

    def coeff_to_extended(self, a):
        a = list(a) + [0] * ((1 << self.extended_k) - len(a))
        return ntt_ext(a, self.extended_omega, pre_base=self.g_coset)

    def extended_to_coeff(self, a):
        c = [x * self.extended_ifft_divisor % R for x in ntt(a, self.extended_omega_inv)]
        w = 1
        out = []
        for x in c:
            out.append(x * w % R)
            w = w * self.g_coset_inv % R
        return out[: self.n * self.quotient_poly_degree]


# ---------------------------------------------------------------------------
# KZG SRS (ParamsKZG::setup restated: g[i] = s^i G, g_lagrange[i] = L_i(s) G)
# ---------------------------------------------------------------------------
def srs_scalars(k: int, s: int):
    n = 1 << k
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * s % R
    lag = intt(pw, omega_for(k))  # L_i(s) = (1/n) sum_j s^j omega^(-ij)
    return pw, lag


def srs(k: int, s: int):
    pw, lag = srs_scalars(k, s)
    g = [g1_mul(x, G1_GEN) for x in pw]
    gl = [g1_mul(x, G1_GEN) for x in lag]
    return g, gl


# ---------------------------------------------------------------------------
# synthetic inputs: counter-based SplitMix64, per-index, seedable
# ---------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def splitmix64_np(idx: np.ndarray, seed: int) -> np.ndarray:
    """vectorised: word i = splitmix64(seed*2^32 + i)."""
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) << np.uint64(32)) + idx.astype(np.uint64)
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _reduce_once_np(limbs: np.ndarray, mod: int) -> np.ndarray:
    """limbs (n,4) u64 with value < 2^254 -> value mod `mod` (one conditional subtract)."""
    m = np.array(int_to_limbs(mod), dtype=np.uint64)
    # lexicographic compare from the top limb
    ge = np.ones(len(limbs), dtype=bool)
    decided = np.zeros(len(limbs), dtype=bool)
    for i in (3, 2, 1, 0):
        gt = limbs[:, i] > m[i]
        lt = limbs[:, i] < m[i]
        ge = np.where(~decided & lt, False, ge)
        decided |= gt | lt
    out = limbs.copy()
    borrow = np.zeros(len(limbs), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(4):
            a = limbs[:, i]
            d = a - m[i]
            b1 = (a < m[i]).astype(np.uint64)
            d2 = d - borrow
            b2 = (d < borrow).astype(np.uint64)
            out[:, i] = np.where(ge, d2, a)
            borrow = b1 | b2
    return out


def random_field_limbs(n: int, seed: int, mod: int = R, start: int = 0) -> np.ndarray:
    """(n,4) u64: limb j of element i = splitmix64(seed, 4*(start+i)+j), top limb masked to
    62 bits (value < 2^254), then one conditional subtract of the modulus.

    The limbs ARE the in-memory (Montgomery) representation: a uniform Montgomery
    representative is a uniform field element, so no encode pass is needed.
    """
    idx = np.arange(4 * start, 4 * (start + n), dtype=np.uint64)
    w = splitmix64_np(idx, seed).reshape(n, 4)
    w[:, 3] &= np.uint64((1 << 62) - 1)
    return _reduce_once_np(w, mod)


def witness_like_limbs(n: int, seed: int) -> np.ndarray:
    """SURVEY 8d second distribution: 90 % zero, 5 % in {0,1}, 5 % uniform (Montgomery limbs)."""
    u = random_field_limbs(n, seed)
    sel = splitmix64_np(np.arange(n, dtype=np.uint64), seed ^ 0x5EED) % np.uint64(100)
    bit = splitmix64_np(np.arange(n, dtype=np.uint64), seed ^ 0xB175) & np.uint64(1)
    one = np.array(int_to_limbs(to_mont(1, R)), dtype=np.uint64)
    out = np.zeros((n, 4), dtype=np.uint64)
    is_bit = (sel >= 90) & (sel < 95) & (bit == 1)
    out[is_bit] = one
    is_uni = sel >= 95
    out[is_uni] = u[is_uni]
    return out


# ---- opening-argument helpers (halo2_proofs::arithmetic::{eval_polynomial, kate_division}, restated) ----
def eval_polynomial(poly, point: int) -> int:
    acc = 0
    for c in reversed(poly):
        acc = (acc * point + c) % R
    return acc


def kate_division(a, b: int):
    """quotient of a(X) by (X - b), remainder dropped: q[i-1] = a[i] + b*q[i] from the top coefficient down."""
    q = [0] * (len(a) - 1)
    tmp = 0
    for i in range(len(a) - 1, 0, -1):
        tmp = (a[i] + b * tmp) % R
        q[i - 1] = tmp
    return q
