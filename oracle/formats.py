"""CPU oracle for the wire formats around the hot path: field / point encodings, the
Blake2b transcript and the SRS file layout.

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  PARITY UNPINNED: restated from SURVEY.md 8a-0 / 8f and
from memory of halo2curves 0.3.x (derive/curve.rs `new_curve_impl`: to_bytes / from_bytes) and
halo2_proofs v2023_02_02 (transcript.rs, poly/kzg/commitment.rs `ParamsKZG::write`); neither crate is
available here and the reference repository holds no proof or SRS bytes.  What is independently pinned:
Blake2b itself (Python's hashlib, RFC 7693), the curve equations (every decompressed point is checked on
the curve), the G2 generator (on the twist, order r).  What stays a hypothesis: flag bit positions, the
transcript's prefix bytes and personalisation, the field order inside the SRS file.

Written independently of the product code: affine chord-and-tangent arithmetic, Tonelli-free square
roots by exponentiation, integers throughout."""
from __future__ import annotations

import hashlib
import struct

from . import bn254 as o

Q, R = o.Q, o.R
FLAG_SIGN, FLAG_INF = 0x40, 0x80


# ---- field elements ---------------------------------------------------------------------------------
def fe_to_repr(v: int) -> bytes:
    return v.to_bytes(32, "little")


def fr_from_repr(b: bytes) -> int | None:
    v = int.from_bytes(b, "little")
    return v if len(b) == 32 and v < R else None


# ---- G1 -----------------------------------------------------------------------------------------------
def g1_to_bytes(p) -> bytes:
    if p is None:
        return bytes(31) + bytes([FLAG_INF])
    x, y = p
    out = bytearray(fe_to_repr(x))
    if y & 1:
        out[31] |= FLAG_SIGN
    return bytes(out)


def g1_from_bytes(b: bytes):
    """-> affine point, None for the identity; raises ValueError for an invalid encoding."""
    if len(b) != 32:
        raise ValueError("length")
    flags = b[31] & 0xC0
    x = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little")
    if x >= Q:
        raise ValueError("x out of range")
    if flags & FLAG_INF:
        if x != 0 or flags & FLAG_SIGN:
            raise ValueError("inconsistent infinity flag")
        return None
    rhs = (x * x * x + 3) % Q
    y = pow(rhs, (Q + 1) // 4, Q)
    if y * y % Q != rhs:
        raise ValueError("not on the curve")
    if (y & 1) != (1 if flags & FLAG_SIGN else 0):
        y = Q - y
    return (x, y)


# ---- G2 (affine, Fq2 = Fq[u]/(u^2+1)) -------------------------------------------------------------------
def _f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def _f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * d % Q, (-a[1]) * d % Q)


def _f2sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def _f2add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


G2_B = _f2mul((3, 0), _f2inv((9, 1)))
G2_GEN = (
    (0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED, 0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
    (0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA, 0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B),
)


def g2_on_curve(p) -> bool:
    x, y = p
    return _f2mul(y, y) == _f2add(_f2mul(_f2mul(x, x), x), G2_B)


def g2_add(P, Qp):
    if P is None:
        return Qp
    if Qp is None:
        return P
    if P[0] == Qp[0]:
        if _f2add(P[1], Qp[1]) == (0, 0):
            return None
        lam = _f2mul(_f2mul((3, 0), _f2mul(P[0], P[0])), _f2inv(_f2add(P[1], P[1])))
    else:
        lam = _f2mul(_f2sub(Qp[1], P[1]), _f2inv(_f2sub(Qp[0], P[0])))
    x3 = _f2sub(_f2sub(_f2mul(lam, lam), P[0]), Qp[0])
    return (x3, _f2sub(_f2mul(lam, _f2sub(P[0], x3)), P[1]))


def g2_mul(k: int, P=G2_GEN):
    acc = None
    k %= R
    while k:
        if k & 1:
            acc = g2_add(acc, P)
        P = g2_add(P, P)
        k >>= 1
    return acc


def g2_to_bytes(p) -> bytes:
    if p is None:
        return bytes(63) + bytes([FLAG_INF])
    (x0, x1), (y0, _) = p
    out = bytearray(fe_to_repr(x0) + fe_to_repr(x1))
    if y0 & 1:
        out[63] |= FLAG_SIGN
    return bytes(out)


# ---- transcript ---------------------------------------------------------------------------------------
class Blake2bTranscript:
    """write side and read side share the hashing; `proof` collects what a writer emits."""

    def __init__(self):
        self.h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = bytearray()

    def common_point(self, p):
        if p is None:
            raise ValueError("identity in transcript")
        self.h.update(b"\x01" + fe_to_repr(p[0]) + fe_to_repr(p[1]))

    def common_scalar(self, s: int):
        self.h.update(b"\x02" + fe_to_repr(s))

    def write_point(self, p):
        self.common_point(p)
        self.proof += g1_to_bytes(p)

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.proof += fe_to_repr(s)

    def squeeze_challenge(self) -> int:
        self.h.update(b"\x00")
        return int.from_bytes(self.h.copy().digest(), "little") % R


# ---- SRS file -------------------------------------------------------------------------------------------
def srs_bytes(k: int, s: int) -> bytes:
    """ParamsKZG::setup(k) with toxic waste s, then ParamsKZG::write."""
    g, gl = o.srs(k, s)
    out = bytearray(struct.pack("<I", k))
    for p in g:
        out += g1_to_bytes(p)
    for p in gl:
        out += g1_to_bytes(p)
    out += g2_to_bytes(G2_GEN) + g2_to_bytes(g2_mul(s))
    return bytes(out)
