"""BN254 G2 on the host (control plane): the two G2 elements of a KZG SRS, g2 and s_g2 = s * g2.

The prover never touches them (they belong to the verifier); ParamsKZG only carries them through
`write` / `read`.  `setup(k, s)` needs one scalar multiplication, done here with Python
integers in Jacobian coordinates over Fq2 = Fq[u] / (u^2 + 1).  The twist is y^2 = x^3 + 3 / (9 + u);
the generator is the standard one (EIP-197 / halo2curves G2::generator) — checked to lie on the twist
and to have order r by tests/test_oracle.py.  Encoding [RECALL halo2curves 0.3.x]: 64 bytes = x.c0 then
x.c1, 32 little-endian canonical bytes each, flags in byte 63: 0x40 = lsb of y.c0, 0x80 = infinity.
"""
from .field import FQ_MODULUS as Q
from .field import FR_MODULUS as R

G2_GENERATOR = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def _mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def _sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % Q, 2 * a[0] * a[1] % Q)


def _add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def _sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def _small(a, k):
    return (a[0] * k % Q, a[1] * k % Q)


def _inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * d % Q, -a[1] * d % Q)


_ZERO, _ONE = (0, 0), (1, 0)


def _dbl(p):  # Jacobian, a = 0
    X, Y, Z = p
    if Z == _ZERO:
        return p
    A, B = _sqr(X), _sqr(Y)
    C = _sqr(B)
    D = _small(_sub(_sub(_sqr(_add(X, B)), A), C), 2)
    E = _small(A, 3)
    X3 = _sub(_sqr(E), _small(D, 2))
    Y3 = _sub(_mul(E, _sub(D, X3)), _small(C, 8))
    return (X3, Y3, _small(_mul(Y, Z), 2))


def _add_affine(p, q):  # Jacobian + affine
    X1, Y1, Z1 = p
    if Z1 == _ZERO:
        return (q[0], q[1], _ONE)
    Z1Z1 = _sqr(Z1)
    U2, S2 = _mul(q[0], Z1Z1), _mul(q[1], _mul(Z1, Z1Z1))
    H, r = _sub(U2, X1), _sub(S2, Y1)
    if H == _ZERO:
        return _dbl(p) if r == _ZERO else (_ONE, _ONE, _ZERO)
    HH = _sqr(H)
    HHH = _mul(H, HH)
    V = _mul(X1, HH)
    X3 = _sub(_sub(_sqr(r), HHH), _small(V, 2))
    Y3 = _sub(_mul(r, _sub(V, X3)), _mul(Y1, HHH))
    return (X3, Y3, _mul(Z1, H))


def scalar_mul(k: int, point=G2_GENERATOR):
    """k * point as an affine pair of Fq2 elements, or None for the identity."""
    k %= R
    acc = (_ONE, _ONE, _ZERO)
    for bit in bin(k)[2:] if k else "":
        acc = _dbl(acc)
        if bit == "1":
            acc = _add_affine(acc, point)
    if acc[2] == _ZERO:
        return None
    zi = _inv(acc[2])
    zi2 = _sqr(zi)
    return (_mul(acc[0], zi2), _mul(acc[1], _mul(zi, zi2)))


def to_bytes(point) -> bytes:
    if point is None:
        out = bytearray(64)
        out[63] |= 0x80
        return bytes(out)
    (x0, x1), (y0, _) = point
    out = bytearray(x0.to_bytes(32, "little") + x1.to_bytes(32, "little"))
    if y0 & 1:
        out[63] |= 0x40
    return bytes(out)
