"""CPU oracle: the Poseidon permutation and sponge the reference's poseidon example hashes with
(examples/poseidon.rs:10-13,27-32: T = 3, RATE = 2, R_F = 8, R_P = 57 over bn256::Fr; `PoseidonChip::new(ctx, R_F, R_P)`,
`update(&[x, y])`, `squeeze`), in plain integer arithmetic.

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  The chip is Axiom's (Cargo.toml:17-18, adapted from Scroll / snark-verifier);
it is not vendored, so this restates the published construction:
  * round constants and MDS matrix from the Grain LFSR of the Poseidon paper's reference script
    (generate_parameters_grain.sage): 80-bit state = field type (2 bits: 1), s-box (4 bits: 0 = x^alpha), field size
    (12 bits: 254), t (12), R_F (10), R_P (10), thirty ones; 160 warm-up clocks; output bits in pairs (first bit 1: emit
    the second, else drop it); constants by rejection sampling of 254-bit big-endian integers, then the Cauchy matrix
    M[i][j] = 1 / (x_i + y_j) from 2t further samples reduced mod r;
  * permutation: R_F / 2 full rounds, R_P partial rounds, R_F / 2 full rounds of (add round constants, x^5, multiply by M);
  * sponge [RECALL snark-verifier `Poseidon`]: state (2^64, 0, 0); inputs absorbed RATE at a time by addition into
    state[1..]; a chunk shorter than RATE (including the empty chunk that follows an exact multiple) adds 1 to the next
    free position; squeeze returns state[1].
PINNED (the first two items): tests/test_oracle_poseidon.py checks the generated constants and matrix against the
values published in circomlib's poseidon_constants (same script, same parameters) and the permutation against
circomlib's known answer poseidon([1, 2]); the sponge convention itself is [RECALL]: parity unpinned.
"""
from __future__ import annotations

from . import bn254 as o

R = o.R


class Grain:
    def __init__(self, t: int, r_f: int, r_p: int, field_bits: int = 254):
        bits = []
        for value, width in ((1, 2), (0, 4), (field_bits, 12), (t, 12), (r_f, 10), (r_p, 10), ((1 << 30) - 1, 30)):
            bits += [(value >> (width - 1 - i)) & 1 for i in range(width)]
        assert len(bits) == 80
        self.state = bits
        self.field_bits = field_bits
        for _ in range(160):
            self._clock()

    def _clock(self) -> int:
        s = self.state
        new = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(new)
        return new

    def next_bit(self) -> int:
        while True:
            first = self._clock()
            second = self._clock()
            if first:
                return second

    def next_int(self) -> int:
        v = 0
        for _ in range(self.field_bits):
            v = (v << 1) | self.next_bit()
        return v

    def next_field_element(self) -> int:  # rejection sampling
        while True:
            v = self.next_int()
            if v < R:
                return v

    def next_field_element_without_rejection(self) -> int:
        return self.next_int() % R


def generate(t: int = 3, r_f: int = 8, r_p: int = 57):
    """-> (round constants as (r_f + r_p) rows of t, MDS matrix t x t)"""
    g = Grain(t, r_f, r_p)
    constants = [[g.next_field_element() for _ in range(t)] for _ in range(r_f + r_p)]
    while True:
        xs = [g.next_field_element_without_rejection() for _ in range(t)]
        ys = [g.next_field_element_without_rejection() for _ in range(t)]
        if len(set(xs + ys)) == 2 * t and all((x + y) % R for x in xs for y in ys):
            break
    mds = [[pow((x + y) % R, -1, R) for y in ys] for x in xs]
    return constants, mds


_CACHE = {}


def params(t=3, r_f=8, r_p=57):
    key = (t, r_f, r_p)
    if key not in _CACHE:
        _CACHE[key] = generate(t, r_f, r_p)
    return _CACHE[key]


def permute(state, t=3, r_f=8, r_p=57):
    constants, mds = params(t, r_f, r_p)
    s = [v % R for v in state]
    for rnd in range(r_f + r_p):
        s = [(v + c) % R for v, c in zip(s, constants[rnd])]
        if rnd < r_f // 2 or rnd >= r_f // 2 + r_p:
            s = [pow(v, 5, R) for v in s]
        else:
            s[0] = pow(s[0], 5, R)
        s = [sum(m * v for m, v in zip(row, s)) % R for row in mds]
    return s


def circomlib_hash(inputs):
    """circomlib's fixed-width hash: state (0, inputs...), one permutation, output state[0]"""
    return permute([0] + list(inputs), t=len(inputs) + 1)[0]


def sponge_hash(inputs, t=3, rate=2, r_f=8, r_p=57):
    """update(inputs) then squeeze() of the chip the reference uses [RECALL, see header]"""
    state = [1 << 64] + [0] * (t - 1)
    chunks = [inputs[i : i + rate] for i in range(0, len(inputs), rate)]
    if len(inputs) % rate == 0:
        chunks.append([])
    for chunk in chunks:
        for i, v in enumerate(chunk):
            state[1 + i] = (state[1 + i] + v) % R
        if len(chunk) < rate:
            state[1 + len(chunk)] = (state[1 + len(chunk)] + 1) % R
        state = permute(state, t, r_f, r_p)
    return state[1]
