"""Mirror of halo2_proofs::poly::kzg::commitment::ParamsKZG<Bn256>.

Reference call sites: ParamsKZG::setup (examples/standard_plonk.rs:29), gen_srs (src/scaffold.rs:119,
174,271).  `setup(k, s)` takes the toxic-waste scalar explicitly (the reference draws it from OsRng);
g[i] = s^i * G, g_lagrange[i] = L_i(s) * G with L_i(s) obtained as the inverse NTT of the powers of s —
all computed by the device kernels and kept registered in HBM.  commit / commit_lagrange are
best_multiexp against the matching base set (KZG ignores the blind).
"""
import ctypes as C
import struct

import numpy as np

from . import field as F
from . import g2 as G2
from . import serde
from ._lib import check, lib
from .device import DevBuf


def chacha20_block(key_words, counter: int, stream: int = 0):
    """one 64-byte block of ChaCha20 (20 rounds; 64-bit block counter, 64-bit stream id: the layout rand_chacha uses) as 16
    little-endian u32 words"""
    M = 0xFFFFFFFF
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & M, counter >> 32, stream & M, stream >> 32]
    x = list(st)
    rotl = lambda v, c: ((v << c) & M) | (v >> (32 - c))

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M for a, b in zip(x, st)]


def gen_srs_secret() -> int:
    """the toxic-waste scalar of the SRS the scaffold generates: `gen_srs(k)` (src/scaffold.rs:119,174,271; halo2-base
    utils::fs) calls ParamsKZG::setup(k, ChaCha20Rng::from_seed(Default::default())), whose first draw is
    s = Fr::random(rng) = from_u512 of eight next_u64 (the first 64 keystream bytes of the all-zero key, little-endian),
    reduced mod r [RECALL: halo2-base and halo2curves are not vendored; the keystream itself is the published ChaCha20
    zero-key vector].  With it `ParamsKZG.setup(k, gen_srs_secret())` is, as far as memory can settle, the very
    params/kzg_bn254_{k}.srs the reference caches — NOT a production SRS (the reference says so itself)."""
    words = chacha20_block([0] * 8, 0)
    wide = sum(w << (32 * i) for i, w in enumerate(words))
    return wide % F.FR_MODULUS


def gen_srs(k: int, params_dir: str = None) -> "ParamsKZG":
    """halo2-base `gen_srs` / `read_or_create_srs`: read {PARAMS_DIR or ./params}/kzg_bn254_{k}.srs if it exists, else
    set the SRS up from the fixed-seed rng above and write the file."""
    import os

    d = params_dir or os.environ.get("PARAMS_DIR", "./params")
    path = os.path.join(d, f"kzg_bn254_{k}.srs")
    if os.path.exists(path):
        with open(path, "rb") as f:
            return ParamsKZG.read(f)
    p = ParamsKZG.setup(k, gen_srs_secret())
    os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        p.write(f)
    return p


class ParamsKZG:
    def __init__(self, k: int):
        self.k = k
        self.n = 1 << k
        self.g_handle = None
        self.g_lagrange_handle = None
        self._g_dev = None
        self._gl_dev = None
        self.lo = 0             # first base of the slice this object commits against (multi-GPU: register_slice)
        self.g2_bytes = None    # the verifier's two G2 elements, carried as their 64-byte encodings
        self.s_g2_bytes = None

    @classmethod
    def setup(cls, k: int, s: int, register: bool = True) -> "ParamsKZG":
        """register=False: generate the SRS on the device but build no MSM tables (a multi-GPU rank registers only
        its slice, see register_slice)."""
        p = cls(k)
        n = p.n
        p.g2_bytes = G2.to_bytes(G2.G2_GENERATOR)
        p.s_g2_bytes = G2.to_bytes(G2.scalar_mul(s))
        s_m = F.fr_to_mont_limbs(s)
        pw = DevBuf(n * 32)
        check(lib.h2mi_fr_powers_dev(pw.ptr, n, s_m.ctypes.data, None), "powers")
        p._g_dev = DevBuf(n * 64)
        check(lib.h2mi_g1_fixed_base_mul_dev(pw.ptr, n, p._g_dev.ptr, None), "g")
        # L_i(s) = (1/n) sum_j s^j omega^(-ij): inverse NTT of the powers vector
        w_inv = F.fr_to_mont_limbs(F.fr_inv(F.omega_for(k)))
        n_inv = F.fr_to_mont_limbs(F.fr_inv(n))
        check(lib.h2mi_ntt_bn254_fr_dev(pw.ptr, k, w_inv.ctypes.data, None, n_inv.ctypes.data, None), "lagrange scalars")
        p._gl_dev = DevBuf(n * 64)
        check(lib.h2mi_g1_fixed_base_mul_dev(pw.ptr, n, p._gl_dev.ptr, None), "g_lagrange")
        check(lib.h2mi_sync(), "sync")
        pw.free()
        if register:
            p._register()
        return p

    def register_slice(self, lo: int, hi: int) -> "ParamsKZG":
        """ParamsKZG over bases [lo, hi) of both sets: the slice one rank of a sliced multi-GPU MSM owns
        (SURVEY.md 8e).  The window tables are built straight from this SRS's device-resident points."""
        assert 0 <= lo < hi <= self.n and self._gl_dev is not None
        p = ParamsKZG(self.k)
        p.n = hi - lo
        p.lo = lo
        p.g2_bytes, p.s_g2_bytes = self.g2_bytes, self.s_g2_bytes
        h = C.c_uint64()
        check(lib.h2mi_bases_register_dev(self._g_dev.ptr + lo * 64, p.n, C.byref(h)), "register g slice")
        p.g_handle = h.value
        h2 = C.c_uint64()
        check(lib.h2mi_bases_register_dev(self._gl_dev.ptr + lo * 64, p.n, C.byref(h2)), "register g_lagrange slice")
        p.g_lagrange_handle = h2.value
        return p

    @classmethod
    def from_bases(cls, k: int, g: np.ndarray, g_lagrange: np.ndarray = None) -> "ParamsKZG":
        p = cls(k)
        assert len(g) == p.n
        p._g_dev = DevBuf.from_numpy(np.ascontiguousarray(g, dtype=np.uint64))
        if g_lagrange is not None:
            p._gl_dev = DevBuf.from_numpy(np.ascontiguousarray(g_lagrange, dtype=np.uint64))
        p._register()
        return p

    @classmethod
    def from_monomial(cls, k: int, g: np.ndarray, g2_bytes: bytes = None, s_g2_bytes: bytes = None) -> "ParamsKZG":
        """the whole SRS from its monomial half, WITHOUT the secret — what ParamsKZG::setup does after its powers of s
        (`best_fft(&mut g_lagrange_projective, root.invert(), k)`, then n^-1; poly/kzg/commitment.rs [RECALL]): the group-valued inverse
        transform of g on the device (h2mi_fft_bn254_g1_dev).  For an SRS that arrives as points (a ceremony file, a fork's own
        `setup`); `setup(k, s)` above takes the shortcut through the exponents that knowing s allows."""
        p = cls(k)
        assert len(g) == p.n
        p.g2_bytes, p.s_g2_bytes = g2_bytes, s_g2_bytes
        p._g_dev = DevBuf.from_numpy(np.ascontiguousarray(g, dtype=np.uint64))
        p._gl_dev = DevBuf(p.n * 64)
        w_inv = F.fr_to_mont_limbs(F.fr_inv(F.omega_for(k)))
        n_inv = F.fr_to_mont_limbs(F.fr_inv(p.n))
        check(lib.h2mi_fft_bn254_g1_dev(p._g_dev.ptr, p._gl_dev.ptr, k, w_inv.ctypes.data, n_inv.ctypes.data, None), "group fft")
        p._register()
        return p

    def _register(self):
        h = C.c_uint64()
        check(lib.h2mi_bases_register_dev(self._g_dev.ptr, self.n, C.byref(h)), "register g")
        self.g_handle = h.value
        if self._gl_dev is not None:
            h2 = C.c_uint64()
            check(lib.h2mi_bases_register_dev(self._gl_dev.ptr, self.n, C.byref(h2)), "register g_lagrange")
            self.g_lagrange_handle = h2.value

    # ---- ParamsKZG::write / read: k (u32 LE), g, g_lagrange (32-byte compressed points), g2, s_g2 ----
    # [RECALL poly/kzg/commitment.rs of v2023_02_02; the file the scaffold caches as params/kzg_bn254_{k}.srs]
    def write(self, writer) -> None:
        if self.g2_bytes is None or self._gl_dev is None:
            raise ValueError("write needs the full SRS (g, g_lagrange, g2, s_g2)")
        writer.write(struct.pack("<I", self.k))
        tmp = DevBuf(self.n * 32)
        for dev in (self._g_dev, self._gl_dev):
            serde.g1_to_bytes_dev(dev, self.n, tmp)
            writer.write(tmp.to_numpy(shape=(self.n, 4)).tobytes())
        tmp.free()
        writer.write(self.g2_bytes)
        writer.write(self.s_g2_bytes)

    @classmethod
    def read(cls, reader) -> "ParamsKZG":
        head = reader.read(4)
        if len(head) != 4:
            raise serde.DecodeError("SRS file too short")
        (k,) = struct.unpack("<I", head)
        if not 1 <= k <= 26:
            raise serde.DecodeError(f"implausible k = {k} in SRS file")
        p = cls(k)
        n = p.n
        tmp = DevBuf(n * 32)
        devs = []
        for _ in range(2):
            raw = reader.read(n * 32)
            if len(raw) != n * 32:
                raise serde.DecodeError("SRS file too short")
            tmp.upload(np.frombuffer(raw, dtype=np.uint64))
            devs.append(serde.g1_from_bytes_dev(tmp, n))
        tmp.free()
        p._g_dev, p._gl_dev = devs
        p.g2_bytes, p.s_g2_bytes = reader.read(64), reader.read(64)
        if len(p.g2_bytes) != 64 or len(p.s_g2_bytes) != 64:
            raise serde.DecodeError("SRS file too short")
        p._register()
        return p

    def get_g(self) -> np.ndarray:
        return self._g_dev.to_numpy(shape=(self.n, 8))

    def get_g_lagrange(self) -> np.ndarray:
        return self._gl_dev.to_numpy(shape=(self.n, 8))

    def _commit(self, handle, poly: np.ndarray) -> np.ndarray:
        poly = np.ascontiguousarray(poly, dtype=np.uint64)
        if len(poly) > self.n:
            raise AssertionError("polynomial longer than the SRS")
        out = np.zeros(12, dtype=np.uint64)
        check(lib.h2mi_msm_bn254_g1(handle, None, poly.ctypes.data, len(poly), out.ctypes.data), "commit")
        return out

    def commit(self, poly_coeff: np.ndarray) -> np.ndarray:
        return self._commit(self.g_handle, poly_coeff)

    def commit_lagrange(self, poly_evals: np.ndarray) -> np.ndarray:
        return self._commit(self.g_lagrange_handle, poly_evals)

    def commit_dev(self, d_poly: DevBuf, d_out: DevBuf, n=None, lagrange=False, stream=None, out_offset=0):
        h = self.g_lagrange_handle if lagrange else self.g_handle
        check(lib.h2mi_msm_bn254_g1_dev(h, d_poly.ptr, self.n if n is None else n, d_out.ptr + out_offset, stream), "commit_dev")

    def release(self):
        for h in (self.g_handle, self.g_lagrange_handle):
            if h:
                lib.h2mi_bases_release(h)
        self.g_handle = self.g_lagrange_handle = None
        for b in (self._g_dev, self._gl_dev):
            if b is not None:
                b.free()
