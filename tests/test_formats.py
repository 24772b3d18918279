"""Wire formats around the hot path: encodings, Blake2b transcript, SRS file.
CPU part: the oracle's own invariants and the host-side product code that needs no GPU (transcript hashing,
G2).  GPU part: the device (de)compression kernels and ParamsKZG.write / read against the oracle."""
import ctypes as C
import hashlib
import io
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import bn254 as o
from oracle import formats as fm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- CPU: oracle invariants
def test_oracle_g1_encoding_round_trip_and_rejections():
    pts = [o.g1_mul(k, o.G1_GEN) for k in (1, 2, 3, 0xABCDEF, o.R - 1)] + [None]
    for p in pts:
        b = fm.g1_to_bytes(p)
        assert len(b) == 32 and fm.g1_from_bytes(b) == p
    assert fm.g1_to_bytes(o.G1_GEN) == (1).to_bytes(32, "little")           # y = 2 is even: no flag
    neg = (1, o.Q - 2)
    assert fm.g1_to_bytes(neg)[31] == 0x40 and fm.g1_from_bytes(fm.g1_to_bytes(neg)) == neg
    x = 2
    while pow((x ** 3 + 3) % o.Q, (o.Q - 1) // 2, o.Q) == 1:
        x += 1
    bad = [
        (o.Q).to_bytes(32, "little"),                   # x >= q
        x.to_bytes(32, "little"),                       # x^3 + 3 is a non-residue
        bytes(31) + bytes([0xC0]),                      # infinity + sign
        (1).to_bytes(31, "little") + bytes([0x80]),     # infinity with x != 0
    ]
    for b in bad:
        with pytest.raises(ValueError):
            fm.g1_from_bytes(b)


def test_oracle_g2_generator_and_srs_layout():
    assert fm.g2_on_curve(fm.G2_GEN)
    assert fm.g2_mul(o.R) is None and fm.g2_on_curve(fm.g2_mul(0x1234567))
    k, s = 3, 0x5EED
    raw = fm.srs_bytes(k, s)
    n = 1 << k
    assert len(raw) == 4 + 2 * n * 32 + 128 and struct.unpack("<I", raw[:4])[0] == k
    g, gl = o.srs(k, s)
    assert [fm.g1_from_bytes(raw[4 + 32 * i: 36 + 32 * i]) for i in range(n)] == g
    assert [fm.g1_from_bytes(raw[4 + 32 * (n + i): 36 + 32 * (n + i)]) for i in range(n)] == gl
    assert raw[-128:-64] == fm.g2_to_bytes(fm.G2_GEN)


def test_oracle_transcript_is_plain_blake2b():
    """the oracle's transcript, recomputed byte by byte with hashlib: one update stream, cloned per challenge."""
    t = fm.Blake2bTranscript()
    p, s = o.g1_mul(5, o.G1_GEN), 0xDEADBEEF
    t.write_point(p)
    t.write_scalar(s)
    c1 = t.squeeze_challenge()
    c2 = t.squeeze_challenge()
    stream = b"\x01" + p[0].to_bytes(32, "little") + p[1].to_bytes(32, "little") + b"\x02" + s.to_bytes(32, "little") + b"\x00"
    h = hashlib.blake2b(stream, digest_size=64, person=b"Halo2-Transcript")
    assert c1 == int.from_bytes(h.digest(), "little") % o.R
    h.update(b"\x00")
    assert c2 == int.from_bytes(h.digest(), "little") % o.R and c1 != c2
    assert bytes(t.proof) == fm.g1_to_bytes(p) + s.to_bytes(32, "little")


# ---------------------------------------------------------------- CPU: product host code without a GPU
def test_host_transcript_hashing_matches_oracle(h2):
    from halo2_scaffold_amd import transcript as T

    tw, to = T.Blake2bWrite(), fm.Blake2bTranscript()
    pts = [o.g1_mul(k, o.G1_GEN) for k in (7, 99)]
    for p in pts:
        tw.common_point(o.pack_points([p])[0])
        to.common_point(p)
    for s in (0, 1, o.R - 1, 0x1234):
        tw.common_scalar(o.pack([s], o.R)[0])
        to.common_scalar(s)
        assert o.unpack(tw.squeeze_challenge().reshape(1, 4), o.R) == [to.squeeze_challenge()]
    with pytest.raises(ValueError):
        tw.common_point(np.zeros(8, dtype=np.uint64))


def test_host_g2_matches_oracle(h2):
    from halo2_scaffold_amd import g2

    assert g2.G2_GENERATOR == fm.G2_GEN
    for k in (1, 2, 3, 0xC0FFEE, o.R - 1):
        assert g2.scalar_mul(k) == fm.g2_mul(k)
        assert g2.to_bytes(g2.scalar_mul(k)) == fm.g2_to_bytes(fm.g2_mul(k))
    assert g2.scalar_mul(0) is None and g2.to_bytes(None) == fm.g2_to_bytes(None)


def test_cpp_blake2b_matches_hashlib(tmp_path):
    """the C++ host layer carries its own Blake2b (RFC 7693) for the transcript: pin it against hashlib."""
    exe = tmp_path / "blake2b_host"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "blake2b_host.cpp"), "-o", str(exe)],
                   check=True)
    rng = np.random.default_rng(5)
    for n in (0, 1, 63, 64, 127, 128, 129, 255, 256, 1000):
        msg = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        out = subprocess.run([str(exe), msg.hex()], check=True, capture_output=True, text=True).stdout.strip()
        assert out == hashlib.blake2b(msg, digest_size=64, person=b"Halo2-Transcript").hexdigest(), n


# ---------------------------------------------------------------- GPU
def _rand_points(n, seed):
    rng = np.random.default_rng(seed)
    ks = [int(x) for x in rng.integers(1, 1 << 62, n)]
    return [o.g1_mul(k, o.G1_GEN) for k in ks]


@pytest.mark.gpu
def test_g1_compress_decompress_vs_oracle(gpu):
    from halo2_scaffold_amd import serde

    pts = _rand_points(61, 1) + [None, o.G1_GEN, (1, o.Q - 2)]
    aff = o.pack_points(pts)
    enc = serde.g1_to_bytes(aff)
    assert [bytes(e) for e in enc] == [fm.g1_to_bytes(p) for p in pts]
    dec = serde.g1_from_bytes(enc)
    assert np.array_equal(dec, aff)
    # invalid encodings: counted, decoded as the identity, and raised by the host layer
    bad = [(o.Q).to_bytes(32, "little"), bytes(31) + bytes([0xC0]), (1).to_bytes(31, "little") + bytes([0x80])]
    x = 2
    while pow((x ** 3 + 3) % o.Q, (o.Q - 1) // 2, o.Q) == 1:
        x += 1
    bad.append(x.to_bytes(32, "little"))  # x^3 + 3 is a non-residue
    raw = np.frombuffer(b"".join(bad) + fm.g1_to_bytes(o.G1_GEN), dtype=np.uint8).reshape(-1, 32)
    out = np.zeros((len(raw), 8), dtype=np.uint64)
    cnt = C.c_uint64()
    assert gpu.lib.h2mi_g1_decompress(raw.ctypes.data, len(raw), out.ctypes.data, C.byref(cnt)) == 0
    assert cnt.value == 4 and not out[:4].any() and o.unpack_points(out[4:]) == [o.G1_GEN]
    with pytest.raises(serde.DecodeError):
        serde.g1_from_bytes(raw)
    for b in bad:
        with pytest.raises(ValueError):
            fm.g1_from_bytes(b)


@pytest.mark.gpu
def test_fe_repr_kernels(gpu):
    n = 1000
    vals = o.unpack(o.random_field_limbs(n, 42), o.R) + [0, 1, o.R - 1]
    mont = o.pack(vals, o.R)
    d_in, d_out = gpu.DevBuf.from_numpy(mont), gpu.DevBuf(len(vals) * 32)
    assert gpu.lib.h2mi_fe_to_repr_dev(1, d_in.ptr, len(vals), d_out.ptr, None) == 0
    rep = d_out.to_numpy(dtype=np.uint8, shape=(len(vals), 32))
    assert [bytes(r) for r in rep] == [fm.fe_to_repr(v) for v in vals]
    back, cnt = gpu.DevBuf(len(vals) * 32), C.c_uint64()
    assert gpu.lib.h2mi_fe_from_repr_dev(1, d_out.ptr, len(vals), back.ptr, C.byref(cnt)) == 0
    assert cnt.value == 0 and np.array_equal(back.to_numpy(shape=(len(vals), 4)), mont)
    # r itself and 2^256 - 1 are not canonical
    badv = np.frombuffer(o.R.to_bytes(32, "little") + b"\xff" * 32 + (5).to_bytes(32, "little"), dtype=np.uint64)
    d_bad = gpu.DevBuf.from_numpy(badv)
    assert gpu.lib.h2mi_fe_from_repr_dev(1, d_bad.ptr, 3, back.ptr, C.byref(cnt)) == 0
    assert cnt.value == 2 and o.unpack(back.to_numpy(shape=(len(vals), 4))[:3], o.R) == [0, 0, 5]


@pytest.mark.gpu
def test_params_write_read_match_oracle_srs_bytes(gpu):
    k, s = 5, 0x5EED5EED
    params = gpu.ParamsKZG.setup(k, s)
    buf = io.BytesIO()
    params.write(buf)
    assert buf.getvalue() == fm.srs_bytes(k, s)
    again = gpu.ParamsKZG.read(io.BytesIO(buf.getvalue()))
    assert again.k == k and np.array_equal(again.get_g(), params.get_g()) and np.array_equal(again.get_g_lagrange(), params.get_g_lagrange())
    assert (again.g2_bytes, again.s_g2_bytes) == (params.g2_bytes, params.s_g2_bytes)
    f = o.random_field_limbs(1 << k, 9)
    from oracle import cref

    assert np.array_equal(cref.normalize(again.commit(f)), cref.normalize(params.commit(f)))
    # a corrupted point (x replaced by a non-residue abscissa) and a truncated file are refused
    from halo2_scaffold_amd import serde

    raw = bytearray(buf.getvalue())
    x = 2
    while pow((x ** 3 + 3) % o.Q, (o.Q - 1) // 2, o.Q) == 1:
        x += 1
    raw[4 + 32 * 3: 4 + 32 * 4] = x.to_bytes(32, "little")
    with pytest.raises(serde.DecodeError):
        gpu.ParamsKZG.read(io.BytesIO(bytes(raw)))
    with pytest.raises(serde.DecodeError):
        gpu.ParamsKZG.read(io.BytesIO(buf.getvalue()[:-1]))
    params.release()
    again.release()


@pytest.mark.gpu
def test_transcript_write_read_round_trip_matches_oracle(gpu):
    """a proof-shaped sequence (commitments, challenges, evaluations) through Blake2bWrite, read back with
    Blake2bRead: same bytes and same challenges as the oracle transcript."""
    from halo2_scaffold_amd import transcript as T

    pts = _rand_points(6, 3)
    evals = o.unpack(o.random_field_limbs(5, 4), o.R)
    tw, to = T.Blake2bWrite.init(), fm.Blake2bTranscript()
    chal_w, chal_o = [], []
    for p in pts[:3]:
        tw.write_point(o.pack_points([p])[0]); to.write_point(p)
    chal_w.append(tw.squeeze_challenge()); chal_o.append(to.squeeze_challenge())
    for p in pts[3:]:
        tw.write_point(o.pack_points([p])[0]); to.write_point(p)
    chal_w.append(tw.squeeze_challenge()); chal_o.append(to.squeeze_challenge())
    for e in evals:
        tw.write_scalar(o.pack([e], o.R)[0]); to.write_scalar(e)
    chal_w.append(tw.squeeze_challenge()); chal_o.append(to.squeeze_challenge())
    proof = tw.finalize()
    assert proof == bytes(to.proof) and len(proof) == 6 * 32 + 5 * 32
    assert [o.unpack(c.reshape(1, 4), o.R)[0] for c in chal_w] == chal_o
    tr = T.Blake2bRead.init(proof)
    got = [tr.read_point() for _ in range(3)]
    c0 = tr.squeeze_challenge()
    got += [tr.read_point() for _ in range(3)]
    c1 = tr.squeeze_challenge()
    sc = [tr.read_scalar() for _ in range(5)]
    c2 = tr.squeeze_challenge()
    assert o.unpack_points(np.stack(got)) == pts and o.unpack(np.stack(sc), o.R) == evals
    assert [o.unpack(c.reshape(1, 4), o.R)[0] for c in (c0, c1, c2)] == chal_o
    with pytest.raises(Exception):
        tr.read_scalar()


@pytest.mark.gpu
def test_cpp_transcript_and_srs_match_python_and_oracle(gpu, tmp_path):
    """the C++ host layer (h2mi_transcript.hpp, ParamsKZG::read / write) on values handed over in the crate's
    in-memory layouts: same proof bytes and challenges as the Python host and the oracle; an SRS file written
    by the Python host is read and re-written byte for byte."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    pts = _rand_points(5, 11)
    evals = o.unpack(o.random_field_limbs(4, 12), o.R)
    blob = struct.pack("<II", len(pts), len(evals)) + o.pack_points(pts).tobytes() + o.pack(evals, o.R).tobytes()
    (tmp_path / "values.bin").write_bytes(blob)
    k, s = 6, 0xABCDEF123
    params = gpu.ParamsKZG.setup(k, s)
    with open(tmp_path / "in.srs", "wb") as f:
        params.write(f)
    params.release()
    assert (tmp_path / "in.srs").read_bytes() == fm.srs_bytes(k, s)
    r = subprocess.run([os.path.join(ROOT, "examples", "transcript_srs"), str(tmp_path / "values.bin"), str(tmp_path / "in.srs"), str(tmp_path / "out.srs")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.startswith(("proof", "challenge")))
    t = fm.Blake2bTranscript()
    for p in pts:
        t.write_point(p)
    c1 = t.squeeze_challenge()
    for e in evals:
        t.write_scalar(e)
    c2 = t.squeeze_challenge()
    assert bytes.fromhex(lines["proof"]) == bytes(t.proof)
    got = [o.unpack(np.frombuffer(bytes.fromhex(lines[n]), dtype=np.uint64).reshape(1, 4), o.R)[0] for n in ("challenge1", "challenge2")]
    assert got == [c1, c2]
    assert "transcript round trip ok" in r.stdout and f"srs k={k} commit(all ones) identity=0" in r.stdout
    assert (tmp_path / "out.srs").read_bytes() == (tmp_path / "in.srs").read_bytes()


# ---------------------------------------------------------------- committed golden vectors (tests/golden)
def _gold():
    import json

    return json.load(open(os.path.join(ROOT, "tests", "golden", "bn254_vectors.json")))["formats"]


def test_oracle_formats_match_golden():
    g = _gold()
    pts = [tuple(int(c, 16) for c in p) for p in g["points"]]
    assert [fm.g1_to_bytes(p).hex() for p in pts] + [fm.g1_to_bytes(None).hex()] == g["compressed"]
    t = fm.Blake2bTranscript()
    for p in pts:
        t.write_point(p)
    ch = [t.squeeze_challenge()]
    for v in g["scalars"]:
        t.write_scalar(int(v, 16))
    ch += [t.squeeze_challenge(), t.squeeze_challenge()]
    assert bytes(t.proof).hex() == g["transcript_proof"] and ch == [int(c, 16) for c in g["transcript_challenges"]]
    raw = fm.srs_bytes(3, 5)
    assert hashlib.sha256(raw).hexdigest() == g["srs_k3_s5_sha256"] and raw[-64:].hex() == g["srs_k3_s5_tail"]
    assert fm.g2_to_bytes(fm.G2_GEN).hex() == g["g2_generator"]


@pytest.mark.gpu
def test_device_formats_match_golden(gpu):
    from halo2_scaffold_amd import serde
    from halo2_scaffold_amd import transcript as T

    g = _gold()
    pts = [tuple(int(c, 16) for c in p) for p in g["points"]]
    aff = o.pack_points(pts + [None])
    assert [bytes(e).hex() for e in serde.g1_to_bytes(aff)] == g["compressed"]
    assert np.array_equal(serde.g1_from_bytes(bytes.fromhex("".join(g["compressed"]))), aff)
    tw = T.Blake2bWrite()
    for row in aff[:-1]:
        tw.write_point(row)
    ch = [tw.squeeze_challenge()]
    for v in g["scalars"]:
        tw.write_scalar(o.pack([int(v, 16)], o.R)[0])
    ch += [tw.squeeze_challenge(), tw.squeeze_challenge()]
    assert tw.finalize().hex() == g["transcript_proof"]
    assert [o.unpack(c.reshape(1, 4), o.R)[0] for c in ch] == [int(c, 16) for c in g["transcript_challenges"]]
    params = gpu.ParamsKZG.setup(3, 5)
    buf = io.BytesIO()
    params.write(buf)
    assert hashlib.sha256(buf.getvalue()).hexdigest() == g["srs_k3_s5_sha256"]
    params.release()


@pytest.mark.gpu
def test_gen_srs_reads_or_creates_the_params_file(gpu, tmp_path):
    """gen_srs(k) as the scaffold calls it (src/scaffold.rs:119,174,271): creates params/kzg_bn254_{k}.srs from the
    fixed-seed secret on first use, reads it back afterwards; both objects commit identically and the file's points
    are s^i G for that secret"""
    k = 6
    first = gpu.gen_srs(k, str(tmp_path))
    path = tmp_path / f"kzg_bn254_{k}.srs"
    assert path.exists() and path.stat().st_size == 4 + 2 * 32 * (1 << k) + 2 * 64
    again = gpu.gen_srs(k, str(tmp_path))
    s = gpu.gen_srs_secret()
    g = o.unpack_points(first.get_g())
    assert g[0] == o.G1_GEN and g[1] == o.g1_mul(s, o.G1_GEN) and g[5] == o.g1_mul(pow(s, 5, o.R), o.G1_GEN)
    assert o.unpack_points(again.get_g()) == g and o.unpack_points(again.get_g_lagrange()) == o.unpack_points(first.get_g_lagrange())
    poly = o.random_field_limbs(1 << k, 7)
    assert o.unpack_jacobian(first.commit(poly)) == o.unpack_jacobian(again.commit(poly))
    first.release()
    again.release()
