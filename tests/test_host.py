"""CPU tests of the host layer: the C-ABI library loads and exports every declared symbol, argument
checking happens before any device work, the synthetic generators agree with the oracle's, and the
N > 1 combine path works over gloo with world_size 2.  No compute call is issued without a GPU."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import bn254 as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(h2):
    hdr = open(os.path.join(ROOT, "include", "h2mi.h")).read() + open(os.path.join(ROOT, "include", "h2mi_prover.h")).read()  # every C header
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # prose in comments names callbacks and macros, not exports
    declared = set(re.findall(r"\b(h2mi_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25 and "h2mi_prover_quotient" in declared
    for name in sorted(declared):
        assert hasattr(h2.lib, name), f"libh2mi.so does not export {name}"
    assert set(h2.lib._h2mi_symbols) == declared  # the ctypes binding covers the whole header
    assert b"gfx950" in h2.lib.h2mi_version()


def test_no_cpu_fallback_without_gpu(h2):
    """every compute entry point refuses to run when no device context exists (never a silent CPU path)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    assert h2.lib.h2mi_init(0) == -2
    a = np.zeros((4, 4), dtype=np.uint64)
    w = np.zeros(4, dtype=np.uint64)
    assert h2.lib.h2mi_ntt_bn254_fr(a.ctypes.data, w.ctypes.data, 2) == -2
    out = np.zeros(12, dtype=np.uint64)
    assert h2.lib.h2mi_msm_bn254_g1(0, np.zeros((4, 8), dtype=np.uint64).ctypes.data, a.ctypes.data, 4, out.ctypes.data) == -2
    with pytest.raises(h2.H2miError):
        h2.best_fft(a, w, 2)
    assert b"no CPU fallback" in h2.lib.h2mi_strerror(-2)


def test_prover_abi_refuses_without_gpu(h2):
    """level B (include/h2mi_prover.h) has no CPU fallback either: keygen and prover creation return H2MI_ENODEV before looking at
    anything else, and a handle that was never issued is H2MI_EHANDLE for every phase."""
    import ctypes as C

    import torch

    from halo2_scaffold_amd import circuits, engine, keygen

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_prover_abi.py")
    cs = keygen.constraint_system(circuits.StandardPlonk, 5)
    cells, keep = engine.pack_cells([{} for _ in range(5)])
    out = C.c_void_p()
    assert h2.lib.h2mi_prover_keygen(C.byref(cs), 1, cells, None, 0, 0, C.byref(out)) == -2 and not out.value
    assert h2.lib.h2mi_prover_create(None, 1, 2, 0, 32, C.byref(out)) == -2
    assert h2.lib.h2mi_prover_keygen(None, 1, cells, None, 0, 0, C.byref(out)) == -1  # argument checks come first
    bogus = C.c_void_p(0x1234)
    pts = np.zeros((8, 8), dtype=np.uint64)
    one = np.array([1, 0, 0, 0], dtype=np.uint64)
    assert h2.lib.h2mi_prover_advice(bogus, cells, None, 0, 1, pts.ctypes.data) == -5
    assert h2.lib.h2mi_prover_quotient(bogus, one.ctypes.data, pts.ctypes.data) == -5
    assert h2.lib.h2mi_prover_destroy(bogus) == -5 and h2.lib.h2mi_prover_pk_release(bogus) == -5
    assert b"not satisfied" in h2.lib.h2mi_strerror(-7)


def test_multi_device_mode_needs_devices(h2):
    """h2mi_init_devices (one process, n GPUs) fails like h2mi_init without a GPU and leaves the library uninitialised;
    the slice partition it applies is the one dist.slice_bounds states (contiguous, sizes differing by at most one)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_multidevice.py")
    assert h2.lib.h2mi_init_devices(4) == -2 and h2.lib.h2mi_device_count() == 0
    assert h2.lib.h2mi_init_devices(0) == -1 and h2.lib.h2mi_init_devices(17) == -1


def test_argument_checks_mirror_reference_asserts(h2):
    with pytest.raises(AssertionError):  # assert_eq!(coeffs.len(), bases.len())
        h2.best_multiexp(np.zeros((3, 4), dtype=np.uint64), np.zeros((4, 8), dtype=np.uint64))
    with pytest.raises(AssertionError):  # assert_eq!(n, 1 << log_n)
        h2.best_fft(np.zeros((3, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64), 2)
    with pytest.raises(ValueError):
        h2.best_multiexp(np.zeros((3, 5), dtype=np.uint64), np.zeros((3, 8), dtype=np.uint64))


def test_domain_constants_match_oracle(h2):
    for k, j in [(5, 3), (8, 4), (20, 3), (22, 5)]:
        d, od = h2.EvaluationDomain(j, k), o.Domain(k, j)
        assert (d.extended_k, d.omega, d.extended_omega, d.g_coset, d.g_coset_inv) == (
            od.extended_k, od.omega, od.extended_omega, od.g_coset, od.g_coset_inv)
        assert d.ifft_divisor == od.ifft_divisor and d.extended_ifft_divisor == od.extended_ifft_divisor
        assert o.unpack(d._omega.reshape(1, 4), o.R) == [od.omega]
    from halo2_scaffold_amd import field as F

    assert F.FR_ROOT_OF_UNITY == o.FR_ROOT_OF_UNITY and F.FR_ZETA == o.FR_ZETA and F.FR_MODULUS == o.R


def test_synth_matches_oracle_generator(h2):
    from halo2_scaffold_amd import synth

    assert np.array_equal(synth.uniform_fr(4096, o.SEED), o.random_field_limbs(4096, o.SEED))
    assert np.array_equal(synth.uniform_fr(100, 7, start=50), o.random_field_limbs(100, 7, start=50))
    assert np.array_equal(synth.witness_like_fr(4096, o.SEED), o.witness_like_limbs(4096, o.SEED))


def test_replay_shape_matches_reference_circuit(h2):
    """3 advice + 3 permutation products + random + (degree-1) h pieces + 2 SHPLONK = 11 MSMs."""
    from halo2_scaffold_amd import replay

    assert replay.MSM_PER_PROOF == 11
    assert replay.NTT_PER_PROOF == {"intt_n": 6, "coset_ntt_ext": 6, "coset_intt_ext": 1}
    sp = replay.STANDARD_PLONK
    assert (sp.n_advice, sp.n_perm_z, sp.cs_degree, sp.n_lookups) == (3, 3, 3, 0)
    # halo2-lib gate circuit: 1 advice + 1 instance, 3 equality columns in chunks of 1, no lookups
    g = replay.HALO2_LIB_GATE
    assert g.msm_per_proof == 1 + 3 + 1 + 2 + 2 and g.ntt_per_proof["intt_n"] == 1 + 1 + 3
    # range circuit, single advice column: the lookup of q_lookup * a => degree 5: one permutation set of 3, 4 h pieces
    r = replay.RANGE_LOOKUP
    assert r.n_perm_z == 1 and r.msm_per_proof == 1 + 2 + 1 + 1 + 1 + 4 + 2


def test_slice_bounds(h2):
    from halo2_scaffold_amd.dist import slice_bounds

    for n in (1 << 20, 1000, 7):
        for world in (1, 2, 3, 8):
            b = [slice_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import _load_pkg
h2 = _load_pkg.load()
from halo2_scaffold_amd.dist import PartialPointCombiner, slice_bounds
from oracle import bn254 as o, cref
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, slots = 96, 3
bases = cref.g1_mul_gen(o.random_field_limbs(n, 5), 1)
lo, hi = slice_bounds(n, rank, world)
part = np.stack([cref.msm(o.random_field_limbs(n, 100 + s)[lo:hi], bases[lo:hi], 1) for s in range(slots)])
comb = PartialPointCombiner(fold=lambda allp: np.stack([cref.g1_sum(allp[:, i]) for i in range(allp.shape[1])]))        # CPU fold injected: exercises the collective plumbing
total = comb(part)
for s in range(slots):
    want = o.unpack_jacobian(cref.msm(o.random_field_limbs(n, 100 + s), bases, 1))
    assert o.unpack_jacobian(total[s]) == want, (rank, s)
dist.barrier()
dist.destroy_process_group()
open(os.path.join({outdir!r}, "rank%d.ok" % rank), "w").write("ok")
"""


def test_sliced_msm_combine_gloo_world2(tmp_path):
    """world_size 2 over gloo: each rank computes its slice's partial points, all-gather + fold == full MSM."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, outdir=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


_PHASE_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import _load_pkg
h2 = _load_pkg.load()
from halo2_scaffold_amd.dist import PhaseCombiner, slice_bounds
from halo2_scaffold_amd import replay
from oracle import bn254 as o, cref
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 64
slots = replay.STANDARD_PLONK.msm_per_proof                      # 11 commitments ...
phases = [3, 4, 2, 1, 1]                                         # ... in the five transcript phases of create_proof
assert sum(phases) == slots
bases = cref.g1_mul_gen(o.random_field_limbs(n, 5), 1)
lo, hi = slice_bounds(n, rank, world)
partial = np.zeros((slots, 12), dtype=np.uint64)
combined = np.zeros((slots, 12), dtype=np.uint64)
fold = lambda allp: np.stack([cref.g1_sum(allp[:, i]) for i in range(allp.shape[1])])
pc = PhaseCombiner(slots, "gloo", host_arrays=(partial, combined, fold))
first = 0
for cnt in phases:
    for s in range(first, first + cnt):                          # the phase's MSMs on this rank's slice
        partial[s] = cref.msm(o.random_field_limbs(n, 100 + s)[lo:hi], bases[lo:hi], 1)
    pc.combine(first, cnt)                                       # the transcript join: all-gather + fold of THIS phase
    for s in range(first, first + cnt):                          # full commitments exist before the next phase starts
        want = o.unpack_jacobian(cref.msm(o.random_field_limbs(n, 100 + s), bases, 1))
        assert o.unpack_jacobian(pc.result()[s]) == want, (rank, s)
    assert not pc.result()[first + cnt:].any()                   # nothing of a later phase has been touched
    first += cnt
assert pc.combines == 5
dist.barrier()
dist.destroy_process_group()
open(os.path.join({outdir!r}, "rank%d.ok" % rank), "w").write("ok")
"""


def test_per_phase_combine_gloo_world2(tmp_path):
    """world_size 2 over gloo: the partial points are combined at every transcript join (five per StandardPlonk
    proof), and after each join the phase's commitments equal the single-process MSMs."""
    script = tmp_path / "worker.py"
    script.write_text(_PHASE_WORKER.format(root=ROOT, outdir=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_cpp_host_example_builds_and_fails_loudly_without_gpu(h2):
    """examples/standard_plonk.cpp (the C++ mirror of the reference example over include/h2mi.hpp) links against
    libh2mi.so; without a GPU it must exit with the library's error, never compute on the CPU."""
    import torch

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    exe = os.path.join(ROOT, "examples", "standard_plonk")
    assert os.path.exists(exe)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the example is run by the -m gpu tests")
    r = subprocess.run([exe, "5"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr


def test_gen_srs_secret_from_the_chacha20_zero_seed(h2):
    """params.gen_srs_secret: halo2-base's gen_srs seeds ChaCha20Rng with zeros; the keystream block is the published
    ChaCha20 zero-key / zero-nonce vector (djb's reference test vector, also RFC 7539 A.1 #1), and the scalar is its first 64
    bytes read as a little-endian integer mod r (Fr::from_u512)"""
    from halo2_scaffold_amd import params as P

    ks = b"".join(w.to_bytes(4, "little") for w in P.chacha20_block([0] * 8, 0))
    assert ks.hex() == ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                        "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    s = P.gen_srs_secret()
    assert s == int.from_bytes(ks, "little") % o.R and 0 < s < o.R
    assert P.chacha20_block([0] * 8, 1) != P.chacha20_block([0] * 8, 0)


_LAUNCH_PROBE = r"""
import json, subprocess, sys
sys.path.insert(0, {root!r})
seen = {{}}
class FakePopen:
    def __init__(self, cmd, **kw):
        # the moment of the spawn: nothing that could hold a GPU context may exist in the parent yet
        seen["cmd"] = cmd
        seen["env_ipc"] = kw.get("env", {{}}).get("HSA_ENABLE_IPC_MODE_LEGACY")
        seen["torch_loaded"] = "torch" in sys.modules
        seen["library_loaded"] = "halo2_scaffold_amd" in sys.modules
        self.stdout = iter(['chatter from a rank\n', '{{"metric": "m", "n_gpus": 2}}\n'])
    def wait(self):
        return 7
subprocess.Popen = FakePopen
import bench
try:
    bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1", "--k", "13"])
except SystemExit as e:
    seen["exit"] = e.code
sys.stderr.write("PROBE " + json.dumps(seen) + "\n")
"""


def test_bench_plain_multi_gpu_launch_spawns_before_any_gpu_call(tmp_path):
    """`python bench.py --gpus N` without WORLD_SIZE (the form the driver's N = 1 command has): the parent must start
    `torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process before torch or the library is even
    imported (a process that has touched the GPU must not exec on this pool, and must not idle on a GPU context), relay rank 0's JSON
    line on stdout and pass the child's exit code on."""
    import json

    script = tmp_path / "probe.py"
    script.write_text(_LAUNCH_PROBE.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    probe = json.loads(next(l for l in r.stderr.splitlines() if l.startswith("PROBE "))[6:])
    cmd = probe["cmd"]
    assert probe["torch_loaded"] is False and probe["library_loaded"] is False
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--k", "13"]
    assert probe["env_ipc"] == "0"
    assert probe["exit"] == 7                                    # the child's exit code, relayed
    assert r.stdout.strip() == '{"metric": "m", "n_gpus": 2}'    # ONE line on stdout: rank 0's
    assert "chatter from a rank" in r.stderr


def test_bench_plain_multi_gpu_launch_end_to_end_without_gpu():
    """the real thing on this GPU-less box: the plain form starts two ranks over gloo, each fails loudly at its first device call (no CPU
    fallback), and the parent reports the failure through its exit code instead of printing a line."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_gpu_parity.py::test_bench_world2_rehearsal_matches_single_gpu runs the plain form for real")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(H2MI_DIST_BACKEND="gloo", H2MI_DEVICE="0", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--k", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "starting" in r.stderr and "torch.distributed.run" in r.stderr
    assert "No HIP GPUs are available" in r.stderr or "no usable GPU" in r.stderr  # the ranks' own failure, relayed on stderr


def test_bench_honours_a_preset_world_size_and_rejects_mismatch():
    """under torch.distributed.run (WORLD_SIZE set) bench.py must not start a second launcher; a WORLD_SIZE that disagrees with
    --gpus is a launch mistake and exits with a message."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in r.stderr and "starting" not in r.stderr


def test_cpp_host_single_element_inversion_matches_pow():
    """include/h2mi.hpp detail::inv_mod_odd (division steps in batches of 62: what fr::invert / fq::invert of the C++ host use on the
    transcript's critical path) against Python's pow(x, -1, p), edge values and random ones, both fields, plain and Montgomery forms."""
    import ctypes as C

    src = os.path.join(ROOT, "tests", "host", "inv_host.cpp")
    so = os.path.join(ROOT, "tests", "host", "libinvhost.so")
    deps = [src, os.path.join(ROOT, "include", "h2mi.hpp"), os.path.join(ROOT, "halo2-scaffold_amd", "csrc", "h2mi_hostmath.hpp"),
            os.path.join(ROOT, "halo2-scaffold_amd", "csrc", "inv_divsteps.cuh")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    L = C.CDLL(so)
    L.h2t_inv_plain.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.h2t_inv_plain32.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.h2t_inv_mont.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(0x1217)
    for field, mod in ((0, o.Q), (1, o.R)):
        vals = [1, 2, 3, mod - 1, mod - 2, (mod + 1) // 2, (mod - 1) // 2, 1 << 62, (1 << 62) - 1, 1 << 124, 1 << 253, (1 << 253) + 1,
                (1 << 64) - 1, 1 << 64, (1 << 128) - 1, (1 << 192) + 1, pow(2, -1, mod), pow(3, -1, mod), pow(2, -62, mod), pow(2, -124, mod)]
        vals += [int.from_bytes(rng.bytes(32), "little") % (mod - 1) + 1 for _ in range(20000)]
        vals += [int(v) for v in rng.integers(1, 1 << 20, size=2000)]  # small values: f and g of very different sizes
        n = len(vals)
        A = np.array([[(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)] for v in vals], dtype=np.uint64)
        out = np.zeros_like(A)
        ok = np.zeros(n, dtype=np.uint8)
        L.h2t_inv_plain(field, A.ctypes.data, out.ctypes.data, ok.ctypes.data, n)
        assert ok.all()  # the fast path itself, not the Fermat fallback behind it
        got = [sum(int(out[j, i]) << (64 * i) for i in range(4)) for j in range(n)]
        assert got == [pow(v, -1, mod) for v in vals]
        # the device's 32-bit form of the same steps (csrc/inv_divsteps.cuh: what the grand products' one inversion runs), on the host
        out32 = np.zeros_like(A)
        ok32 = np.zeros(n, dtype=np.uint8)
        L.h2t_inv_plain32(field, A.ctypes.data, out32.ctypes.data, ok32.ctypes.data, n)
        assert ok32.all() and np.array_equal(out32, out)
        # Montgomery forms through fr::invert / fq::invert, against the exponentiation they replace; zero stays zero
        M = o.pack([0] + vals[:3000], mod)
        fast, fermat = np.zeros_like(M), np.zeros_like(M)
        L.h2t_inv_mont(field, 0, M.ctypes.data, fast.ctypes.data, len(M))
        L.h2t_inv_mont(field, 1, M.ctypes.data, fermat.ctypes.data, len(M))
        assert np.array_equal(fast, fermat)
        assert o.unpack(fast, mod) == [0] + [pow(v, -1, mod) for v in vals[:3000]]


def _chacha20_block(key: bytes, counter: int, stream: int, ctr32_nonce96: bytes = None) -> bytes:
    """one 64-byte ChaCha20 block, written from RFC 7539 2.3: 64-bit block counter + 64-bit stream id in words 12 .. 15 (the layout
    rand_chacha uses), or — for the RFC's own test vector — a 32-bit counter and a 96-bit nonce"""
    M = 0xFFFFFFFF
    kw = [int.from_bytes(key[4 * i : 4 * i + 4], "little") for i in range(8)]
    if ctr32_nonce96 is not None:
        tail = [counter & M] + [int.from_bytes(ctr32_nonce96[4 * i : 4 * i + 4], "little") for i in range(3)]
    else:
        tail = [counter & M, (counter >> 32) & M, stream & M, (stream >> 32) & M]
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + kw + tail
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
    return b"".join(((a + b) & M).to_bytes(4, "little") for a, b in zip(x, st))


def test_prover_host_helpers_match_oracle():
    """what csrc/h2mi_prover.cpp computes on the CPU around its kernels (csrc/h2mi_hostmath.hpp), without a GPU: the copy
    constraints' cycle structure (permutation/keygen.rs Assembly::copy: merge the smaller cycle into the larger, swap two mapping
    entries) against oracle/plonk.py's dense Assembly on random constrain_equal sequences incl. repeats and self-copies;
    G1::batch_normalize of a phase's Jacobian points (one inversion) incl. the identity; the counter-based blinding stream; the
    canonical order SHPLONK sorts its rotation points by."""
    import ctypes as C
    import random

    from oracle import plonk as P

    src = os.path.join(ROOT, "tests", "host", "prover_host.cpp")
    so = os.path.join(ROOT, "tests", "host", "libproverhost.so")
    deps = [src, os.path.join(ROOT, "include", "h2mi.hpp"), os.path.join(ROOT, "halo2-scaffold_amd", "csrc", "h2mi_hostmath.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    L = C.CDLL(so)
    L.h2t_assembly.argtypes, L.h2t_assembly.restype = [C.c_void_p, C.c_size_t, C.c_void_p], C.c_size_t
    L.h2t_uniform_fr.argtypes = [C.c_uint64, C.c_size_t, C.c_uint64, C.c_void_p]
    L.h2t_normalize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.h2t_canonical_less.argtypes, L.h2t_canonical_less.restype = [C.c_void_p, C.c_void_p], C.c_int
    rng = random.Random(0xA55E)
    for trial in range(60):
        cols, n = rng.randint(1, 8), rng.choice([4, 16, 64])
        m = rng.randint(0, 3 * n)
        copies = []
        for _ in range(m):
            a, b = (rng.randrange(cols), rng.randrange(n)), (rng.randrange(cols), rng.randrange(n))
            if rng.random() < 0.05:
                b = a  # constrain_equal of a cell with itself
            if copies and rng.random() < 0.1:
                a, b = copies[rng.randrange(len(copies))]  # a repeated constraint
            copies.append((a, b))
        ref = P.Assembly(cols, n)
        for a, b in copies:
            ref.copy(a, b)
        want = sorted((c, r) + ref.mapping[c][r] for c in range(cols) for r in range(n) if ref.mapping[c][r] != (c, r))
        flat = np.array([[a[0], a[1], b[0], b[1]] for a, b in copies], dtype=np.uint32).reshape(-1, 4)
        out = np.zeros((2 * max(m, 1), 4), dtype=np.uint32)
        k = L.h2t_assembly(flat.ctypes.data, m, out.ctypes.data)
        assert [tuple(int(v) for v in row) for row in out[:k]] == want, trial
    # the blinding stream: seed, count, start
    for seed, count, start in ((1, 7, 0), (78, 18, 5), (0xFFFF_FFFF, 33, 1 << 20)):
        got = np.zeros((count, 4), dtype=np.uint64)
        L.h2t_uniform_fr(seed, count, start, got.ctypes.data)
        assert np.array_equal(got, o.random_field_limbs(count, seed, start=start))
    # the keyed stream: ChaCha20 (checked here against the RFC 7539 2.3.2 block vector) one block per scalar, Fr::from_u512
    ks = _chacha20_block(bytes(range(32)), 1, int.from_bytes(bytes([0, 0, 0, 9, 0, 0, 0, 0x4A, 0, 0, 0, 0])[4:], "little"), ctr32_nonce96=bytes([0, 0, 0, 9, 0, 0, 0, 0x4A, 0, 0, 0, 0]))
    assert ks.hex().startswith("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e")
    L.h2t_chacha_fr.argtypes = [C.c_char_p, C.c_uint64, C.c_size_t, C.c_uint64, C.c_void_p]
    key = bytes((7 * i + 3) & 0xFF for i in range(32))
    for stream, count, start in ((1, 9, 0), ((5 << 3) | 3, 40, 1 << 33), ((1 << 63) + 5, 3, 7)):
        got = np.zeros((count, 4), dtype=np.uint64)
        L.h2t_chacha_fr(key, stream, count, start, got.ctypes.data)
        assert o.unpack(got, o.R) == [int.from_bytes(_chacha20_block(key, start + i, stream), "little") % o.R for i in range(count)]
    # batch_normalize: random Jacobian representatives, the identity in the middle and at the ends
    pts = [None, o.g1_mul(5, o.G1_GEN), None, o.g1_mul(o.R - 1, o.G1_GEN), o.g1_mul(123456789, o.G1_GEN), None]
    jac = np.stack([o.pack_jacobian(p, z=rng.randrange(2, o.Q)) if p is not None else o.pack_jacobian(None) for p in pts])
    aff = np.zeros((len(pts), 8), dtype=np.uint64)
    L.h2t_normalize(np.ascontiguousarray(jac).ctypes.data, len(pts), aff.ctypes.data)
    assert o.unpack_points(aff) == pts
    # Fr's Ord is by canonical value, not by the Montgomery limbs
    vals = [0, 1, 2, o.R - 1, o.R // 2, 1 << 200, rng.randrange(o.R), rng.randrange(o.R)]
    lim = o.pack(vals, o.R)
    for i, a in enumerate(vals):
        for j, b in enumerate(vals):
            assert L.h2t_canonical_less(lim[i].ctypes.data, lim[j].ctypes.data) == (1 if a < b else 0)


def test_abi_headers_are_strict_c(tmp_path):
    """the drop-in boundary is a C ABI: both headers compile as ISO C99 and C11 with -pedantic -Werror (no C++-isms, no torch types),
    and the plain-C caller of the prover ABI builds against the library (it fails loudly without a GPU)."""
    src = tmp_path / "abi.c"
    src.write_text('#include "h2mi.h"\n#include "h2mi_prover.h"\nint main(void) { h2mi_constraint_system cs; h2mi_column_cells c; (void)cs; (void)c; '
                   "return (int)sizeof(h2mi_prover_counts) - 20; }\n")
    for std in ("c99", "c11"):
        subprocess.check_call(["gcc", f"-std={std}", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                               "-o", str(tmp_path / "abi.o")])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s", "prover_abi"])
    import torch

    if not torch.cuda.is_available():
        r = subprocess.run([os.path.join(ROOT, "examples", "prover_abi"), "5"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "no CPU fallback" in r.stderr


def test_host_layers_clean_under_sanitizers(tmp_path):
    """AddressSanitizer + UndefinedBehaviorSanitizer over everything that compiles for the host — the 29-bit-limb field / curve headers
    the kernels are built from, the division-step inversion, the prover's host helpers (permutation assembly, batch normalisation,
    blinding streams) — on seeded pseudo-random operands (tests/host/sanitize_main.cpp).  GPU sanitizers are not available on the
    pool; this is where the shared code gets its sanitizer run.  Any report aborts the program (-fno-sanitize-recover)."""
    exe = tmp_path / "sanitize_main"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-o", str(exe), os.path.join(ROOT, "tests", "host", "sanitize_main.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sanitize_main: done" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_cpp_host_layout_clean_under_sanitizers(tmp_path, h2):
    """the C++ host's CPU side — Context, the break-point layout over gate / lookup-advice / constants columns, configure,
    StandardPlonk's synthesize — under AddressSanitizer + UndefinedBehaviorSanitizer over ~150 configurations (DEGREE 4 .. 12,
    LOOKUP_BITS 1 .. 8, 1 .. 12 range checks, poseidon, halo2_lib; refusals included): tests/host/sanitize_flex.cpp.  No GPU call."""
    exe = tmp_path / "sanitize_flex"
    libdir = os.path.join(ROOT, "halo2-scaffold_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe),
                           os.path.join(ROOT, "tests", "host", "sanitize_flex.cpp"), "-L", libdir, "-lh2mi", f"-Wl,-rpath,{libdir}"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")  # the HIP runtime the library pulls in keeps its own allocations until exit
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "sanitize_flex: done" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_rust_mirror_of_the_prover_abi_matches_the_header():
    """interop/prove (Rust, compiled by nobody here) declares h2mi_constraint_system by hand: its array lengths must be the header's
    H2MI_MAX_* (a stale mirror would hand keygen a struct of the wrong size), its field order the header's, and the Python mirror
    (engine.py) the same numbers."""
    import re

    hdr = open(os.path.join(ROOT, "include", "h2mi_prover.h")).read() + open(os.path.join(ROOT, "include", "h2mi.h")).read()
    flex = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define H2MI_FLEX_MAX_(\w+) (\d+)", hdr)}
    want = {"GATES": flex["GATES"], "PERM": flex["PERM"], "LOOKUPS": flex["LOOKUPS"],
            "QUERIES": int(re.search(r"#define H2MI_MAX_QUERIES (\d+)", hdr).group(1))}
    rs = open(os.path.join(ROOT, "interop", "prove", "src", "main.rs")).read()
    got = {m.group(1): int(m.group(2)) for m in re.finditer(r"const MAX_(\w+): usize = (\d+);", rs)}
    assert got == want
    struct_c = re.search(r"typedef struct \{\s*uint32_t k;.*?\} h2mi_constraint_system;", hdr, re.S).group(0)
    fields_c = re.findall(r"\b(k|n_advice|n_fixed|n_instance|degree|blinding_factors|gates|n_gates|gate_advice|gate_selector|n_perm|perm_columns|n_lookups|"
                          r"lookups|n_advice_queries|n_fixed_queries|advice_queries|fixed_queries)\b(?=[\[;,])", struct_c)
    struct_rs = re.search(r"struct ConstraintSystem \{(.*?)\n\}", rs, re.S).group(1)
    fields_rs = re.findall(r"^\s*(\w+):", struct_rs, re.M)
    assert fields_rs == fields_c, (fields_rs, fields_c)
    eng = open(os.path.join(ROOT, "halo2-scaffold_amd", "engine.py")).read()
    m = re.search(r"MAX_GATES, MAX_PERM, MAX_LOOKUPS, MAX_QUERIES = (\d+), (\d+), (\d+), (\d+)", eng)
    assert [int(v) for v in m.groups()] == [want["GATES"], want["PERM"], want["LOOKUPS"], want["QUERIES"]]
