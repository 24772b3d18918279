"""GPU test of the single-process multi-device mode (h2mi_init_devices): one host thread drives n
devices through the unchanged C ABI.  On a one-GPU box the n entries are virtual (H2MI_VIRTUAL_DEVICES=1: all on GPU 0),
which exercises everything but the physical peer copies: sharded registration, scalar distribution, per-device MSM
pipelines and reductions, gather + fold at the join.  Runs in a child process (the library is initialised once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import ctypes as C, json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch  # first: one HIP runtime
import _load_pkg
h2 = _load_pkg.load()
from oracle import bn254 as o, cref
lib = h2.lib
NDEV = {ndev}
assert lib.h2mi_device_count() == 0
assert lib.h2mi_init_devices(NDEV) == 0 and lib.h2mi_device_count() == NDEV
assert lib.h2mi_init_devices(NDEV) == 0 and lib.h2mi_init(0) == -1      # idempotent; exclusive with h2mi_init
k = 13
n = 1 << k
bases = cref.g1_mul_gen(o.random_field_limbs(n, 11), 4)
hreg = C.c_uint64()
assert lib.h2mi_bases_register(bases.ctypes.data, n, C.byref(hreg)) == 0  # host bases -> one slice per device
nn = C.c_uint64(); cc = C.c_uint32()
assert lib.h2mi_bases_info(hreg.value, C.byref(cc), None, None, C.byref(nn)) == 0 and nn.value == n
out = np.zeros(12, dtype=np.uint64)
for seed, m in [(1, n), (2, n), (3, n - 1), (4, n // NDEV + 5), (5, 3), (6, n)]:
    sc = o.random_field_limbs(n, 40 + seed)[:m].copy()
    if seed == 6:
        sc[:] = sc[0]   # a constant column: every slice takes the dominant-value path
    assert lib.h2mi_msm_bn254_g1(hreg.value, None, sc.ctypes.data, m, out.ctypes.data) == 0
    assert o.unpack_jacobian(out) == o.unpack_jacobian(cref.msm(sc, bases[:m], 4)), (seed, m)
# device-resident scalars on the primary device, several MSMs queued before one join (more than the ring holds)
scal = [o.random_field_limbs(n, 70 + i) for i in range(20)]
d_sc = [h2.DevBuf.from_numpy(s) for s in scal]
d_out = h2.DevBuf(96 * len(scal))
for i, d in enumerate(d_sc):
    assert lib.h2mi_msm_bn254_g1_dev(hreg.value, d.ptr, n, d_out.ptr + 96 * i, None) == 0
got = d_out.to_numpy(shape=(len(scal), 12))   # joins, gathers, folds
for i, s in enumerate(scal):
    assert o.unpack_jacobian(got[i]) == o.unpack_jacobian(cref.msm(s, bases, 4)), i
ba, ra = C.c_uint64(), C.c_uint64()
assert lib.h2mi_msm_last_stats(hreg.value, C.byref(ba), C.byref(ra)) == 0 and ba.value > n * 10
assert lib.h2mi_bases_release(hreg.value) == 0
# the whole prover on top of it, unchanged: SRS generated on the primary device, registered (sharded) from device
# memory, every commitment a sharded MSM: the proof bytes must equal the committed golden proof
from halo2_scaffold_amd import circuits, keygen, prover
gold = json.load(open(os.path.join({root!r}, "tests", "golden", "standard_plonk_proofs.json")))
case = gold["cases"][1]
params = h2.ParamsKZG.setup(case["k"], int(gold["srs_secret"], 16))
circuit = circuits.StandardPlonk(None)
vk = keygen.keygen_vk(params, circuit)
pk = keygen.keygen_pk(params, vk, circuit)
assert vk.to_bytes().hex() == case["vk_bytes"]
proof = prover.create_proof(params, pk, circuits.StandardPlonk(int(case["witness_x"], 16)), case["seed"])
assert proof.hex() == case["proof"]
# and the Range builder's prover (lookup argument, side streams, sparse grand products) over sharded commitments: same
# bytes as the oracle engine's proof
from halo2_scaffold_amd import flex
from oracle import flex as FX
S2, k2, bits, x2 = 0x5EC2E7 + 0x48324D49, 8, 6, 0xFEEDC0DE77
p2 = h2.ParamsKZG.setup(k2, S2)
cs = flex.FlexGateCS(lookup=True)
asg = flex.range_closure(cs, x2, bits)
keys = flex.FlexKeys(p2, cs, asg)
proof2 = flex.create_proof(p2, keys, asg, 21)
ocs = FX.flex_gate_cs(True)
oasg = FX.range_assignment(ocs, x2, bits, 1 << k2)
okeys = FX.Keys(ocs, k2, S2, oasg.fixed, oasg.copies)
assert keys.vk_bytes() == okeys.vk_bytes()
assert proof2 == FX.prove(okeys, oasg, 21)["proof"]
print("MULTIDEV_OK", NDEV)
"""


@pytest.mark.parametrize("ndev", [4, 3])
def test_single_process_multi_device_virtual(gpu, tmp_path, ndev):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, ndev=ndev))
    env = dict(os.environ, H2MI_VIRTUAL_DEVICES="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and f"MULTIDEV_OK {ndev}" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_single_process_two_physical_devices(gpu, tmp_path):
    """the physical peer-copy path (hipMemcpyPeerAsync between two real devices, cross-device hipStreamWaitEvent): the same
    worker WITHOUT H2MI_VIRTUAL_DEVICES — MSMs against the C oracle, the golden StandardPlonk proof and the Range builder's
    proof over sharded commitments.  Skipped on a one-GPU box (the builder's boxes); the first node with >= 2 GPUs runs it.
    N > 1 on hardware is otherwise unmeasured."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two physical GPUs (hipGetDeviceCount() >= 2)")
    script = tmp_path / "worker2.py"
    script.write_text(_WORKER.format(root=ROOT, ndev=2))
    env = {k: v for k, v in os.environ.items() if k != "H2MI_VIRTUAL_DEVICES"}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "MULTIDEV_OK 2" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


# ---- the ABI from two host threads ("calls may arrive concurrently from rayon worker threads") ---------------------
# HIP's current device belongs to the HOST THREAD.  Round 3 cached "the device hipSetDevice last selected" process-wide, so a second
# thread calling after h2mi_init(d != 0) allocated and launched on device 0.  The worker drives the library from two threads at
# once (ctypes releases the GIL inside every call, so the library's mutex and the per-thread device binding are really exercised):
# thread A: host-pointer MSMs against a registered handle + device-pointer MSMs; thread B: host-pointer NTTs, device allocations,
# uploads, evaluations — every result against the oracle.  DEVICE != 0 additionally initialises from a NON-main thread.
_THREAD_WORKER = r"""
import ctypes as C, os, sys, threading
sys.path.insert(0, {root!r})
import numpy as np
import torch  # first: one HIP runtime
import _load_pkg
h2 = _load_pkg.load()
from oracle import bn254 as o, cref
from halo2_scaffold_amd import arithmetic as A, field as F
lib = h2.lib
DEVICE = {device}
errors = []

def guarded(fn):
    def run():
        try:
            fn()
        except BaseException as e:  # surfaces in the main thread below
            import traceback
            errors.append(traceback.format_exc())
    return run

init_thread = threading.Thread(target=guarded(lambda: h2.init(DEVICE)))   # h2mi_init from a thread that is NOT the caller of the rest
init_thread.start(); init_thread.join()
assert not errors, errors
k = 12
n = 1 << k
bases = cref.g1_mul_gen(o.random_field_limbs(n, 21), 4)
hreg = C.c_uint64()
assert lib.h2mi_bases_register(bases.ctypes.data, n, C.byref(hreg)) == 0
ITER = 12

def thread_a():
    out = np.zeros(12, dtype=np.uint64)
    for it in range(ITER):
        m = [n, n - 3, 257, 31, 1, n][it % 6]
        sc = o.random_field_limbs(n, 300 + it)[:m].copy()
        assert lib.h2mi_msm_bn254_g1(hreg.value, None, sc.ctypes.data, m, out.ctypes.data) == 0
        assert o.unpack_jacobian(out) == o.unpack_jacobian(cref.msm(sc, bases[:m], 2)), ("host msm", it)
        d = h2.DevBuf.from_numpy(sc)                       # h2mi_malloc + h2mi_memcpy_h2d on this thread
        r = h2.DevBuf(96)
        assert lib.h2mi_msm_bn254_g1_dev(hreg.value, d.ptr, m, r.ptr, None) == 0
        got = r.to_numpy(shape=(12,))                      # joins + D2H
        assert o.unpack_jacobian(got) == o.unpack_jacobian(cref.msm(sc, bases[:m], 2)), ("dev msm", it)
        d.free(); r.free()

def thread_b():
    for it in range(ITER):
        lg = [10, 12, 7, 13][it % 4]
        a = o.random_field_limbs(1 << lg, 500 + it)
        want = o.ntt(o.unpack(a, o.R), o.omega_for(lg))
        w = F.fr_to_mont_limbs(F.omega_for(lg))
        A.best_fft(a, w, lg)                               # host-pointer NTT
        assert o.unpack(a, o.R) == want, ("ntt", it)
        coeffs = o.random_field_limbs(777, 900 + it)
        pt = (0x1234567 * (it + 1)) % o.R
        got = A.eval_polynomial(coeffs, F.fr_to_mont_limbs(pt))   # malloc, upload, device evaluation, download
        assert o.unpack(got.reshape(1, 4), o.R)[0] == o.eval_polynomial(o.unpack(coeffs, o.R), pt), ("eval", it)

ts = [threading.Thread(target=guarded(thread_a)), threading.Thread(target=guarded(thread_b))]
for t in ts: t.start()
for t in ts: t.join()
assert not errors, "\n".join(errors)
# everything the threads allocated lives on DEVICE: a device-0 allocation on a box where DEVICE != 0 is the round-3 bug
if torch.cuda.device_count() > 1 and DEVICE != 0:
    probe = h2.DevBuf(1 << 20)
    attr_dev = C.c_int(-1)
    hip = C.CDLL("libamdhip64.so")
    class Attr(C.Structure):
        _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p),
                    ("isManaged", C.c_int), ("allocationFlags", C.c_uint)]
    at = Attr()
    assert hip.hipPointerGetAttributes(C.byref(at), C.c_void_p(probe.ptr)) == 0
    assert at.device == DEVICE, (at.device, DEVICE)
    probe.free()
assert lib.h2mi_bases_release(hreg.value) == 0
print("THREADS_OK", DEVICE)
"""


def test_abi_from_two_host_threads(gpu, tmp_path):
    """two Python threads interleave MSMs and NTTs through the C ABI on device 0 (initialised from a third thread): oracle parity
    on every call."""
    script = tmp_path / "threads0.py"
    script.write_text(_THREAD_WORKER.format(root=ROOT, device=0))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0 and "THREADS_OK 0" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_abi_from_two_host_threads_on_device_one(gpu, tmp_path):
    """the same with h2mi_init(1) from a non-main thread: every worker thread must end up on device 1 (allocation owner checked
    with hipPointerGetAttributes).  Needs two physical GPUs."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two physical GPUs (hipGetDeviceCount() >= 2)")
    script = tmp_path / "threads1.py"
    script.write_text(_THREAD_WORKER.format(root=ROOT, device=1))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0 and "THREADS_OK 1" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
