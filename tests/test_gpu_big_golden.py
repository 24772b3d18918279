"""Proof BYTES at the sizes BASELINE.json names: the device provers against tests/golden/big_proofs.json — golden proofs
made in the build container by the large-size oracle prover (oracle/fastflex.py; byte-identical to the Python-integer
oracle wherever that one finishes: tests/test_oracle_fast.py), each accepted by the oracle verifier against the closed-form
verifying key before it was committed.

  standard_plonk  DEGREE 16 (BASELINE configs[1]: "proof bytes == CPU") and DEGREE 20 (the north-star size)
                  reference: examples/standard_plonk.rs:41-50
  halo2_lib       DEGREE 20 (configs[2]); poseidon DEGREE 20 (configs[4]); range LOOKUP_BITS 12 at DEGREE 16 and
                  LOOKUP_BITS 16 at DEGREE 22 (configs[3]) — reference: src/scaffold.rs:322-331
Three hosts of the same C ABI must reproduce every byte: the Python host, the C++ host (examples/*.cpp) and the
single-process mode with four (virtual) devices.  These sizes are the only ones that run the multi-pass transforms, the
16 / 17-bit windows, the dominant-value shift against full window tables and the sparse grand products over 3 * 2^20 rows;
a verifier's acceptance (tests/test_gpu_prover.py, tests/test_gpu_flex.py) cannot see a wrong blinding row or a mis-ordered
but consistent transcript write — byte equality can."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "big_proofs.json")))
SECRET = int(GOLD["srs_secret"], 16)
CASES = {c["name"]: c for c in GOLD["cases"]}


def _closure(flex, poseidon, cs, case):
    shape, x = case["shape"], int(case["x"], 16)
    if shape == "range":
        return flex.range_closure(cs, x, case["lookup_bits"])
    if shape == "poseidon":
        return poseidon.hash_two_closure(cs, x, x + 1)
    return flex.halo2_lib_closure(cs, x)


@pytest.mark.parametrize("name", ["standard_plonk_k16", "standard_plonk_k20"])
def test_python_host_standard_plonk_bytes(gpu, name):
    from halo2_scaffold_amd import circuits, keygen, prover

    case = CASES[name]
    params = gpu.ParamsKZG.setup(case["k"], SECRET)
    circuit = circuits.StandardPlonk(None)
    vk = keygen.keygen_vk(params, circuit)
    pk = keygen.keygen_pk(params, vk, circuit)
    assert vk.to_bytes().hex() == case["vk_bytes"]
    trace = {}
    ws = prover.ProverWorkspace(params, pk)
    proof = prover.create_proof(params, pk, circuits.StandardPlonk(int(case["x"], 16)), case["seed"], ws=ws, trace=trace)
    for ch in ("theta", "beta", "gamma", "y", "x"):  # the first divergent challenge says which phase went wrong
        assert trace[ch] == int(case["challenges"][ch], 16), ch
    assert hashlib.sha256(proof).hexdigest() == case["proof_sha256"]
    assert proof.hex() == case["proof"]
    # a second proof through the reused workspace, then the golden one again: nothing stale survives in the buffers
    other = prover.create_proof(params, pk, circuits.StandardPlonk(7), 1, ws=ws)
    assert other != proof
    assert prover.create_proof(params, pk, circuits.StandardPlonk(int(case["x"], 16)), case["seed"], ws=ws) == proof
    ws.release()
    pk.release()
    params.release()


@pytest.mark.parametrize("name", ["halo2_lib_k20", "poseidon_k20", "range_k16_bits12", "range_k22_bits16"])
def test_python_host_halo2_lib_builders_bytes(gpu, name):
    from halo2_scaffold_amd import flex, poseidon

    case = CASES[name]
    params = gpu.ParamsKZG.setup(case["k"], SECRET)
    cs = flex.FlexGateCS(lookup=case["shape"] == "range")
    asg = _closure(flex, poseidon, cs, case)
    assert asg.instance == [int(v, 16) for v in case["instance"]]
    keys = flex.FlexKeys(params, cs, asg)
    assert keys.vk_bytes().hex() == case["vk_bytes"]
    trace = {}
    proof = flex.create_proof(params, keys, asg, case["seed"], trace=trace)
    for ch in ("theta", "beta", "gamma", "y", "x"):
        assert trace[ch] == int(case["challenges"][ch], 16), ch
    assert hashlib.sha256(proof).hexdigest() == case["proof_sha256"]
    assert proof.hex() == case["proof"]
    keys.release()
    params.release()


def test_cpp_host_bytes(gpu):
    """examples/standard_plonk.cpp and examples/halo2_lib.cpp (include/h2mi_plonk.hpp, h2mi_flex.hpp) at the same sizes"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    for name in ("standard_plonk_k16", "standard_plonk_k20"):
        case = CASES[name]
        r = subprocess.run([os.path.join(ROOT, "examples", "standard_plonk"), str(case["k"]), hex(SECRET), case["x"], str(case["seed"])],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
        out = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.startswith(("vk ", "proof ")))
        assert out["vk"] == case["vk_bytes"], name
        assert out["proof"] == case["proof"], name
    for name in ("halo2_lib_k20", "poseidon_k20", "range_k16_bits12", "range_k22_bits16"):
        case = CASES[name]
        r = subprocess.run([os.path.join(ROOT, "examples", "halo2_lib"), case["shape"], str(case["k"]), str(case["lookup_bits"]), str(int(case["x"], 16)),
                            hex(SECRET), str(case["seed"])], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
        out = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.startswith(("vk ", "proof ")))
        assert out["vk"] == case["vk_bytes"], name
        assert out["proof"] == case["proof"], name


_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import torch  # first: one HIP runtime
import _load_pkg
h2 = _load_pkg.load()
lib = h2.lib
assert lib.h2mi_init_devices(4) == 0 and lib.h2mi_device_count() == 4
gold = json.load(open(os.path.join({root!r}, "tests", "golden", "big_proofs.json")))
secret = int(gold["srs_secret"], 16)
cases = {{c["name"]: c for c in gold["cases"]}}
from halo2_scaffold_amd import circuits, keygen, prover, flex, poseidon
case = cases["standard_plonk_k20"]
params = h2.ParamsKZG.setup(case["k"], secret)          # generated on the primary device, registered sharded
circuit = circuits.StandardPlonk(None)
vk = keygen.keygen_vk(params, circuit)
pk = keygen.keygen_pk(params, vk, circuit)
assert vk.to_bytes().hex() == case["vk_bytes"]
proof = prover.create_proof(params, pk, circuits.StandardPlonk(int(case["x"], 16)), case["seed"])
assert proof.hex() == case["proof"], "standard_plonk_k20"
pk.release()
for name in ("poseidon_k20", "halo2_lib_k20"):
    case = cases[name]
    cs = flex.FlexGateCS(lookup=False)
    x = int(case["x"], 16)
    asg = poseidon.hash_two_closure(cs, x, x + 1) if case["shape"] == "poseidon" else flex.halo2_lib_closure(cs, x)
    keys = flex.FlexKeys(params, cs, asg)
    assert keys.vk_bytes().hex() == case["vk_bytes"]
    assert flex.create_proof(params, keys, asg, case["seed"]).hex() == case["proof"], name
    keys.release()
params.release()
case = cases["range_k16_bits12"]
params = h2.ParamsKZG.setup(case["k"], secret)
cs = flex.FlexGateCS(lookup=True)
asg = flex.range_closure(cs, int(case["x"], 16), case["lookup_bits"])
keys = flex.FlexKeys(params, cs, asg)
assert flex.create_proof(params, keys, asg, case["seed"]).hex() == case["proof"], "range_k16_bits12"
print("BIG_MULTIDEV_OK")
"""


def test_four_virtual_devices_bytes(gpu, tmp_path):
    """h2mi_init_devices(4) (H2MI_VIRTUAL_DEVICES=1 on a one-GPU box): every commitment a sharded MSM — slice pipelines per
    device, gather + fold at the joins — and still the same bytes at 2^20 rows"""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, H2MI_VIRTUAL_DEVICES="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "BIG_MULTIDEV_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


_WORKER8 = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import torch  # first: one HIP runtime
import _load_pkg
h2 = _load_pkg.load()
lib = h2.lib
assert lib.h2mi_init_devices(8) == 0 and lib.h2mi_device_count() == 8
gold = json.load(open(os.path.join({root!r}, "tests", "golden", "big_proofs.json")))
secret = int(gold["srs_secret"], 16)
case = {{c["name"]: c for c in gold["cases"]}}["range_k22_bits16"]
from halo2_scaffold_amd import flex
import ctypes as C
params = h2.ParamsKZG.setup(case["k"], secret)          # 2^22 bases: eight slices of 2^19, as on the real node
c, W, nb, nn = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
assert lib.h2mi_bases_info(params.g_handle, C.byref(c), C.byref(W), C.byref(nb), C.byref(nn)) == 0
assert nn.value == 1 << 22 and c.value == 16, (nn.value, c.value)   # the slices' own window width (2^19 points), not the 20 bits of 2^22
cs = flex.FlexGateCS(lookup=True)
asg = flex.range_closure(cs, int(case["x"], 16), case["lookup_bits"])
assert asg.instance == [int(v, 16) for v in case["instance"]]
keys = flex.FlexKeys(params, cs, asg)
assert keys.vk_bytes().hex() == case["vk_bytes"]
trace = {{}}
proof = flex.create_proof(params, keys, asg, case["seed"], trace=trace)
for ch in ("theta", "beta", "gamma", "y", "x"):
    assert trace[ch] == int(case["challenges"][ch], 16), ch
assert proof.hex() == case["proof"], "range_k22_bits16"
print("BIG_MULTIDEV8_OK")
"""


def test_eight_virtual_devices_range_k22_bytes(gpu, tmp_path):
    """BASELINE configs[3] at its own size and its own partition: range_check with LOOKUP_BITS = 16 at DEGREE 22, every commitment a
    sharded MSM over EIGHT slices of 2^19 bases (h2mi_init_devices(8), virtual on a one-GPU box: 8 x 2 window tables of 0.5 GB) —
    the slices take 16-bit windows where the single-GPU run takes 20 — against the committed golden proof (src/scaffold.rs:434-485)."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8.format(root=ROOT))
    env = dict(os.environ, H2MI_VIRTUAL_DEVICES="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0 and "BIG_MULTIDEV8_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
