"""GPU tests of the halo2-lib-shaped provers (BASELINE configs[2] halo2_lib and configs[3] range):
`flex.FlexKeys` / `flex.create_proof` on the device against the oracle's data-driven prover and verifier
(oracle/flex.py) — key material and proof bytes equal at small k, and the oracle's verifier (transcript replay, the
constraint identity at x, SHPLONK, the pairing when asked) accepts device proofs at larger k."""
import numpy as np
import pytest

from oracle import bn254 as o
from oracle import flex as FX
from oracle import formats as fm

pytestmark = pytest.mark.gpu

SRS_SECRET = 0x5EC2E7 + 0x48324D49


def _vals(buf, count):
    return o.unpack(buf.to_numpy(shape=(count, 4), nbytes=count * 32), o.R)


def _oracle_keys(cs, k, oasg):
    return FX.Keys(cs, k, SRS_SECRET, oasg.fixed, oasg.copies)


def _check_keys(keys, okeys, n):
    assert o.unpack_points(keys.fixed_commitments) == okeys.fixed_commitments
    assert o.unpack_points(keys.permutation_commitments) == okeys.permutation_commitments
    assert keys.vk_bytes() == okeys.vk_bytes() and keys.transcript_repr == okeys.transcript_repr
    for dev, want in zip(keys.sigma_values, okeys.sigma):
        assert _vals(dev, n) == want
    for dev, want in zip(keys.fixed_polys, okeys.fixed_polys):
        assert _vals(dev, n) == want


@pytest.mark.parametrize("k", [6, 9])
def test_halo2_lib_proof_bytes_match_oracle(gpu, k):
    """reference examples/halo2_lib.rs through scaffold::prove (src/scaffold.rs:246-366) with DEGREE = k: the circuit's
    cells, the keys and the proof, byte for byte; the oracle's verifier accepts it with the public inputs [x, x^2 + 72]
    and rejects it with others."""
    from halo2_scaffold_amd import flex

    x, seed = 12, 2024
    params = gpu.ParamsKZG.setup(k, SRS_SECRET)
    cs = flex.FlexGateCS(lookup=False)
    asg = flex.halo2_lib_closure(cs, x)
    ocs = FX.flex_gate_cs(False)
    oasg = FX.halo2_lib_assignment(ocs, x)
    assert asg.advice == oasg.advice and asg.fixed == oasg.fixed and asg.copies == oasg.copies and [asg.instance] == oasg.instance
    assert asg.instance == [x, x * x + 72]
    keys = flex.FlexKeys(params, cs, asg)
    okeys = _oracle_keys(ocs, k, oasg)
    _check_keys(keys, okeys, 1 << k)
    trace = {}
    proof = flex.create_proof(params, keys, asg, seed, trace=trace)
    want = FX.prove(okeys, oasg, seed)
    for name in ("theta", "beta", "gamma", "y", "x"):
        assert trace[name] == want[name], name
    assert proof == want["proof"]
    assert len(proof) == 864
    assert FX.verify(okeys, proof, [asg.instance])
    assert not FX.verify(okeys, proof, [[x, x * x + 73]])
    keys.release()
    params.release()


@pytest.mark.parametrize("k,lookup_bits,x", [(7, 4, 0xDEADBEEFCAFE1234), (8, 7, (1 << 64) - 1), (8, 6, 0)])
def test_range_proof_bytes_match_oracle(gpu, k, lookup_bits, x):
    """reference examples/range.rs: range_check(x, 64) with LOOKUP_BITS limbs (64 is a multiple of 4; with 7 the top limb
    is one bit: assert_bit; with 6 it is four bits: the shifted cell is looked up too), lookup argument — input expression
    q_lookup * a — and all."""
    from halo2_scaffold_amd import flex

    seed = 99
    params = gpu.ParamsKZG.setup(k, SRS_SECRET)
    cs = flex.FlexGateCS(lookup=True)
    asg = flex.range_closure(cs, x, lookup_bits)
    ocs = FX.flex_gate_cs(True)
    oasg = FX.range_assignment(ocs, x, lookup_bits, 1 << k)
    assert asg.advice == oasg.advice and asg.copies == oasg.copies and [asg.instance] == oasg.instance
    assert all(asg.fixed[c] == oasg.fixed[c] for c in (cs.col_const, cs.col_qlookup, cs.col_q)) and asg.fixed[cs.col_table] is None
    assert (cs.col_table, cs.col_const, cs.col_qlookup, cs.col_q) == (ocs.col_table, ocs.col_const, ocs.col_qlookup, ocs.col_q)
    keys = flex.FlexKeys(params, cs, asg)
    okeys = _oracle_keys(ocs, k, oasg)
    _check_keys(keys, okeys, 1 << k)
    trace = {}
    proof = flex.create_proof(params, keys, asg, seed, trace=trace)
    want = FX.prove(okeys, oasg, seed)
    for name in ("theta", "beta", "gamma", "y", "x"):
        assert trace[name] == want[name], name
    assert proof == want["proof"]
    assert len(proof) == 992  # 12 commitments + 19 evaluations (degree 5: one permutation set, four h pieces)
    assert FX.verify(okeys, proof, [asg.instance])
    bad = bytearray(proof)
    bad[700] ^= 1
    assert not FX.verify(okeys, bytes(bad), [asg.instance])
    keys.release()
    params.release()


def test_range_rejects_out_of_range_witness(gpu):
    """a looked-up cell that is not a table value: the crate's permute_expression_pair fails the proof; so does
    the device's counting sort (no proof is produced)."""
    from halo2_scaffold_amd import flex

    params = gpu.ParamsKZG.setup(7, SRS_SECRET)
    cs = flex.FlexGateCS(lookup=True)
    asg = flex.range_closure(cs, 1234567, 4)
    keys = flex.FlexKeys(params, cs, asg)
    asg.advice[0][sorted(asg.fixed[cs.col_qlookup])[2]] = 16  # one past the 4-bit table, on a row with q_lookup enabled
    with pytest.raises(ValueError, match="not in the table"):
        flex.create_proof(params, keys, asg, 5)
    keys.release()
    params.release()


@pytest.mark.parametrize("k,lookup_bits", [(12, 11), (13, 8), (22, 16)])
def test_range_proof_verifies_at_larger_k(gpu, k, lookup_bits):
    """sizes the pure-Python prover does not reach: the oracle's verifier needs only the verifying key (closed form:
    FX.VerifierKeys, time proportional to the assigned cells) and the proof.  k = 22 with LOOKUP_BITS = 16 is BASELINE
    configs[3]'s circuit on one device; k = 13 finishes with the real pairing check against the SRS's G2 elements, as
    the reference's verify_proof does."""
    import time

    from halo2_scaffold_amd import flex

    x, seed = 0x0123456789ABCDEF, 31337
    params = gpu.ParamsKZG.setup(k, SRS_SECRET)
    cs = flex.FlexGateCS(lookup=True)
    asg = flex.range_closure(cs, x, lookup_bits)
    keys = flex.FlexKeys(params, cs, asg)
    t0 = time.perf_counter()
    proof = flex.create_proof(params, keys, asg, seed)
    print(f"range k={k} lookup_bits={lookup_bits}: create_proof {1e3 * (time.perf_counter() - t0):.1f} ms (first call, buffers allocated inside)")
    ocs = FX.flex_gate_cs(True)
    oasg = FX.range_assignment(ocs, x, lookup_bits, 1 << k)
    vk = FX.VerifierKeys(ocs, k, SRS_SECRET, oasg.fixed, oasg.copies)
    assert keys.vk_bytes()[8:] == b"".join(fm.g1_to_bytes(c) for c in vk.fixed_commitments + vk.permutation_commitments)
    assert keys.transcript_repr == vk.transcript_repr
    if k == 13:
        assert FX.verify(vk, proof, [asg.instance], g2=fm.G2_GEN, s_g2=fm.g2_mul(SRS_SECRET))
    else:
        assert FX.verify(vk, proof, [asg.instance])
    assert not FX.verify(vk, proof, [[x + 1]])
    keys.release()
    params.release()


def test_halo2_lib_proof_verifies_at_degree_20(gpu):
    """BASELINE configs[2]: examples/halo2_lib.rs at DEGREE = 20"""
    import time

    from halo2_scaffold_amd import flex

    k, x, seed = 20, 0xC0FFEE, 4242
    params = gpu.ParamsKZG.setup(k, SRS_SECRET)
    cs = flex.FlexGateCS(lookup=False)
    asg = flex.halo2_lib_closure(cs, x)
    keys = flex.FlexKeys(params, cs, asg)
    t0 = time.perf_counter()
    proof = flex.create_proof(params, keys, asg, seed)
    print(f"halo2_lib k={k}: create_proof {1e3 * (time.perf_counter() - t0):.1f} ms")
    ocs = FX.flex_gate_cs(False)
    oasg = FX.halo2_lib_assignment(ocs, x)
    vk = FX.VerifierKeys(ocs, k, SRS_SECRET, oasg.fixed, oasg.copies)
    assert keys.transcript_repr == vk.transcript_repr
    assert FX.verify(vk, proof, [asg.instance])
    assert not FX.verify(vk, proof, [[x, x * x + 71]])
    keys.release()
    params.release()


def test_workspace_reuse_and_sliced_srs_world2(gpu):
    """tools/flex_proof.py: repeated proofs against one pk reuse the workspace's buffers; with two ranks sharing this GPU
    over gloo each commitment is a slice MSM combined at every transcript write, and both ranks end with the single-
    process proof (BASELINE configs[3]'s partition, rehearsed at k = 13)."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["tools/flex_proof.py", "--shape", "range", "--k", "13", "--lookup-bits", "8", "--proofs", "2"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r1 = subprocess.run([sys.executable] + common, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert one["proof_bytes"] == 992 and one["combines_per_proof"] == 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + common + ["--gpus", "2"]
    r2 = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900, env=dict(env, H2MI_DIST_BACKEND="gloo", H2MI_DEVICE="0"))
    assert r2.returncode == 0, r2.stdout[-1000:] + r2.stderr[-2000:]
    two = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["proof_sha256"] == one["proof_sha256"]
    assert two["combines_per_proof"] == 6  # advice; permuted columns; z, lookup z, random; h pieces; the two SHPLONK commitments
    # many columns (24 range checks at DEGREE 7: 11 gate + 4 lookup-advice columns): a phase now returns up to 15 points, more than the
    # eight result slots the sliced path had until round 5 — every rank still ends with the single-process proof
    wide = ["tools/flex_proof.py", "--shape", "range", "--k", "7", "--lookup-bits", "4", "--count", "24", "--configure", "--proofs", "2"]
    r1 = subprocess.run([sys.executable] + wide, cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert one["columns"] == [11, 4, 1] and one["combines_per_proof"] == 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + wide + ["--gpus", "2"]
    r2 = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900, env=dict(env, H2MI_DIST_BACKEND="gloo", H2MI_DEVICE="0"))
    assert r2.returncode == 0, r2.stdout[-1000:] + r2.stderr[-2000:]
    two = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["proof_sha256"] == one["proof_sha256"] and two["combines_per_proof"] == 6


def _oracle_assignment(ocs, asg):
    oa = FX.Assignment(ocs)
    oa.advice = [dict(c) for c in asg.advice]
    oa.fixed = [dict(c) for c in asg.fixed]
    oa.instance = [list(asg.instance)]
    oa.copies = list(asg.copies)
    return oa


def test_poseidon_proof_bytes_match_oracle(gpu):
    """reference examples/poseidon.rs `hash_two` (T = 3, RATE = 2, R_F = 8, R_P = 57) as ~7.4 k FlexGate cells at k = 13:
    the public hash equals the oracle's Poseidon sponge (pinned by circomlib's published constants), and the device proof
    equals the oracle engine's proof of the same cells byte for byte."""
    from oracle import poseidon as OPS

    from halo2_scaffold_amd import flex, poseidon

    k, x, y, seed = 13, 0xFEEDFACE, 0xC0DE, 17
    params = gpu.ParamsKZG.setup(k, SRS_SECRET)
    cs = flex.FlexGateCS(lookup=False)
    asg = poseidon.hash_two_closure(cs, x, y)
    assert asg.instance == [x, y, OPS.sponge_hash([x, y])]
    keys = flex.FlexKeys(params, cs, asg)
    proof = flex.create_proof(params, keys, asg, seed)
    ocs = FX.flex_gate_cs(False)
    oasg = _oracle_assignment(ocs, asg)
    okeys = _oracle_keys(ocs, k, oasg)
    assert keys.vk_bytes() == okeys.vk_bytes()
    assert proof == FX.prove(okeys, oasg, seed)["proof"]
    assert FX.verify(okeys, proof, [asg.instance])
    assert not FX.verify(okeys, proof, [[x, y, asg.instance[2] ^ 1]])
    keys.release()
    params.release()


def test_poseidon_proof_verifies_at_degree_20(gpu):
    """BASELINE configs[4]: examples/poseidon.rs at DEGREE = 20 on one device"""
    from halo2_scaffold_amd import flex, poseidon

    k, x, y, seed = 20, 3, 4, 99
    params = gpu.ParamsKZG.setup(k, SRS_SECRET)
    cs = flex.FlexGateCS(lookup=False)
    asg = poseidon.hash_two_closure(cs, x, y)
    keys = flex.FlexKeys(params, cs, asg)
    proof = flex.create_proof(params, keys, asg, seed)
    ocs = FX.flex_gate_cs(False)
    oasg = _oracle_assignment(ocs, asg)
    vk = FX.VerifierKeys(ocs, k, SRS_SECRET, oasg.fixed, oasg.copies)
    assert keys.transcript_repr == vk.transcript_repr
    assert FX.verify(vk, proof, [asg.instance])
    keys.release()
    params.release()


def test_cpp_host_halo2_lib_examples(gpu):
    """the C++ host layer for the halo2-lib builders (include/h2mi_flex.hpp, examples/halo2_lib.cpp): Context, keygen
    and create_proof in C++ over the same C ABI.  Its verifying key and proof bytes equal the oracle engine's (and so
    the Python host's) for the halo2_lib closure and for the range closure (LOOKUP_BITS 4: no remainder, 7: one-bit top
    limb, 6: shifted top limb) and for poseidon's hash_two; at DEGREE 13 its range proof is accepted against the closed-form
    verifying key."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "-s"])
    exe = os.path.join(root, "examples", "halo2_lib")

    def run(shape, k, bits, x, seed):
        r = subprocess.run([exe, shape, str(k), str(bits), str(x), hex(SRS_SECRET), str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
        lines = [l.split(" ", 1) for l in r.stdout.splitlines() if l.startswith(("vk ", "proof ", "instance "))]
        return ([v for n, v in lines if n == "vk"][0], [v for n, v in lines if n == "proof"][0], [int(v, 16) for n, v in lines if n == "instance"])

    for shape, k, bits, x, seed in (("halo2_lib", 6, 0, 12, 2024), ("range", 7, 4, 0xDEADBEEFCAFE1234, 99), ("range", 8, 7, (1 << 64) - 1, 5),
                                    ("range", 8, 6, 77, 6)):
        vk, proof, instance = run(shape, k, bits, x, seed)
        ocs = FX.flex_gate_cs(shape == "range")
        oasg = FX.range_assignment(ocs, x, bits, 1 << k) if shape == "range" else FX.halo2_lib_assignment(ocs, x)
        okeys = _oracle_keys(ocs, k, oasg)
        assert vk == okeys.vk_bytes().hex(), shape
        assert [instance] == oasg.instance
        assert proof == FX.prove(okeys, oasg, seed)["proof"].hex(), (shape, bits)
    # examples/poseidon.rs hash_two(x, x + 1): the C++ Grain LFSR / chip lay out the same 7.4 k cells as the Python host,
    # the public hash is the oracle's sponge, the proof equals the oracle engine's
    from oracle import poseidon as OPS

    from halo2_scaffold_amd import flex, poseidon

    k, x, seed = 13, 0xFEEDFACE, 17
    vk, proof, instance = run("poseidon", k, 0, x, seed)
    assert instance == [x, x + 1, OPS.sponge_hash([x, x + 1])]
    ocs = FX.flex_gate_cs(False)
    oasg = _oracle_assignment(ocs, poseidon.hash_two_closure(flex.FlexGateCS(lookup=False), x, x + 1))
    okeys = _oracle_keys(ocs, k, oasg)
    assert vk == okeys.vk_bytes().hex()
    assert proof == FX.prove(okeys, oasg, seed)["proof"].hex()
    k, bits, x = 13, 9, 0x0123456789ABCDEF
    vk, proof, instance = run("range", k, bits, x, 3)
    ocs = FX.flex_gate_cs(True)
    oasg = FX.range_assignment(ocs, x, bits, 1 << k)
    ovk = FX.VerifierKeys(ocs, k, SRS_SECRET, oasg.fixed, oasg.copies)
    assert FX.verify(ovk, bytes.fromhex(proof), [instance])
    assert not FX.verify(ovk, bytes.fromhex(proof), [[x ^ 1]])


def test_both_hosts_reproduce_the_committed_golden_proofs(gpu):
    """tests/golden/flex_proofs.json — proofs of the halo2_lib, range (LOOKUP_BITS 4 / 7 / 6) and poseidon closures made by
    the oracle engine (the poseidon cells laid out there from the oracle's own permutation) — against the Python host and
    the C++ host (examples/halo2_lib): verifying key and proof bytes equal, public inputs equal."""
    import json
    import os
    import subprocess

    from halo2_scaffold_amd import flex, poseidon

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", "flex_proofs.json")))
    secret = int(g["srs_secret"], 16)
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "-s"])
    exe = os.path.join(root, "examples", "halo2_lib")
    for case in g["cases"]:
        shape, k, bits, x, seed = case["shape"], case["k"], case["lookup_bits"], int(case["x"], 16), case["seed"]
        params = gpu.ParamsKZG.setup(k, secret)
        cs = flex.FlexGateCS(lookup=shape == "range")
        asg = (flex.range_closure(cs, x, bits) if shape == "range" else poseidon.hash_two_closure(cs, x, x + 1) if shape == "poseidon"
               else flex.halo2_lib_closure(cs, x))
        keys = flex.FlexKeys(params, cs, asg)
        assert keys.vk_bytes().hex() == case["vk_bytes"], shape
        assert flex.create_proof(params, keys, asg, seed).hex() == case["proof"], (shape, bits)
        assert asg.instance == [int(v, 16) for v in case["instance"]]
        keys.release()
        params.release()
        r = subprocess.run([exe, shape, str(k), str(bits), str(x), hex(secret), str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1000:]
        out = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.startswith(("vk ", "proof ")))
        assert out["vk"] == case["vk_bytes"] and out["proof"] == case["proof"], ("C++", shape, bits)


# ---- round 4: more than one gate column (builder.config(k, Some(minimum_rows)) on overflow: src/scaffold.rs:268) --------------------
def test_multi_column_proofs_match_oracle_and_golden(gpu):
    """the closures at a DEGREE where their cells overflow one advice column: `flex.configure` takes the column counts
    GateThreadBuilder::config would (range LOOKUP_BITS 4 at DEGREE 5: 3 gate + 1 lookup-advice column; LOOKUP_BITS 3 at DEGREE 6:
    2 + 1; poseidon at DEGREE 11: 4 gate columns), the product's incremental layout equals the oracle's closed-form one cell for
    cell and constrain_equal for constrain_equal, keys and proof bytes equal the oracle's (live at DEGREE <= 6) and the committed
    golden (tests/golden/flex_multi_proofs.json), and the oracle's verifier accepts / rejects as it should.  This is an ORACLE
    SELF-CHECK, not parity with the crate: the multi-column layout (break-point rule, constrain_equal order) is restated from memory
    of halo2-base 0.3, which is not under /root/reference, and the golden was made by this repository's oracle.  The quotient runs
    through the general kernel (h2mi_plonk_evaluate_h_flex_dev): one gate per column, one lookup per lookup-advice column."""
    import json
    import os

    from halo2_scaffold_amd import flex, poseidon

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", "flex_multi_proofs.json")))
    secret = int(g["srs_secret"], 16)
    for case in g["cases"]:
        shape, k, bits, x, seed = case["shape"], case["k"], case["lookup_bits"], int(case["x"], 16), case["seed"]
        closure = ((lambda cs: flex.range_closure(cs, x, bits)) if shape == "range" else (lambda cs: poseidon.hash_two_closure(cs, x, x + 1)))
        cs = flex.configure(shape == "range", k, closure)
        assert (cs.num_advice, cs.num_lookup_advice) == (case["num_advice"], case["num_lookup_advice"]) and cs.num_advice > 1
        asg = closure(cs)
        flex.mock(asg)
        assert asg.instance == [int(v, 16) for v in case["instance"]]
        params = gpu.ParamsKZG.setup(k, secret)
        keys = flex.FlexKeys(params, cs, asg)
        assert keys.vk_bytes().hex() == case["vk_bytes"], (shape, k)
        trace = {}
        proof = flex.create_proof(params, keys, asg, seed, trace=trace)
        assert proof.hex() == case["proof"], (shape, k)
        ocs = FX.flex_multi_cs(shape == "range", cs.num_advice, cs.num_lookup_advice)
        if shape == "range":
            oasg = FX.range_assignment_multi(ocs, x, bits, k)
            assert asg.advice == oasg.advice and asg.copies == oasg.copies and [asg.instance] == oasg.instance
            assert [dict(c) for i, c in enumerate(asg.fixed) if i != cs.col_table] == [dict(c) for i, c in enumerate(oasg.fixed) if i != ocs.col_table]
            okeys = _oracle_keys(ocs, k, oasg)
            _check_keys(keys, okeys, 1 << k)
            want = FX.prove(okeys, oasg, seed)
            for name in ("theta", "beta", "gamma", "y", "x"):
                assert trace[name] == want[name], name
            assert proof == want["proof"]
            assert FX.verify(okeys, proof, [asg.instance]) and not FX.verify(okeys, proof, [[x ^ 1]])
            # a second proof through a reused workspace, other witness and seed: verified, different bytes
            ws = flex.FlexWorkspace(params, keys)
            asg2 = flex.range_closure(cs, x ^ 0xFFFF, bits)
            p2 = flex.create_proof(params, keys, asg2, seed + 1, ws=ws)
            assert p2 != proof and FX.verify(okeys, p2, [asg2.instance])
            assert flex.create_proof(params, keys, asg, seed, ws=ws) == proof
            ws.release()
        keys.release()
        params.release()
        # the C++ host (include/h2mi_flex.hpp: its own Context / configure / break-point layout over the same prover ABI): same bytes
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "-s"])
        r = subprocess.run([os.path.join(root, "examples", "halo2_lib"), shape, str(k), str(bits), str(x), hex(secret), str(seed)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1000:]
        out = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.startswith(("vk ", "proof ", "columns ")))
        assert out["columns"] == f"{cs.num_advice} gate + {cs.num_lookup_advice} lookup-advice"
        assert out["vk"] == case["vk_bytes"] and out["proof"] == case["proof"], ("C++", shape, k)
    # the column count the crate's formula gives can be too small (a gate never straddles two columns): same failure as halo2-base
    cs = flex.configure(False, 4, lambda c: flex.halo2_lib_closure(c, 12))
    assert cs.num_advice == 3
    with pytest.raises(ValueError, match="NOT ENOUGH ADVICE COLUMNS"):
        flex.halo2_lib_closure(cs, 12)


def test_wide_column_counts_match_golden(gpu):
    """round 5: the prover ABI's column limits (32 gate columns, 8 lookup-advice columns, 64 permutation columns, 192 queries) —
    poseidon at DEGREE 8 / 9 (31 / 15 gate columns through `flex.configure`), 8 and 24 range checks in one context at DEGREE 6 / 7
    (8 + 3 and 11 + 4 columns), 10 range checks over 11 + 8 columns set by hand (eight lookup arguments), one over 5 + 2 columns with
    TWO constants columns (the constants dealt out round-robin); halo2_lib and poseidon under the RANGE builder (LOOKUP_BITS set, nothing
    looked up: no lookup-advice column, the table committed but never queried).  Keys and proof bytes equal
    the committed golden (tests/golden/flex_wide_proofs.json, made by the oracle's vector engine and re-verified on the CPU by
    tests/test_oracle_fast.py); the oracle's verifier accepts the device proof and refuses another public input; the C++ host
    (its own Context / configure over the same prover ABI) prints the same bytes.  A configuration whose constants overflow the
    constants column's usable rows is refused by keygen with H2MI_ERANGE — halo2's NotEnoughRowsAvailable — instead of yielding a
    proof that cannot verify.  ORACLE SELF-CHECK like the test above: the layouts are restated from memory of halo2-base."""
    import json
    import os
    import subprocess
    import sys

    from halo2_scaffold_amd import flex, poseidon

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_flex_wide_golden as MW

    g = json.load(open(os.path.join(root, "tests", "golden", "flex_wide_proofs.json")))
    secret = int(g["srs_secret"], 16)
    subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "-s"])
    for case in g["cases"]:
        shape, k, bits, x, seed, count = case["shape"], case["k"], case["lookup_bits"], int(case["x"], 16), case["seed"], case["count"]
        base, _, builder = shape.partition("+")
        range_builder = base == "range" or builder == "range_builder"  # the reference takes the Range builder whenever LOOKUP_BITS is set
        if base == "range":
            closure = lambda cs: flex.range_closure(cs, x, bits, count)
        else:
            plain = (lambda cs: poseidon.hash_two_closure(cs, x, x + 1)) if base == "poseidon" else (lambda cs: flex.halo2_lib_closure(cs, x))
            closure = (lambda cs: flex.load_lookup_table(plain(cs), bits)) if range_builder else plain
        if case["explicit"]:
            cs = flex.FlexGateCS(True, case["num_advice"], case["num_lookup_advice"], k=k, num_fixed=case["num_fixed"])
        else:
            cs = flex.configure(range_builder, k, closure)
        assert (cs.num_advice, cs.num_lookup_advice, cs.num_fixed) == (case["num_advice"], case["num_lookup_advice"], case["num_fixed"])
        asg = closure(cs)
        flex.mock(asg)
        assert asg.instance == [int(v, 16) for v in case["instance"]]
        params = gpu.ParamsKZG.setup(k, secret)
        keys = flex.FlexKeys(params, cs, asg)
        assert keys.vk_bytes().hex() == case["vk_bytes"], (shape, k)
        ws = flex.FlexWorkspace(params, keys)
        proof = flex.create_proof(params, keys, asg, seed, ws=ws)
        assert proof.hex() == case["proof"], (shape, k, cs.num_advice, cs.num_lookup_advice)
        assert flex.create_proof(params, keys, asg, seed, ws=ws) == proof  # the workspace is reusable at these widths too
        ocs, oasg = MW.build(shape, k, bits, x, count, (cs.num_advice, cs.num_lookup_advice, cs.num_fixed) if case["explicit"] else None)
        vk = FX.VerifierKeys(ocs, k, secret, oasg.fixed, oasg.copies)
        assert FX.verify(vk, proof, oasg.instance) and not FX.verify(vk, proof, [[v ^ 1 for v in oasg.instance[0]]])
        ws.release()
        keys.release()
        params.release()
        argv = [os.path.join(root, "examples", "halo2_lib"), base, str(k), str(bits), str(x), hex(secret), str(seed), str(max(count, 1))]
        if case["explicit"]:
            argv += [str(cs.num_advice), str(cs.num_lookup_advice), str(cs.num_fixed)]
        env = dict(os.environ, LOOKUP_BITS=str(bits)) if builder else {k_: v for k_, v in os.environ.items() if k_ != "LOOKUP_BITS"}
        r = subprocess.run(argv, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-1000:]
        out = dict(l.split(" ", 1) for l in r.stdout.splitlines() if l.startswith(("vk ", "proof ", "columns ")))
        assert out.get("columns", "1 gate + 0 lookup-advice") == f"{cs.num_advice} gate + {cs.num_lookup_advice} lookup-advice"
        assert out["vk"] == case["vk_bytes"] and out["proof"] == case["proof"], ("C++", shape, k)
    # 32 limb bases (LOOKUP_BITS 2) do not fit the 25 usable rows of a DEGREE-5 constants column, and config's ceil(32 / 2^5) says one
    # column (the golden's last case sets two by hand)
    cs = flex.configure(True, 5, lambda c: flex.range_closure(c, 0xDEADBEEFCAFE1234, 2))
    assert (cs.num_advice, cs.num_lookup_advice, cs.num_fixed) == (5, 2, 1)
    asg = flex.range_closure(cs, 0xDEADBEEFCAFE1234, 2)
    params = gpu.ParamsKZG.setup(5, secret)
    with pytest.raises(RuntimeError, match="(?i)range|rows"):
        flex.FlexKeys(params, cs, asg)
    params.release()


def test_scaffold_functions_mirror_the_reference(gpu, tmp_path, monkeypatch):
    """halo2_scaffold_amd.scaffold = the reference's src/scaffold.rs name for name (mock :39, gen_key :95, prove_private :158, prove :246):
    closures f(ctx, input, make_public) like its examples, DEGREE / LOOKUP_BITS / MINIMUM_ROWS from the environment, the fixed-seed SRS of
    gen_srs cached under PARAMS_DIR, column counts from builder.config.  The examples/range.rs closure at DEGREE 6 (one column) and at
    DEGREE 5 (3 gate + 1 lookup-advice columns: gen_key's break points are the ones interop/probe expects from the real crate, [22, 21]),
    examples/halo2_lib.rs under the Gate builder and — LOOKUP_BITS set — under the Range builder: every proof accepted by the oracle's
    verifier against the closed-form verifying key, the public inputs returned as the reference returns them; an unsatisfied closure
    fails in mock as MockProver does."""
    import json
    import os

    from halo2_scaffold_amd import scaffold
    from halo2_scaffold_amd.params import gen_srs_secret

    monkeypatch.setenv("PARAMS_DIR", str(tmp_path))
    monkeypatch.delenv("MINIMUM_ROWS", raising=False)

    def range_example(ctx, x, make_public):  # examples/range.rs:10-34
        xc = ctx.load_witness(x)
        make_public.append(xc)
        ctx.range_check(xc, 64, ctx.lookup_bits)
        ctx.add(xc, xc)

    def halo2_lib_example(ctx, x, make_public):  # examples/halo2_lib.rs:14-60
        xc = ctx.load_witness(x)
        make_public.append(xc)
        x_sq = ctx.mul(xc, xc)
        make_public.append(ctx.add_constant(x_sq, 72))
        ctx.assign_region_last([("constant", 72), ("existing", xc), ("existing", xc), ("witness", x * x + 72)], [0])
        ctx.mul_add_constant(xc, xc, 72)

    s = gen_srs_secret()
    x = 0xDEADBEEFCAFE1234
    # examples/range.rs, one column
    monkeypatch.setenv("DEGREE", "6")
    monkeypatch.setenv("LOOKUP_BITS", "4")
    scaffold.mock(range_example, x)
    pk, bp = scaffold.gen_key(range_example, 0)
    assert bp == [[]] and os.path.exists(tmp_path / "kzg_bn254_6.srs")
    assert scaffold.prove_private(range_example, x, pk, bp) == [x]
    ocs = FX.flex_gate_cs(True)
    oasg = FX.range_assignment(ocs, x, 4, 64)
    vk = FX.VerifierKeys(ocs, 6, s, oasg.fixed, oasg.copies)
    assert pk.get_vk().transcript_repr == vk.transcript_repr
    first = pk.last_proof
    assert FX.verify(vk, first, [[x]]) and not FX.verify(vk, first, [[x + 1]])
    assert scaffold.prove_private(range_example, x, pk, bp) == [x] and pk.last_proof != first and FX.verify(vk, pk.last_proof, [[x]])  # fresh blinding
    pk.release()
    # ... over 3 gate + 1 lookup-advice columns: the break points the probe expects from the real crate
    monkeypatch.setenv("DEGREE", "5")
    pk, bp = scaffold.gen_key(range_example, 0)
    expect = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "recall_expectations.json")))
    assert bp == [expect["range_k5_bits4_break_points_phase0"]] == [[22, 21]]
    assert (pk.cs.num_advice, pk.cs.num_lookup_advice) == (3, 1)
    assert scaffold.prove_private(range_example, x, pk, bp) == [x]
    ocs = FX.flex_multi_cs(True, 3, 1)
    oasg = FX.range_assignment_multi(ocs, x, 4, 5)
    vk = FX.VerifierKeys(ocs, 5, s, oasg.fixed, oasg.copies)
    assert [fm.g1_to_bytes(c).hex() for c in vk.fixed_commitments] == expect["range_k5_bits4_fixed_commitments"]  # the same keys the probe pins
    assert FX.verify(vk, pk.last_proof, [[x]])
    pk.release()
    # examples/halo2_lib.rs: Gate builder, then (LOOKUP_BITS set) the Range builder
    monkeypatch.delenv("LOOKUP_BITS")
    proof, public = scaffold.prove(halo2_lib_example, 12, 3)
    assert public == [12, 12 * 12 + 72] and len(proof) == 864
    ocs = FX.flex_gate_cs(False)
    oasg = FX.halo2_lib_assignment(ocs, 12)
    assert FX.verify(FX.VerifierKeys(ocs, 5, s, oasg.fixed, oasg.copies), proof, [public])
    monkeypatch.setenv("DEGREE", "6")
    monkeypatch.setenv("LOOKUP_BITS", "4")
    proof, public = scaffold.prove(halo2_lib_example, 12, 3)
    ocs = FX.flex_gate_cs(True)
    oasg = FX.halo2_lib_assignment(ocs, 12)
    oasg.fixed[ocs.col_table] = {i: i for i in range(16)}
    assert public == [12, 216] and len(proof) == 992 and FX.verify(FX.VerifierKeys(ocs, 6, s, oasg.fixed, oasg.copies), proof, [public])

    # MockProver's verdicts: an unsatisfied gate; LOOKUP_BITS >= DEGREE
    def broken(ctx, x, make_public):
        c = ctx.load_witness(x)
        ctx.assign_region_last([("constant", 0), ("existing", c), ("existing", c), ("witness", x * x + 1)], [0])

    with pytest.raises(ValueError, match="gate not satisfied"):
        scaffold.mock(broken, 5)
    monkeypatch.setenv("LOOKUP_BITS", "6")
    with pytest.raises(AssertionError, match="LOOKUP_BITS needs to be less than DEGREE"):
        scaffold.mock(range_example, x)


def test_general_quotient_kernel_agrees_with_the_specialised_one(gpu):
    """k_evaluate_h_flex (every operand converted to the multiplier's radix, Horner in y as the oracle writes it) and
    k_evaluate_h_range (level bookkeeping, shared reductions) are two independent implementations of the same function of their
    inputs: on RANDOM cosets — single gate, three permutation columns, the selector form of the lookup input and the lookup-advice
    form, and the Gate shape without a lookup — they must produce the same h, element for element."""
    from halo2_scaffold_amd import plonk as gp
    from halo2_scaffold_amd import synth
    from halo2_scaffold_amd.device import DevBuf

    k = 9
    for lookup, selector_form in ((False, False), (True, True), (True, False)):
        degree = 5 if selector_form else 4 if lookup else 3
        d = gpu.EvaluationDomain(degree, k)
        ext = d.extended_len()
        rnd = lambda i: DevBuf.from_numpy(synth.uniform_fr(ext, 4000 + i))
        a, q, table, la, ql = rnd(0), rnd(1), rnd(2), rnd(3), rnd(4)
        chunk = degree - 2
        perm_v, perm_s = [rnd(10 + j) for j in range(3)], [rnd(20 + j) for j in range(3)]
        zs = [rnd(30 + j) for j in range(-(-3 // chunk))]
        pin, ptab, lz = rnd(40), rnd(41), rnd(42)
        l0, ll, lact = rnd(50), rnd(51), rnd(52)
        beta, gamma, y = 0x1234567 ^ k, 0xABCDEF01, 0x777 + degree
        h1, h2 = DevBuf(ext * 32), DevBuf(ext * 32)
        gp.evaluate_h_range(d, a, (la if lookup and not selector_form else None), q, table if lookup else None, perm_v, perm_s, zs,
                            pin if lookup else None, ptab if lookup else None, lz if lookup else None, l0, ll, lact, beta, gamma, y, h1,
                            blinding_factors=6, lookup_selector=ql if selector_form else None, chunk_len=chunk)
        lookups = [((ql if selector_form else la), (a if selector_form else None), table, pin, ptab, lz)] if lookup else []
        gp.evaluate_h_flex(d, [(a, q)], perm_v, perm_s, zs, chunk, lookups, l0, ll, lact, beta, gamma, y, h2, blinding_factors=6)
        assert np.array_equal(h1.to_numpy(shape=(ext, 4)), h2.to_numpy(shape=(ext, 4))), (lookup, selector_form)
        for b in [a, q, table, la, ql, pin, ptab, lz, l0, ll, lact, h1, h2] + perm_v + perm_s + zs:
            b.free()
