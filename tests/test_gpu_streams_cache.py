"""GPU tests of the library's host-side plumbing: the bounded power-table cache under eviction, caller-provided
streams interleaved with the library's own, and the small device helpers the prover pipeline uses.
Every result is checked against the oracle (oracle/bn254.py); nothing is compared with itself."""
import ctypes as C

import numpy as np
import pytest

from oracle import bn254 as o

pytestmark = pytest.mark.gpu


def _fr(h2, v):
    return h2.field.fr_to_mont_limbs(v)


def test_power_table_cache_survives_eviction(gpu):
    """More distinct evaluation points than the cache holds (64 entries): eval_polynomial and kate_division
    key their power tables by the point, so a long-running prover walks through the eviction path.  Every
    result is compared with the oracle — a table freed while a call still holds its pointer (the b / b^-1
    pair of kate_division, the plan + coset table of an NTT) would show up as a wrong value."""
    h2 = gpu
    from halo2_scaffold_amd import arithmetic as A

    n = 3000
    coeffs = o.random_field_limbs(n, o.SEED + 5)
    cv = o.unpack(coeffs, o.R)
    dom = h2.EvaluationDomain(3, 11)
    lag = o.random_field_limbs(1 << 11, o.SEED + 6)
    want_ntt = o.Domain(11, 3).coeff_to_extended(o.unpack(lag, o.R))
    rng = np.random.default_rng(11)
    for it in range(90):
        pt = int.from_bytes(rng.bytes(32), "little") % o.R or 1
        got = A.eval_polynomial(coeffs, _fr(h2, pt))
        assert o.unpack(got.reshape(1, 4), o.R)[0] == o.eval_polynomial(cv, pt), it
        q = A.kate_division(coeffs, _fr(h2, pt))
        assert o.unpack(q, o.R) == o.kate_division(cv, pt), it
        if it % 30 == 29:  # a transform between evictions: plan and coset tables are rebuilt and still right
            assert o.unpack(dom.coeff_to_extended(lag), o.R) == want_ntt


def test_random_fill_add_head_match_host(gpu):
    """h2mi_fr_random_dev reproduces synth.uniform_fr bit for bit (the oracle prover draws the same stream);
    h2mi_fr_fill_dev / h2mi_fr_add_head_dev against plain integer arithmetic."""
    h2 = gpu
    from halo2_scaffold_amd import synth
    from halo2_scaffold_amd.device import DevBuf

    lib = h2.lib
    n = 5000
    d = DevBuf(n * 32)
    for seed, start in [(synth.SEED, 0), (7, 123), (0xFFFFFFFF, 1 << 20)]:
        assert lib.h2mi_fr_random_dev(d.ptr, n, seed, start, None) == 0
        assert np.array_equal(d.to_numpy(shape=(n, 4)), synth.uniform_fr(n, seed, start))
    assert lib.h2mi_fr_random_dev(d.ptr, n, 1 << 32, 0, None) == h2.lib.h2mi_fr_random_dev(d.ptr, 0, 1, 0, None) == -1
    v = 0x1234567890ABCDEF1234567890ABCDEF % o.R
    vl = _fr(h2, v)
    assert lib.h2mi_fr_fill_dev(d.ptr, n, vl.ctypes.data, None) == 0
    assert o.unpack(d.to_numpy(shape=(n, 4)), o.R) == [v] * n
    head = [o.R - 1, 5, 0, o.R - v]
    hl = np.ascontiguousarray(np.stack([_fr(h2, x) for x in head]))
    assert lib.h2mi_fr_add_head_dev(d.ptr, hl.ctypes.data, len(head), None) == 0
    got = o.unpack(d.to_numpy(shape=(n, 4)), o.R)
    assert got[:4] == [(v + x) % o.R for x in head] and got[4:] == [v] * (n - 4)
    d.free()


def test_fold_groups_and_normalize_on_device(gpu):
    """the per-phase combine of the sliced multi-GPU MSM without leaving HBM: fold (world, k) Jacobian points and
    normalise, both device to device, against oracle point additions (identity and P + (-P) included)."""
    h2 = gpu
    from halo2_scaffold_amd.device import DevBuf

    lib = h2.lib
    world, k = 5, 7
    rng = np.random.default_rng(3)
    pts = [[o.g1_mul(int(rng.integers(1, 1 << 60)), o.G1_GEN) for _ in range(k)] for _ in range(world)]
    pts[1][2] = None
    pts[3][4] = o.g1_neg(pts[0][4])
    pts[2][4] = pts[4][4] = pts[1][4] = None  # slot 4 sums to the identity
    jac = np.zeros((world, k, 12), dtype=np.uint64)
    for r in range(world):
        for j in range(k):
            jac[r, j] = o.pack_jacobian(pts[r][j], z=int(rng.integers(2, 1 << 60)))
    want = []
    for j in range(k):
        acc = None
        for r in range(world):
            acc = o.g1_add(acc, pts[r][j])
        want.append(acc)
    assert want[4] is None
    d_in, d_out, d_aff = DevBuf.from_numpy(jac), DevBuf(k * 96), DevBuf(k * 64)
    assert lib.h2mi_g1_fold_groups_dev(d_in.ptr, world, k, d_out.ptr, None) == 0
    assert lib.h2mi_g1_batch_normalize_dev(d_out.ptr, k, d_aff.ptr, None) == 0
    got_j = d_out.to_numpy(shape=(k, 12))
    assert [o.unpack_jacobian(got_j[j]) for j in range(k)] == want
    assert o.unpack_points(d_aff.to_numpy(shape=(k, 8))) == want
    for b in (d_in, d_out, d_aff):
        b.free()


def test_caller_streams_interleaved_with_library_stream(gpu):
    """MSMs, NTTs and polynomial helpers issued alternately on the library's stream (NULL) and on two
    caller-provided streams, sharing bases handles (so workspace slots change hands between streams), cached
    plans / power tables (built on one stream, read on another) and the shared scratch vector.  Results against
    the oracle / the C restatement."""
    import torch

    h2 = gpu
    from halo2_scaffold_amd.device import DevBuf
    from oracle import cref

    lib = h2.lib
    k = 12
    n = 1 << k
    params = h2.ParamsKZG.setup(k, 0xC0FFEE)
    bases = params.get_g()
    streams = [None, torch.cuda.Stream(), torch.cuda.Stream()]
    sp = [None if s is None else C.c_void_p(s.cuda_stream) for s in streams]
    dom = h2.EvaluationDomain(3, k)
    wl = _fr(h2, dom.omega)
    scal = [o.random_field_limbs(n, o.SEED + 50 + i) for i in range(12)]
    d_scal = [DevBuf.from_numpy(s) for s in scal]
    d_out = DevBuf(96 * len(scal))
    d_ntt = [DevBuf.from_numpy(s) for s in scal]
    d_ev = DevBuf(32 * len(scal))
    pt = 0xABCDEF0123456789 % o.R
    for rnd in range(2):  # second round: every slot, plan and table is reused from another stream than it was built on
        for i in range(len(scal)):
            s = sp[(i + rnd) % 3]
            assert lib.h2mi_msm_bn254_g1_dev(params.g_handle, d_scal[i].ptr, n, d_out.ptr + 96 * i, s) == 0
            s2 = sp[(i + rnd + 1) % 3]
            if rnd == 0:
                assert lib.h2mi_ntt_bn254_fr_dev(d_ntt[i].ptr, k, wl.ctypes.data, None, None, s2) == 0
            pl = _fr(h2, pt + i)
            assert lib.h2mi_fr_eval_poly_dev(d_scal[i].ptr, n, pl.ctypes.data, d_ev.ptr + 32 * i, s2) == 0
        for s in streams[1:]:
            s.synchronize()
        assert lib.h2mi_sync() == 0
        got = d_out.to_numpy(shape=(len(scal), 12))
        ev = d_ev.to_numpy(shape=(len(scal), 4))
        for i in range(len(scal)):
            assert o.unpack_jacobian(got[i]) == o.unpack_jacobian(cref.msm(scal[i], bases, 4)), (rnd, i)
            assert o.unpack(ev[i : i + 1], o.R)[0] == o.eval_polynomial(o.unpack(scal[i], o.R), (pt + i) % o.R)
    w = o.omega_for(k)
    for i in (0, 5, 11):
        assert o.unpack(d_ntt[i].to_numpy(shape=(n, 4)), o.R) == o.ntt(o.unpack(scal[i], o.R), w)
    for b in d_scal + d_ntt + [d_out, d_ev]:
        b.free()
    params.release()


@pytest.mark.parametrize("k", [12, 16])
def test_msm_dominant_value_columns(gpu, k):
    """columns that are one value repeated (what a permutation grand product looks like in a padded circuit: constant
    wherever a row takes part in no copy constraint): the MSM subtracts the majority value and adds value * (sum of
    the bases) instead — same group element as the C restatement's Pippenger, for every mix; a partial-length MSM
    (n below the registered count, where the shift must stay off) too."""
    import ctypes as C

    from oracle import cref

    h2 = gpu
    n = 1 << k
    params = h2.ParamsKZG.setup(k, 0xD0D0 + k)
    bases = params.get_g_lagrange()
    rnd = o.random_field_limbs(n, 600 + k)
    c1 = o.pack([0x123456789ABCDEF0123456789ABCDEF % o.R], o.R)[0]
    cols = {}
    cols["constant"] = np.tile(c1, (n, 1))
    pc = np.tile(c1, (n, 1))
    pc[:3] = rnd[:3]
    pc[n - 5 :] = rnd[n - 5 :]
    cols["grand-product-like"] = pc
    two = np.tile(c1, (n, 1))
    two[n // 3 :] = o.pack([o.R - 1], o.R)[0]  # two long runs: the majority (2/3) value is shifted away
    cols["two runs"] = two
    mix = rnd.copy()
    mix[::3] = c1  # a third of the rows: no majority, nothing is shifted
    cols["minority"] = mix
    cols["ones"] = np.tile(o.pack([1], o.R)[0], (n, 1))
    cols["minus one"] = np.tile(o.pack([o.R - 1], o.R)[0], (n, 1))
    for name, sc in cols.items():
        sc = np.ascontiguousarray(sc)
        got = params.commit_lagrange(sc)
        assert o.unpack_jacobian(got) == o.unpack_jacobian(cref.msm(sc, bases, 4)), name
    part = np.ascontiguousarray(cols["constant"][: n - 7])
    out = np.zeros(12, dtype=np.uint64)
    assert h2.lib.h2mi_msm_bn254_g1(params.g_lagrange_handle, None, part.ctypes.data, n - 7, out.ctypes.data) == 0
    assert o.unpack_jacobian(out) == o.unpack_jacobian(cref.msm(part, bases[: n - 7], 4))
    # the shift really is in effect: a constant column costs a handful of bucket insertions, not n * windows
    ba, ra = C.c_uint64(), C.c_uint64()
    params.commit_lagrange(np.ascontiguousarray(cols["grand-product-like"]))
    assert h2.lib.h2mi_msm_last_stats(params.g_lagrange_handle, C.byref(ba), C.byref(ra)) == 0
    if k >= 15:
        assert ba.value < 64 * 20, ba.value
    else:  # base sets up to 2^14 take the bucketless small path (round 4): no bucket can run hot there, so nothing is shifted —
        assert 0 < ba.value <= n * 128, ba.value  # a constant column costs what a uniform one costs, one gather per non-zero digit
    params.release()


def test_adhoc_bases_are_cached_not_rebuilt(gpu):
    """best_multiexp(coeffs, bases) with a plain slice of bases (handle = 0): the first call registers them, the
    following calls with the same bytes find the registration by its device-side fingerprint and build nothing;
    changing one limb of one point is a different set (re-registered, right result); results against the C oracle."""
    import ctypes as C

    from oracle import cref

    h2 = gpu
    n = 5000
    bases = cref.g1_mul_gen(o.random_field_limbs(n, 71), 2)
    sc = [o.random_field_limbs(n, 72 + i) for i in range(3)]
    builds = C.c_uint64()

    def nbuilds():
        assert h2.lib.h2mi_msm_adhoc_builds(C.byref(builds)) == 0
        return builds.value

    b0 = nbuilds()
    for s in sc:
        assert o.unpack_jacobian(h2.best_multiexp(s, bases)) == o.unpack_jacobian(cref.msm(s, bases, 2))
    assert nbuilds() == b0 + 1  # three commits against the same slice: one registration
    other = bases.copy()
    other[n // 2] = bases[0]  # still valid curve points, different bytes
    assert o.unpack_jacobian(h2.best_multiexp(sc[0], other)) == o.unpack_jacobian(cref.msm(sc[0], other, 2))
    assert nbuilds() == b0 + 2
    assert o.unpack_jacobian(h2.best_multiexp(sc[1], bases)) == o.unpack_jacobian(cref.msm(sc[1], bases, 2))
    assert nbuilds() == b0 + 2  # the first set is still cached
    # more distinct sets than the cache holds: the least recently used registration is dropped, nothing leaks or breaks
    for j in range(5):
        bj = np.ascontiguousarray(bases[j + 1 : j + 1 + 1000])
        sj = np.ascontiguousarray(sc[0][:1000])
        assert o.unpack_jacobian(h2.best_multiexp(sj, bj)) == o.unpack_jacobian(cref.msm(sj, bj, 1))
    assert nbuilds() == b0 + 7


def test_side_streams_run_transforms_and_divisions_beside_the_library_stream(gpu):
    """h2mi_stream_create / h2mi_stream_wait: six side streams (more than the library keeps scratch vectors for, so the
    least-recently-used scratch is handed on behind its last user) each run lagrange_to_coeff -> coeff_to_extended of
    their own column and a kate_division, interleaved with the same work on the library stream; every result equals the
    oracle's.  Ordering across streams is only what after_library() / join_library() state."""
    h2 = gpu
    from halo2_scaffold_amd.device import DevBuf, SideStream

    k = 12
    n = 1 << k
    dom = h2.EvaluationDomain(3, k)
    odom = o.Domain(k, 3)
    ext = dom.extended_len()
    sides = [SideStream() for _ in range(6)]
    lanes = [None] + [s.handle for s in sides]
    cols, polys, cosets, quots, want = [], [], [], [], []
    for i in range(len(lanes)):
        vals = o.random_field_limbs(n, o.SEED + 100 + i)
        cols.append(DevBuf.from_numpy(vals))
        polys.append(DevBuf(n * 32))
        cosets.append(DevBuf(ext * 32))
        quots.append(DevBuf(n * 32))
        coeffs = odom.lagrange_to_coeff(o.unpack(vals, o.R))
        want.append((coeffs, odom.coeff_to_extended(coeffs), o.kate_division(coeffs, 1234567 + i)))
    for s in sides:
        s.after_library()  # the uploads above went through the library stream
    for rnd in range(2):  # twice: the second round reuses scratches that changed hands in the first
        for i, st in enumerate(lanes):
            dom.lagrange_to_coeff_oop_dev(cols[i], polys[i], stream=st)
            dom.coeff_to_extended_oop_dev(polys[i], cosets[i], stream=st)
            b, b_inv = _fr(h2, 1234567 + i), _fr(h2, pow(1234567 + i, -1, o.R))
            assert h2.lib.h2mi_fr_kate_division_dev(polys[i].ptr, n, b.ctypes.data, b_inv.ctypes.data, quots[i].ptr, st) == 0
    for s in sides:
        s.join_library()
    for i in range(len(lanes)):
        coeffs, coset, quot = want[i]
        assert o.unpack(polys[i].to_numpy(shape=(n, 4)), o.R) == coeffs, i
        assert o.unpack(cosets[i].to_numpy(shape=(ext, 4)), o.R) == coset, i
        assert o.unpack(quots[i].to_numpy(shape=(n, 4)), o.R)[: n - 1] == quot, i
    for b in cols + polys + cosets + quots:
        b.free()
    for s in sides:
        s.free()
    # waiting on oneself and on the library stream from the library stream are no-ops, a null stream handle is refused
    assert h2.lib.h2mi_stream_wait(None, None) == 0
    assert h2.lib.h2mi_stream_destroy(None) != 0


def test_fr_mul_elementwise_matches_oracle(gpu):
    """h2mi_fr_mul_dev (the row values of a product expression such as q_lookup * a): out of place, in place, odd lengths,
    zeros and p - 1; argument checking"""
    h2 = gpu
    for n in (1, 257, 5000):
        a = o.unpack(o.random_field_limbs(n, o.SEED + 40), o.R)
        b = o.unpack(o.random_field_limbs(n, o.SEED + 41), o.R)
        a[0], b[0] = o.R - 1, o.R - 1
        if n > 2:
            a[1], b[2] = 0, 0
        da, db, dc = h2.DevBuf.from_numpy(o.pack(a, o.R)), h2.DevBuf.from_numpy(o.pack(b, o.R)), h2.DevBuf(n * 32)
        assert h2.lib.h2mi_fr_mul_dev(da.ptr, db.ptr, n, dc.ptr, None) == 0
        want = [x * y % o.R for x, y in zip(a, b)]
        assert o.unpack(dc.to_numpy(shape=(n, 4)), o.R) == want
        assert h2.lib.h2mi_fr_mul_dev(da.ptr, db.ptr, n, da.ptr, None) == 0  # in place
        assert o.unpack(da.to_numpy(shape=(n, 4)), o.R) == want
        for buf in (da, db, dc):
            buf.free()
    assert h2.lib.h2mi_fr_mul_dev(None, None, 4, None, None) != 0


_REINIT_WORKER = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
import torch  # first: one HIP runtime
import _load_pkg
h2 = _load_pkg.load()
from oracle import bn254 as o, cref
from halo2_scaffold_amd import flex
lib = h2.lib
def one_round(tag):
    h2.init(0)
    k = 10
    params = h2.ParamsKZG.setup(k, 0xBEEF + tag)
    coeffs = o.random_field_limbs(1 << k, 7 + tag)
    dom = h2.EvaluationDomain(3, k)
    evals = dom.coeff_to_lagrange(coeffs)
    a = coeffs.copy(); cref.ntt(a, h2.field.fr_to_mont_limbs(h2.field.omega_for(k)), k, 1)
    assert (evals == a).all(), "ntt after re-init"
    c1, c2 = o.unpack_jacobian(params.commit(coeffs)), o.unpack_jacobian(params.commit_lagrange(evals))
    assert c1 == c2 == o.unpack_jacobian(cref.msm(coeffs, params.get_g(), 4)), "msm after re-init"
    cs = flex.FlexGateCS(lookup=True)              # lookup scratch, side streams, sparse grand products
    asg = flex.range_closure(cs, 0x1234567 + tag, 6)
    keys = flex.FlexKeys(params, cs, asg)
    proof = flex.create_proof(params, keys, asg, 3)
    keys.release()
    lib.h2mi_shutdown()                            # with params' two registrations still alive: the teardown releases them                            # handles, plans, tables, scratch: all released with the device
    assert lib.h2mi_malloc(16, None) != 0          # nothing works until the next init
    return proof
p1, p2, p3 = one_round(0), one_round(0), one_round(1)
assert p1 == p2 and p1 != p3
print("REINIT_OK")
"""


def test_shutdown_releases_module_state_and_reinit_works(gpu, tmp_path):
    """the NTT plan / table / scratch caches, the MSM registrations and the lookup scratch were process-global
    statics that survived h2mi_shutdown; a later h2mi_init then reused stale device pointers and events.  h2mi_shutdown now
    runs per-module teardown hooks: three init -> prove -> shutdown rounds in one process give the same proofs."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "reinit.py"
    script.write_text(_REINIT_WORKER.format(root=root))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0 and "REINIT_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_patch_cells_writes_every_cell_once(gpu):
    """h2mi_fr_patch_cells_dev: field elements carried in kernel arguments to scattered device addresses — counts below, at and above
    the 64 cells of one launch, two destination buffers, the rest of the buffers untouched; argument checks."""
    import ctypes as C

    from halo2_scaffold_amd.device import DevBuf
    from oracle import bn254 as o

    n = 1000
    base_a, base_b = o.random_field_limbs(n, 41), o.random_field_limbs(n, 42)
    for count in (0, 1, 5, 64, 65, 200):
        a, b = DevBuf.from_numpy(base_a), DevBuf.from_numpy(base_b)
        rng = np.random.default_rng(count)
        rows = rng.choice(2 * n, size=count, replace=False)
        vals = o.random_field_limbs(max(count, 1), 43 + count)[:count]
        addrs = [(a.ptr if r < n else b.ptr) + int(r % n) * 32 for r in rows]
        ptrs = (C.c_void_p * max(count, 1))(*addrs) if count else None
        assert gpu.lib.h2mi_fr_patch_cells_dev(ptrs, np.ascontiguousarray(vals).ctypes.data if count else None, count, None) == 0
        want_a, want_b = base_a.copy(), base_b.copy()
        for r, v in zip(rows, vals):
            (want_a if r < n else want_b)[r % n] = v
        assert np.array_equal(a.to_numpy(shape=(n, 4)), want_a) and np.array_equal(b.to_numpy(shape=(n, 4)), want_b), count
        a.free()
        b.free()
    d = DevBuf.from_numpy(base_a)
    one = (C.c_void_p * 1)(d.ptr + 8)  # misaligned cell
    assert gpu.lib.h2mi_fr_patch_cells_dev(one, base_b.ctypes.data, 1, None) != 0
    assert gpu.lib.h2mi_fr_patch_cells_dev(None, base_b.ctypes.data, 1, None) != 0
    d.free()
