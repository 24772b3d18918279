import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def h2():
    """the product package (loads libh2mi.so; raises if it is not built)."""
    import _load_pkg

    return _load_pkg.load()


@pytest.fixture(scope="session")
def gpu(h2):
    """initialised device context; GPU tests fail (not skip) if the library cannot reach a GPU."""
    import torch  # noqa: F401  (import first so one HIP runtime is shared)

    h2.init(0)
    return h2


@pytest.fixture(scope="session")
def hooks(gpu):
    """libh2mi_hooks.so — the product's objects + csrc/h2mi_hooks.hip: elementwise device arithmetic on host arrays (every field
    operation, the XYZZ point formulas, the lane-cooperative quad operations) for the parity tests.  The product library neither
    contains nor exports these; the hooks library is a separate instance with its own state, initialised on the same GPU."""
    import ctypes as C

    path = os.path.join(ROOT, "halo2-scaffold_amd", "libh2mi_hooks.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C halo2-scaffold_amd/csrc` (or __graft_entry__.build())")
    L = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    for name, args in (("h2mi_dbg_field_op", [C.c_int, C.c_int, vp, vp, vp, sz]), ("h2mi_dbg_g1_op", [C.c_int, vp, vp, vp, sz]),
                       ("h2mi_dbg_g1_quad_op", [C.c_int, vp, vp, vp, sz]), ("h2mi_init", [C.c_int])):
        fn = getattr(L, name)
        fn.argtypes, fn.restype = args, C.c_int
    L.h2mi_strerror.argtypes, L.h2mi_strerror.restype = [C.c_int], C.c_char_p
    assert L.h2mi_init(0) == 0
    return L
