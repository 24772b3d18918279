"""halo2-scaffold_amd — MI355X (gfx950) backend for the Halo2/KZG prover hot path.

Host-side mirror (Python, over the C ABI of ``libh2mi.so``) of the interfaces the reference
reaches through ``create_proof`` (reference examples/standard_plonk.rs:41-49, src/scaffold.rs:322-331):

  arithmetic.best_multiexp / best_fft      halo2_proofs::arithmetic
  domain.EvaluationDomain                  halo2_proofs::poly::EvaluationDomain
  params.ParamsKZG                         halo2_proofs::poly::kzg::commitment::ParamsKZG
  replay.StandardPlonkReplay               the MSM/NTT sequence one StandardPlonk proof issues
  transcript.Blake2bWrite / Blake2bRead    halo2_proofs::transcript (Challenge255), serde: to_repr / to_bytes
  scaffold.mock / gen_key / prove_private / prove   the reference's src/scaffold.rs, name for name, over flex.* (the halo2-lib builders)

All arithmetic runs in hand-written HIP kernels; there is no CPU fallback — importing this package
without a built ``libh2mi.so`` raises, and compute calls without a GPU return H2MI_ENODEV.
The directory name contains a hyphen (it mirrors the reference's name); load it with
``_load_pkg.load()`` at the repo root, which registers it as module ``halo2_scaffold_amd``.
"""
from . import field  # noqa: F401
from ._lib import H2miError, lib, init, lib_path  # noqa: F401
from .device import DevBuf  # noqa: F401
from .arithmetic import best_fft, best_multiexp, eval_polynomial, kate_division, lincomb  # noqa: F401
from .domain import EvaluationDomain  # noqa: F401
from .params import ParamsKZG, gen_srs, gen_srs_secret  # noqa: F401
from . import serde, transcript  # noqa: F401
