"""The reference's `src/scaffold.rs`, name for name, over the device prover: `mock`, `gen_key`, `prove_private`, `prove`.

The reference's examples are closures `f(ctx, input, make_public)` handed to these four functions (examples/halo2_lib.rs:62-71,
examples/range.rs:36-43, examples/linear_regression.rs:126-195); DEGREE, LOOKUP_BITS and MINIMUM_ROWS come from the environment
(src/scaffold.rs:44-50), the SRS from `gen_srs(k)` (read or created under PARAMS_DIR / ./params), the column counts from
`builder.config(k, Some(minimum_rows))`.  Here `ctx` is `flex.Context` (halo2-base's Context with the Gate / Range instructions as
methods: `ctx.mul`, `ctx.add`, `ctx.range_check(a, bits, lookup_bits)`, ...), `make_public` a list of its cells; the Range builder is
taken whenever LOOKUP_BITS is set, as in the reference.  Differences, by necessity: `prove_private` / `prove` do not run the crate's
`verify_proof` (there is no verifier in the product: the oracle's is test infrastructure) and return the proof bytes next to the
public inputs; blinding comes from a fresh 256-bit key per prover (`h2mi_prover_set_rng_key`), standing where the reference passes
`OsRng`.  Errors: an unsatisfied circuit raises ValueError from `mock` (MockProver's assert_satisfied), a lookup input outside the
table raises ValueError from the prover (ConstraintSystemFailure), "LOOKUP_BITS needs to be less than DEGREE" is asserted as there.
"""
import os

from . import flex
from .params import gen_srs


def _env():
    k = int(os.environ.get("DEGREE", "18"))
    lookup_bits = int(os.environ["LOOKUP_BITS"]) if "LOOKUP_BITS" in os.environ else None
    if lookup_bits is not None:
        assert lookup_bits < k, "LOOKUP_BITS needs to be less than DEGREE"
    return k, lookup_bits, int(os.environ.get("MINIMUM_ROWS", "9"))


def _synthesize(f, inputs, cs, lookup_bits):
    """run the closure in a fresh context over `cs` -> the assignment (cells, copies, public inputs[, the Range builder's table])"""
    asg = flex.Assignment(cs)
    ctx = flex.Context(asg)
    ctx.lookup_bits = lookup_bits  # what RangeChip::default(lookup_bits) would carry
    make_public = []
    f(ctx, inputs, make_public)
    ctx.finish(make_public)
    if cs.lookup:
        flex.load_lookup_table(asg, lookup_bits)
    return asg


def _config(f, inputs, k, lookup_bits, minimum_rows):
    return flex.configure(lookup_bits is not None, k, lambda cs: _synthesize(f, inputs, cs, lookup_bits), minimum_rows)


def mock(f, private_inputs):
    """src/scaffold.rs:39-93: configure the circuit for the closure, then MockProver::run(k, &circuit, vec![public]).assert_satisfied()"""
    k, lookup_bits, minimum_rows = _env()
    cs = _config(f, private_inputs, k, lookup_bits, minimum_rows)
    flex.mock(_synthesize(f, private_inputs, cs, lookup_bits), k)


class ProvingKey:
    """what gen_key hands to prove_private: the device-resident keys, the SRS they were made against and one reusable prover"""

    def __init__(self, params, keys):
        self.params, self.keys = params, keys
        self.workspace = flex.FlexWorkspace(params, keys)
        self.workspace.prover.set_rng_key(os.urandom(32))  # the reference passes OsRng
        self.proofs = 0
        self.last_proof = None

    @property
    def cs(self):
        return self.keys.cs

    def get_vk(self):
        return self.keys

    def release(self):
        self.workspace.release()
        self.keys.release()
        self.params.release()


def gen_key(f, dummy_inputs):
    """src/scaffold.rs:95-156: the circuit's shape from a run on dummy inputs (GateThreadBuilder::keygen + builder.config), the SRS
    from gen_srs(k), keygen_vk + keygen_pk -> (pk, break_points)"""
    k, lookup_bits, minimum_rows = _env()
    cs = _config(f, dummy_inputs, k, lookup_bits, minimum_rows)
    asg = _synthesize(f, dummy_inputs, cs, lookup_bits)
    params = gen_srs(k)
    pk = ProvingKey(params, flex.FlexKeys(params, cs, asg))
    pk.lookup_bits = lookup_bits
    return pk, [list(asg.break_points)]


def prove_private(f, private_inputs, pk: ProvingKey, break_points):
    """src/scaffold.rs:158-244: witness generation with the stored break points (GateThreadBuilder::prover), create_proof -> the public
    inputs (the proof bytes are kept in pk.last_proof)"""
    _, lookup_bits, _ = _env()
    asg = _synthesize(f, private_inputs, pk.cs, lookup_bits)
    if [list(asg.break_points)] != [list(b) for b in break_points]:
        raise ValueError("the circuit's break points differ from the ones stored at keygen: the closure's shape depends on its inputs")
    pk.proofs += 1
    pk.last_proof = flex.create_proof(pk.params, pk.keys, asg, pk.proofs, ws=pk.workspace)  # the seed is the keyed stream's per-proof nonce
    return list(asg.instance)


def prove(f, private_inputs, dummy_inputs):
    """src/scaffold.rs:246-366: keygen on dummy inputs, then one proof of the private inputs -> (proof bytes, public inputs)"""
    pk, break_points = gen_key(f, dummy_inputs)
    try:
        public_io = prove_private(f, private_inputs, pk, break_points)
        return pk.last_proof, public_io
    finally:
        pk.release()
