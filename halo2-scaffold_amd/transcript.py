"""Mirror of halo2_proofs::transcript::{Blake2bWrite, Blake2bRead, Challenge255} for bn256::G1Affine
(reference call sites: Blake2bWrite::<_, G1Affine, Challenge255<_>>::init, examples/standard_plonk.rs:40-49,
src/scaffold.rs:190-199; Blake2bRead in the verify path, examples/standard_plonk.rs:56).

Restated [RECALL halo2_proofs v2023_02_02 transcript.rs]: Blake2b-512 with personalisation
"Halo2-Transcript"; every absorbed item is prefixed by one byte (0 challenge, 1 point, 2 scalar); a point
is absorbed as x.to_repr() || y.to_repr() (64 bytes) and written to the proof as its 32-byte compressed
form; a scalar is absorbed and written as to_repr(); a challenge is the 64-byte digest of a clone of the
running state (after absorbing the prefix byte 0) reduced mod r.  Host control plane: Python's hashlib."""
import hashlib
import io

import numpy as np

from . import field as F
from . import serde

PREFIX_CHALLENGE, PREFIX_POINT, PREFIX_SCALAR = b"\x00", b"\x01", b"\x02"
_RINV_Q = pow(1 << 256, -1, F.FQ_MODULUS)


def _fq_repr(limbs) -> bytes:
    v = sum(int(limbs[i]) << (64 * i) for i in range(4))
    return (v * _RINV_Q % F.FQ_MODULUS).to_bytes(32, "little")


class _Blake2bTranscript:
    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")

    def squeeze_challenge(self) -> np.ndarray:
        """Challenge255: the scalar, as 4 Montgomery limbs."""
        self.state.update(PREFIX_CHALLENGE)
        return serde.fr_from_bytes_wide(self.state.copy().digest())

    def common_point(self, affine) -> tuple:
        """absorbs x || y; returns their 32-byte encodings (the writer derives the compressed form from them)"""
        a = np.asarray(affine, dtype=np.uint64).reshape(8)
        if not a.any():
            raise ValueError("cannot write points at infinity to the transcript")
        xr, yr = _fq_repr(a[:4]), _fq_repr(a[4:])
        self.state.update(PREFIX_POINT)
        self.state.update(xr)
        self.state.update(yr)
        return xr, yr

    def common_scalar(self, limbs) -> None:
        self.state.update(PREFIX_SCALAR)
        self.state.update(serde.fr_to_repr(limbs))


class Blake2bWrite(_Blake2bTranscript):
    def __init__(self, writer=None):
        super().__init__()
        self.writer = writer if writer is not None else io.BytesIO()

    @classmethod
    def init(cls, writer=None) -> "Blake2bWrite":
        """TranscriptWriterBuffer::init, as the reference calls it: Blake2bWrite::<_, _, Challenge255<_>>::init(vec![])"""
        return cls(writer)

    def write_point(self, affine) -> None:
        # G1Affine::to_bytes on the host (a proof holds eleven points): x with the sign of y in bit 6 of the last byte
        xr, yr = self.common_point(affine)
        self.writer.write(xr[:31] + bytes([xr[31] | ((yr[0] & 1) << 6)]))

    def write_point_xy(self, x: int, y: int) -> None:
        """write_point for a point the caller already holds as canonical integers (G1::batch_normalize on the host)"""
        xr, yr = x.to_bytes(32, "little"), y.to_bytes(32, "little")
        self.state.update(PREFIX_POINT)
        self.state.update(xr)
        self.state.update(yr)
        self.writer.write(xr[:31] + bytes([xr[31] | ((yr[0] & 1) << 6)]))

    def write_scalar(self, limbs) -> None:
        self.common_scalar(limbs)
        self.writer.write(serde.fr_to_repr(limbs))

    def write_scalar_int(self, v: int) -> None:
        b = v.to_bytes(32, "little")
        self.state.update(PREFIX_SCALAR)
        self.state.update(b)
        self.writer.write(b)

    def finalize(self) -> bytes:
        return self.writer.getvalue()


class Blake2bRead(_Blake2bTranscript):
    def __init__(self, proof: bytes):
        super().__init__()
        self.reader = io.BytesIO(proof)

    @classmethod
    def init(cls, proof: bytes) -> "Blake2bRead":
        """TranscriptReadBuffer::init: Blake2bRead::<_, _, Challenge255<_>>::init(&proof[..])"""
        return cls(proof)

    def _take(self, n: int) -> bytes:
        b = self.reader.read(n)
        if len(b) != n:
            raise serde.DecodeError("proof too short")
        return b

    def read_point(self) -> np.ndarray:
        p = serde.g1_from_bytes(self._take(32))[0]
        self.common_point(p)
        return p

    def read_scalar(self) -> np.ndarray:
        s = serde.fr_from_repr(self._take(32))
        self.common_scalar(s)
        return s
