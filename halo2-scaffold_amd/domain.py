"""Mirror of halo2_proofs::poly::EvaluationDomain for Fr.

The reference never names EvaluationDomain itself; create_proof / keygen_pk build one from the circuit
(reference examples/standard_plonk.rs:33-34,41-49).  Method names and meanings follow the crate:
lagrange_to_coeff (iFFT * n^-1), coeff_to_extended (zero-pad, distribute powers of zeta, FFT over the
extended domain), extended_to_coeff (inverse, undo the coset, truncate).  The scaling sweeps are fused
into the NTT passes on the device (h2mi_ntt_ext_bn254_fr).
"""
import numpy as np

from . import field as F
from ._lib import check, lib
from .device import DevBuf


class EvaluationDomain:
    def __init__(self, j: int, k: int):
        """j = constraint-system degree, k = log2(rows)  (EvaluationDomain::new(j, k))."""
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ext = k
        while (1 << ext) < self.n * self.quotient_poly_degree:
            ext += 1
        self.extended_k = ext
        self.omega = F.omega_for(k)
        self.omega_inv = F.fr_inv(self.omega)
        self.extended_omega = F.omega_for(ext)
        self.extended_omega_inv = F.fr_inv(self.extended_omega)
        self.g_coset = F.FR_ZETA
        self.g_coset_inv = F.FR_ZETA * F.FR_ZETA % F.FR_MODULUS
        self.ifft_divisor = F.fr_inv(self.n)
        self.extended_ifft_divisor = F.fr_inv(1 << ext)
        m = F.fr_to_mont_limbs
        self._omega, self._omega_inv = m(self.omega), m(self.omega_inv)
        self._eomega, self._eomega_inv = m(self.extended_omega), m(self.extended_omega_inv)
        self._zeta, self._zeta_inv = m(self.g_coset), m(self.g_coset_inv)
        self._ninv, self._eninv = m(self.ifft_divisor), m(self.extended_ifft_divisor)

    def extended_len(self) -> int:
        return 1 << self.extended_k

    # ---- host-array forms (copy in, transform on the GPU, copy out) ----
    def lagrange_to_coeff(self, a: np.ndarray) -> np.ndarray:
        assert len(a) == self.n
        out = np.array(a, dtype=np.uint64, order="C", copy=True)
        check(lib.h2mi_ntt_ext_bn254_fr(out.ctypes.data, self.k, self._omega_inv.ctypes.data, None, self._ninv.ctypes.data), "ifft")
        return out

    def coeff_to_lagrange(self, a: np.ndarray) -> np.ndarray:
        assert len(a) == self.n
        out = np.array(a, dtype=np.uint64, order="C", copy=True)
        check(lib.h2mi_ntt_bn254_fr(out.ctypes.data, self._omega.ctypes.data, self.k), "fft")
        return out

    def coeff_to_extended(self, a: np.ndarray) -> np.ndarray:
        assert len(a) == self.n
        out = np.zeros((self.extended_len(), 4), dtype=np.uint64)
        out[: self.n] = a
        check(lib.h2mi_ntt_ext_bn254_fr(out.ctypes.data, self.extended_k, self._eomega.ctypes.data, self._zeta.ctypes.data, None), "coset fft")
        return out

    def extended_to_coeff(self, a: np.ndarray) -> np.ndarray:
        assert len(a) == self.extended_len()
        d = DevBuf.from_numpy(np.ascontiguousarray(a, dtype=np.uint64))
        self.extended_to_coeff_dev(d)
        out = d.to_numpy(shape=(self.extended_len(), 4))
        d.free()
        return out[: self.n * self.quotient_poly_degree].copy()

    # ---- device-resident forms: buffers stay in HBM ----
    def lagrange_to_coeff_dev(self, d: DevBuf, stream=None):
        check(lib.h2mi_ntt_bn254_fr_dev(d.ptr, self.k, self._omega_inv.ctypes.data, None, self._ninv.ctypes.data, stream), "ifft_dev")

    def coeff_to_extended_dev(self, d_ext: DevBuf, stream=None):
        """d_ext holds extended_len() elements: the n coefficients followed by zeros."""
        check(lib.h2mi_ntt_bn254_fr_dev(d_ext.ptr, self.extended_k, self._eomega.ctypes.data, self._zeta.ctypes.data, None, stream), "coset_fft_dev")

    # out-of-place forms: the source column stays intact (the prover still needs it), no clone, no zero padding
    def lagrange_to_coeff_oop_dev(self, d_lagrange: DevBuf, d_coeff: DevBuf, stream=None):
        check(lib.h2mi_ntt_bn254_fr_oop_dev(d_lagrange.ptr, self.n, d_coeff.ptr, self.k, self._omega_inv.ctypes.data, None,
                                            self._ninv.ctypes.data, stream), "ifft_oop_dev")

    def coeff_to_extended_oop_dev(self, d_coeff: DevBuf, d_ext: DevBuf, stream=None):
        """d_coeff: n coefficients; d_ext receives the extended_len() coset evaluations."""
        check(lib.h2mi_ntt_bn254_fr_oop_dev(d_coeff.ptr, self.n, d_ext.ptr, self.extended_k, self._eomega.ctypes.data,
                                            self._zeta.ctypes.data, None, stream), "coset_fft_oop_dev")

    def extended_to_coeff_dev(self, d_ext: DevBuf, stream=None):
        check(lib.h2mi_ntt_bn254_fr_dev(d_ext.ptr, self.extended_k, self._eomega_inv.ctypes.data, None, None, stream), "coset_ifft_dev")
        check(
            lib.h2mi_fr_scale_powers_dev(d_ext.ptr, self.extended_len(), self._zeta_inv.ctypes.data, self._eninv.ctypes.data, stream),
            "distribute_powers_zeta",
        )
