"""Algebras with a diagonal metric (reference: src/algebra.rs)."""
from __future__ import annotations

import ctypes as C

from . import _lib


class MetricAlgebra:
    """`[f64; D]` of the reference (algebra.rs:148-165): the squares of the base vectors."""

    def __init__(self, diag):
        self.diag = [float(x) for x in diag]

    def vec_space_dim(self):                       # algebra.rs:16
        return len(self.diag)

    def grade_dim(self, k):                        # algebra.rs:25-27
        return n_choose_k(len(self.diag), k)

    def base_vec_dot(self, v1, v2):                # algebra.rs:156-165
        return self.diag[v1] if v1 == v2 else 0.0

    def _c_diag(self):
        return (C.c_double * max(1, len(self.diag)))(*self.diag)

    def ortho_basis_blades_gp(self, b1, b2):       # algebra.rs:73-83
        res = C.c_uint64()
        coeff = _lib.lib().gaast_blades_gp(len(self.diag), self._c_diag(), b1, b2, C.byref(res))
        return res.value, coeff

    def component_to_basis_blade(self, grade, index):   # algebra.rs:31-37
        return _lib.lib().gaast_component_to_blade(len(self.diag), grade, index)

    def basis_blade_to_component(self, blade):           # algebra.rs:41-45
        g = C.c_int()
        idx = _lib.lib().gaast_blade_to_component(len(self.diag), blade, C.byref(g))
        return g.value, idx


class OrthoEuclidN(MetricAlgebra):
    """OrthoEuclidN(N) (algebra.rs:173-192): N orthogonal base vectors squaring to 1."""

    def __init__(self, n):
        super().__init__([1.0] * int(n))


def as_algebra(alg):
    if isinstance(alg, MetricAlgebra):
        return alg
    if isinstance(alg, int):
        return OrthoEuclidN(alg)
    return MetricAlgebra(alg)


def n_choose_k(n, k):                              # algebra.rs:252-254
    return _lib.lib().gaast_n_choose_k(int(n), int(k))
