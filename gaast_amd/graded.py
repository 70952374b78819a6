"""Graded storage (reference: src/graded.rs).

GradeMapMV -- host value, one dense array per grade (graded.rs:173-202); equality is exact.
DeviceMV   -- the device-side GradedDataMut: `batch` graded rows in HBM behind a
              gaast_hip_mv_t handle (include/gaast_hip.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .algebra import n_choose_k
from .grade_set import GradeSet

_NP = {_lib.F64: np.float64, _lib.F32: np.float32}


def _mask_of(grades):
    m = 0
    for k in grades:
        m |= 1 << int(k)
    return m


def row_len(dim, mask):
    return sum(n_choose_k(dim, k) for k in range(64) if (mask >> k) & 1)


class GradeMapMV:
    """HashMap<Grade, Vec<f64>> of the reference; `dim` sizes the slices (C(dim, k) each)."""

    def __init__(self, grades=None, dim=None):
        self.slabs = {int(k): np.array(v, dtype=np.float64).reshape(-1) for k, v in (grades or {}).items()}
        self.dim = self._infer_dim() if dim is None else int(dim)
        for k, v in self.slabs.items():
            if v.size != n_choose_k(self.dim, k):
                raise ValueError(f"grade {k} holds {v.size} components, C({self.dim},{k}) expected")

    def _infer_dim(self):
        for d in range(0, _lib.C.sizeof(C.c_uint16) * 8 + 1):
            if all(n_choose_k(d, k) == v.size for k, v in self.slabs.items()):
                return d
        raise ValueError("slice lengths are not those of any vector-space dimension")

    @staticmethod
    def init_null_mv(dim, gs):                      # graded.rs:195-201
        return GradeMapMV({k: np.zeros(n_choose_k(dim, k)) for k in gs.iter()}, dim=dim)

    def grade_set(self):                            # graded.rs:176-184
        return GradeSet.of(self.slabs.keys())

    def grade_slice(self, k):                       # graded.rs:186-190
        return self.slabs[k]

    def row(self):
        ks = sorted(self.slabs)
        return np.concatenate([self.slabs[k] for k in ks]) if ks else np.zeros(0)

    def to_dict(self):
        return {k: v.copy() for k, v in self.slabs.items()}

    def __eq__(self, other):                        # #[derive(PartialEq)], graded.rs:173
        if not isinstance(other, GradeMapMV):
            return NotImplemented
        if set(self.slabs) != set(other.slabs):
            return False
        return all(np.array_equal(self.slabs[k], other.slabs[k]) for k in self.slabs)

    def __repr__(self):
        return "GradeMapMV(" + ", ".join(f"{k}: {self.slabs[k].tolist()}" for k in sorted(self.slabs)) + ")"


def grade_map_mv(grades, dim=None):
    """grade_map_mv!(k => x y z, ...) (graded.rs:209-223)."""
    return GradeMapMV(grades, dim=dim)


class DeviceMV:
    """`batch` multivectors on the GPU, stored as graded rows (see include/gaast_hip.h)."""

    def __init__(self, handle, dim, mask, batch, dtype, keep=None):
        self._h = handle
        self.dim, self.mask, self.batch, self.dtype = dim, mask, batch, dtype
        self.row_len = row_len(dim, mask)
        self._keep = keep

    # -- construction ---------------------------------------------------------------------
    @staticmethod
    def alloc(dim, grades, batch, dtype=_lib.F64):
        """init_null_mv(dim, gs) for `batch` items (zero-filled)."""
        _lib.init_device()
        mask = grades.mask if isinstance(grades, GradeSet) else _mask_of(grades)
        h = C.c_void_p()
        _lib.check(_lib.lib().gaast_hip_mv_alloc(dim, mask, batch, dtype, C.byref(h)))
        return DeviceMV(h, dim, mask, batch, dtype)

    @staticmethod
    def from_rows(dim, grades, rows, dtype=_lib.F64):
        rows = np.ascontiguousarray(rows, dtype=_NP[dtype])
        if rows.ndim == 1:
            rows = rows[None, :]
        m = DeviceMV.alloc(dim, grades, rows.shape[0], dtype)
        m.upload_rows(rows)
        return m

    @staticmethod
    def from_value(value, dtype=_lib.F64):
        """One GradeMapMV -> batch-1 device multivector (shared by every item when bound)."""
        return DeviceMV.from_rows(value.dim, value.slabs.keys(), value.row()[None, :], dtype)

    @staticmethod
    def wrap_tensor(tensor, dim, grades):
        """View a 2-D contiguous torch CUDA tensor [batch, row_len] as a DeviceMV (no copy)."""
        import torch
        _lib.init_device()
        mask = grades.mask if isinstance(grades, GradeSet) else _mask_of(grades)
        dtype = {torch.float64: _lib.F64, torch.float32: _lib.F32}[tensor.dtype]
        assert tensor.dim() == 2 and tensor.stride(1) == 1 and tensor.shape[1] == row_len(dim, mask)
        h = C.c_void_p()
        _lib.check(_lib.lib().gaast_hip_mv_wrap(C.c_void_p(tensor.data_ptr()), dim, mask, tensor.shape[0],
                                                dtype, tensor.stride(0), C.byref(h)))
        return DeviceMV(h, dim, mask, tensor.shape[0], dtype, keep=tensor)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().gaast_hip_mv_free(self._h)
        except Exception:
            pass

    # -- GradedData / GradedDataMut ---------------------------------------------------------
    def grade_set(self):
        return GradeSet(self.mask)

    def upload(self, k, values):
        """grade_slice_mut(k) of every item <- values[batch, C(dim,k)]"""
        a = np.ascontiguousarray(values, dtype=_NP[self.dtype])
        _lib.check(_lib.lib().gaast_hip_mv_upload(self._h, int(k), a.ctypes.data_as(C.c_void_p), a.size))

    def download(self, k):
        """grade_slice(k) of every item -> [batch, C(dim,k)]"""
        a = np.empty((self.batch, n_choose_k(self.dim, k)), dtype=_NP[self.dtype])
        _lib.check(_lib.lib().gaast_hip_mv_download(self._h, int(k), a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def upload_rows(self, rows):
        a = np.ascontiguousarray(rows, dtype=_NP[self.dtype])
        _lib.check(_lib.lib().gaast_hip_mv_upload_rows(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def download_rows(self):
        a = np.empty((self.batch, self.row_len), dtype=_NP[self.dtype])
        _lib.check(_lib.lib().gaast_hip_mv_download_rows(self._h, a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def zero(self):
        _lib.check(_lib.lib().gaast_hip_mv_zero(self._h))

    def item(self, i=0):
        """Item i as a GradeMapMV (f64)."""
        row = self.download_rows()[i].astype(np.float64)
        out, pos = {}, 0
        for k in range(64):
            if (self.mask >> k) & 1:
                g = n_choose_k(self.dim, k)
                out[k] = row[pos:pos + g]
                pos += g
        return GradeMapMV(out, dim=self.dim)
