"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It wraps oracle/_build/libgaast_oracle.so (built by `make -C oracle`) with the
same operator surface as the reference's `Expr` (src/ast/expr.rs) so that a test can be
written once and run against both the oracle and the HIP product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgaast_oracle.so")

OG_OK = 0
PANICS = {1: "MISSING_GRADE", 2: "TODO", 3: "ASSERT", 4: "OVERFLOW", 5: "BAD_ARG"}
SEL_GEOMETRIC, SEL_OUTER, SEL_INNER, SEL_LCONTRACT, SEL_RCONTRACT = range(5)
EVAL_RELEASE, EVAL_DEBUG = 0, 1
EVAL_F32 = 4          # bit flag: every operand and result rounded to binary32 (checks the f32 exact kernels bit for bit)
EVAL_EXT_EXPLOG = 2   # bit flag: the exp / log EXTENSION (no reference behaviour, parity unpinned), see gaast_oracle.c
PANIC_DOMAIN = 5
NODE_KINDS = ["GradedObj", "Addition", "Product", "Negation", "Exponential", "Logarithm",
              "GradeProjection", "Reverse", "GradeInvolution", "ScalarInversion", "ScalarSqrt"]


class OraclePanic(RuntimeError):
    """The reference would have panicked here."""

    def __init__(self, code, msg):
        super().__init__(f"reference panic {PANICS.get(code, code)}: {msg}")
        self.code = code


class GradeSetC(C.Structure):
    _fields_ = [("bits", C.c_uint64), ("len", C.c_int)]


class AlgebraC(C.Structure):
    _fields_ = [("dim", C.c_int), ("is_euclid", C.c_int), ("diag", C.c_double * 64)]


class CompMulC(C.Structure):
    _fields_ = [("left_grade", C.c_size_t), ("left_index", C.c_size_t),
                ("right_grade", C.c_size_t), ("right_index", C.c_size_t),
                ("result_grade", C.c_size_t), ("result_index", C.c_size_t),
                ("coeff", C.c_double)]


class NodeInfoC(C.Structure):
    _fields_ = [("kind", C.c_int), ("child0", C.c_int), ("child1", C.c_int),
                ("maximal", C.c_uint64), ("minimal", C.c_uint64), ("vec_space_dim", C.c_int),
                ("num_uses", C.c_int), ("n_comp_muls", C.c_size_t), ("input", C.c_void_p)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "gaast_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, i64, u64, dbl, ci = C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int
    sigs = {
        "og_gs_empty": (GradeSetC, []),
        "og_gs_single": (GradeSetC, [i64]),
        "og_gs_range": (GradeSetC, [ci, ci]),
        "og_gs_intersection": (GradeSetC, [GradeSetC, GradeSetC]),
        "og_gs_add": (GradeSetC, [GradeSetC, GradeSetC]),
        "og_gs_mul": (GradeSetC, [GradeSetC, GradeSetC]),
        "og_gs_add_grade": (GradeSetC, [GradeSetC, ci]),
        "og_gs_rm_grade": (GradeSetC, [GradeSetC, ci]),
        "og_gs_eq": (ci, [GradeSetC, GradeSetC]),
        "og_gs_is_empty": (ci, [GradeSetC]),
        "og_gs_is_single": (ci, [GradeSetC]),
        "og_gs_contains": (ci, [GradeSetC, ci]),
        "og_gs_includes": (ci, [GradeSetC, GradeSetC]),
        "og_gs_is_just": (ci, [GradeSetC, ci]),
        "og_gs_iter": (ci, [GradeSetC, C.POINTER(ci), ci]),
        "og_gs_parts_contributing_to_product": (None, [GradeSetC, ci, GradeSetC, GradeSetC,
                                                       C.POINTER(GradeSetC), C.POINTER(GradeSetC)]),
        "og_n_choose_k": (u64, [u64, u64]),
        "og_index_to_bitfield_permut": (u64, [ci, ci, u64]),
        "og_bitfield_permut_to_index": (u64, [ci, ci, u64]),
        "og_canonical_reordering_sign": (dbl, [u64, u64]),
        "og_ortho_basis_blades_gp": (dbl, [C.POINTER(AlgebraC), u64, u64, C.POINTER(u64)]),
        "og_mv_new": (vp, []),
        "og_mv_free": (None, [vp]),
        "og_mv_set_grade": (ci, [vp, ci, C.POINTER(dbl), C.c_size_t]),
        "og_mv_grade_mask": (u64, [vp]),
        "og_mv_grade_len": (C.c_size_t, [vp, ci]),
        "og_mv_grade_ptr": (C.POINTER(dbl), [vp, ci]),
        "og_expr_retain": (vp, [vp]),
        "og_expr_release": (None, [vp]),
        "og_expr_mv": (vp, [vp]),
        "og_expr_from_f64": (vp, [dbl]),
        "og_expr_basis_vector": (vp, [ci, ci]),
        "og_expr_product": (vp, [vp, vp, ci]),
        "og_expr_add": (vp, [vp, vp]),
        "og_expr_neg": (vp, [vp]),
        "og_expr_sub": (vp, [vp, vp]),
        "og_expr_div_scalar": (vp, [vp, dbl]),
        "og_expr_rev": (vp, [vp]),
        "og_expr_ginvol": (vp, [vp]),
        "og_expr_exp": (vp, [vp]),
        "og_expr_log": (vp, [vp]),
        "og_expr_pow": (vp, [vp, vp]),
        "og_expr_sqrt": (vp, [vp]),
        "og_expr_g": (vp, [vp, i64]),
        "og_expr_gselect_mask": (vp, [vp, u64]),
        "og_expr_conj": (vp, [vp]),
        "og_expr_scal": (vp, [vp, vp]),
        "og_expr_norm_sq": (vp, [vp]),
        "og_expr_sinv": (vp, [vp]),
        "og_expr_vinv": (vp, [vp]),
        "og_specialize": (vp, [vp, C.POINTER(AlgebraC), C.POINTER(ci)]),
        "og_spec_free": (None, [vp]),
        "og_last_panic": (C.c_char_p, []),
        "og_spec_num_nodes": (ci, [vp]),
        "og_spec_root": (ci, [vp]),
        "og_spec_node": (ci, [vp, ci, C.POINTER(NodeInfoC)]),
        "og_spec_comp_muls": (C.POINTER(CompMulC), [vp, ci]),
        "og_eval": (ci, [vp, ci, C.POINTER(vp)]),
        "og_eval_batch": (ci, [vp, ci, C.POINTER(vp), C.POINTER(C.POINTER(dbl)), ci, i64,
                               C.POINTER(dbl), C.c_size_t]),
        "og_product_loop": (None, [C.POINTER(CompMulC), C.c_size_t, C.POINTER(C.POINTER(dbl)),
                                   C.POINTER(C.POINTER(dbl)), C.POINTER(C.POINTER(dbl))]),
        # cpu-packed baselines: bench infrastructure
        "og_pack_root_product": (C.c_size_t, [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "og_packed_free": (None, [vp]),
        "og_packed_eval_batch": (C.c_double, [vp, C.c_size_t, C.POINTER(dbl), C.c_size_t, C.POINTER(dbl), C.c_size_t,
                                              C.POINTER(dbl), C.c_size_t, i64, ci]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# --------------------------------------------------------------------------------------
# GradeSet (src/grade_set.rs)
# --------------------------------------------------------------------------------------
class GradeSet:
    def __init__(self, c):
        self.c = c

    @staticmethod
    def empty():
        return GradeSet(lib().og_gs_empty())

    @staticmethod
    def single(k):
        return GradeSet(lib().og_gs_single(k))

    @staticmethod
    def range(x, y):
        return GradeSet(lib().og_gs_range(x, y))

    @staticmethod
    def from_mask(mask):
        return GradeSet(GradeSetC(mask, 64))

    def intersection(self, rhs):
        return GradeSet(lib().og_gs_intersection(self.c, rhs.c))

    def __add__(self, rhs):
        return GradeSet(lib().og_gs_add(self.c, rhs.c))

    def __mul__(self, rhs):
        return GradeSet(lib().og_gs_mul(self.c, rhs.c))

    def __eq__(self, rhs):
        return bool(lib().og_gs_eq(self.c, rhs.c))

    def __ne__(self, rhs):
        return not self == rhs

    def is_empty(self):
        return bool(lib().og_gs_is_empty(self.c))

    def is_single(self):
        return bool(lib().og_gs_is_single(self.c))

    def contains(self, k):
        return bool(lib().og_gs_contains(self.c, k))

    def includes(self, other):
        return bool(lib().og_gs_includes(self.c, other.c))

    def add_grade(self, k):
        return GradeSet(lib().og_gs_add_grade(self.c, k))

    def rm_grade(self, k):
        return GradeSet(lib().og_gs_rm_grade(self.c, k))

    def iter(self):
        buf = (C.c_int * 64)()
        n = lib().og_gs_iter(self.c, buf, 64)
        return [buf[i] for i in range(n)]

    def parts_contributing_to_product(self, sel_kind, left, right):
        ol, orr = GradeSetC(), GradeSetC()
        lib().og_gs_parts_contributing_to_product(self.c, sel_kind, left.c, right.c,
                                                  C.byref(ol), C.byref(orr))
        return GradeSet(ol), GradeSet(orr)

    @property
    def mask(self):
        return self.c.bits & ((1 << self.c.len) - 1 if self.c.len < 64 else (1 << 64) - 1)

    def __repr__(self):
        return f"GradeSet{self.iter()}"


# --------------------------------------------------------------------------------------
# Algebra (src/algebra.rs)
# --------------------------------------------------------------------------------------
def ortho_euclid_n(n):
    a = AlgebraC()
    a.dim, a.is_euclid = n, 1
    return a


def diag_metric(diag):
    a = AlgebraC()
    a.dim, a.is_euclid = len(diag), 0
    for i, d in enumerate(diag):
        a.diag[i] = float(d)
    return a


def as_algebra(alg):
    """int n -> OrthoEuclidN(n); sequence -> [f64; D]."""
    if isinstance(alg, AlgebraC):
        return alg
    if isinstance(alg, int):
        return ortho_euclid_n(alg)
    return diag_metric(list(alg))


# --------------------------------------------------------------------------------------
# GradeMapMV (src/graded.rs:173-202)
# --------------------------------------------------------------------------------------
class GradeMapMV:
    """Owns an og_mv; equality is exact f64 equality of the whole grade map (graded.rs:173)."""

    def __init__(self, grades=None, _ptr=None):
        self._p = _ptr if _ptr is not None else lib().og_mv_new()
        if grades:
            for k, vals in grades.items():
                self.set_grade(k, vals)

    def set_grade(self, k, vals):
        arr = np.ascontiguousarray(vals, dtype=np.float64)
        lib().og_mv_set_grade(self._p, int(k), arr.ctypes.data_as(C.POINTER(C.c_double)), arr.size)

    def grades(self):
        m = lib().og_mv_grade_mask(self._p)
        return [k for k in range(64) if (m >> k) & 1]

    def grade_slice(self, k):
        n = lib().og_mv_grade_len(self._p, k)
        p = lib().og_mv_grade_ptr(self._p, k)
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0)

    def to_dict(self):
        return {k: self.grade_slice(k) for k in self.grades()}

    def __eq__(self, other):
        a, b = self.to_dict(), (other.to_dict() if isinstance(other, GradeMapMV) else other)
        if set(a) != set(b):
            return False
        return all(np.array_equal(a[k], np.asarray(b[k], dtype=np.float64)) for k in a)

    def __repr__(self):
        return "GradeMapMV(" + ", ".join(f"{k}: {v.tolist()}" for k, v in self.to_dict().items()) + ")"

    def __del__(self):
        try:
            lib().og_mv_free(self._p)
        except Exception:
            pass


def grade_map_mv(grades):
    """grade_map_mv!(k => x y z, ...)  (graded.rs:209-223)"""
    return GradeMapMV(grades)


# --------------------------------------------------------------------------------------
# Expr (src/ast/expr.rs)
# --------------------------------------------------------------------------------------
class Expr:
    def __init__(self, ptr, keep=()):
        self._p = ptr
        self._keep = tuple(keep)  # python objects (input MVs, children) that must outlive us

    def __del__(self):
        try:
            lib().og_expr_release(self._p)
        except Exception:
            pass

    @staticmethod
    def _lift(x):
        if isinstance(x, Expr):
            return x
        if isinstance(x, (int, float)):
            return Expr(lib().og_expr_from_f64(float(x)))
        raise TypeError(type(x))

    @staticmethod
    def basis_vectors(dim):
        return [Expr(lib().og_expr_basis_vector(dim, i)) for i in range(dim)]

    def _bin(self, rhs, fn, *extra):
        rhs = Expr._lift(rhs)
        return Expr(fn(self._p, rhs._p, *extra), keep=(self, rhs))

    def _un(self, fn, *extra):
        return Expr(fn(self._p, *extra), keep=(self,))

    # products, expr.rs:180-197
    def __mul__(self, rhs):
        return self._bin(rhs, lib().og_expr_product, SEL_GEOMETRIC)

    def __rmul__(self, lhs):
        return Expr._lift(lhs)._bin(self, lib().og_expr_product, SEL_GEOMETRIC)

    def __xor__(self, rhs):
        return self._bin(rhs, lib().og_expr_product, SEL_OUTER)

    def __and__(self, rhs):
        return self._bin(rhs, lib().og_expr_product, SEL_INNER)

    def __lshift__(self, rhs):
        return self._bin(rhs, lib().og_expr_product, SEL_LCONTRACT)

    def __rshift__(self, rhs):
        return self._bin(rhs, lib().og_expr_product, SEL_RCONTRACT)

    def __add__(self, rhs):
        return self._bin(rhs, lib().og_expr_add)

    def __radd__(self, lhs):
        return Expr._lift(lhs)._bin(self, lib().og_expr_add)

    def __neg__(self):
        return self._un(lib().og_expr_neg)

    def __sub__(self, rhs):
        return self._bin(rhs, lib().og_expr_sub)

    def __truediv__(self, s):
        return self._un(lib().og_expr_div_scalar, float(s))

    def rev(self):
        return self._un(lib().og_expr_rev)

    def ginvol(self):
        return self._un(lib().og_expr_ginvol)

    def exp(self):
        return self._un(lib().og_expr_exp)

    def log(self):
        return self._un(lib().og_expr_log)

    def pow(self, p):
        return self._bin(p, lib().og_expr_pow)

    def sqrt(self):
        return self._un(lib().og_expr_sqrt)

    def g(self, k):
        return self._un(lib().og_expr_g, int(k))

    def gselect(self, grades):
        mask = 0
        for k in grades:
            mask |= 1 << k
        return self._un(lib().og_expr_gselect_mask, mask)

    def conj(self):
        return self._un(lib().og_expr_conj)

    def scal(self, rhs):
        return self._bin(rhs, lib().og_expr_scal)

    def norm_sq(self):
        return self._un(lib().og_expr_norm_sq)

    def sinv(self):
        return self._un(lib().og_expr_sinv)

    def vinv(self):
        return self._un(lib().og_expr_vinv)

    def specialize(self, alg):
        return SpecializedAst(self, alg)


def mv(x):
    """mv(x) (expr.rs:162-164): the expression borrows x."""
    return Expr(lib().og_expr_mv(x._p), keep=(x,))


class SpecializedAst:
    def __init__(self, expr, alg):
        self._alg = as_algebra(alg)
        st = C.c_int(0)
        self._p = lib().og_specialize(expr._p, C.byref(self._alg), C.byref(st))
        self._expr = expr
        if not self._p:
            raise OraclePanic(st.value, lib().og_last_panic().decode())

    def __del__(self):
        try:
            if self._p:
                lib().og_spec_free(self._p)
        except Exception:
            pass

    def eval(self, mode=EVAL_RELEASE):
        out = C.c_void_p()
        st = lib().og_eval(self._p, mode, C.byref(out))
        if st != OG_OK:
            raise OraclePanic(st, lib().og_last_panic().decode())
        return GradeMapMV(_ptr=out.value)

    def packed_root_product(self):
        """(handle, entries, left_len, right_len, out_len) of the root product's list as 16-byte packed entries on flat rows
        (cpu-packed variants), or None when the root is not a product of two leaves"""
        p = C.c_void_p()
        ll, rl, ol = C.c_size_t(), C.c_size_t(), C.c_size_t()
        n = lib().og_pack_root_product(self._p, C.byref(p), C.byref(ll), C.byref(rl), C.byref(ol))
        if not n:
            return None
        return p, int(n), int(ll.value), int(rl.value), int(ol.value)

    def eval_batch(self, inputs, in_data, batch, out_row_len, mode=EVAL_RELEASE):
        """inputs: list of GradeMapMV bound in the expression; in_data: list of (batch, row) f64
        arrays or None (shared); returns (batch, out_row_len) array."""
        n = len(inputs)
        ptrs = (C.c_void_p * n)(*[m._p for m in inputs])
        arrs = [None if d is None else np.ascontiguousarray(d, dtype=np.float64) for d in in_data]
        dptr = (C.POINTER(C.c_double) * n)()
        for j, a in enumerate(arrs):
            dptr[j] = a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
        out = np.zeros((batch, out_row_len), dtype=np.float64)
        st = lib().og_eval_batch(self._p, mode, ptrs, dptr, n, batch,
                                 out.ctypes.data_as(C.POINTER(C.c_double)), out_row_len)
        if st != OG_OK:
            raise OraclePanic(st, lib().og_last_panic().decode())
        return out

    # introspection -------------------------------------------------------------
    def nodes(self):
        out = []
        for i in range(lib().og_spec_num_nodes(self._p)):
            info = NodeInfoC()
            lib().og_spec_node(self._p, i, C.byref(info))
            out.append(info)
        return out

    def root(self):
        return lib().og_spec_root(self._p)

    def comp_muls(self, idx):
        info = NodeInfoC()
        lib().og_spec_node(self._p, idx, C.byref(info))
        p = lib().og_spec_comp_muls(self._p, idx)
        return [(p[i].left_grade, p[i].left_index, p[i].right_grade, p[i].right_index,
                 p[i].result_grade, p[i].result_index, p[i].coeff) for i in range(info.n_comp_muls)]

    def comp_muls_raw(self, idx):
        info = NodeInfoC()
        lib().og_spec_node(self._p, idx, C.byref(info))
        return lib().og_spec_comp_muls(self._p, idx), info.n_comp_muls
