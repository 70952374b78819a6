"""Expr / SpecializedAst (reference: src/ast/expr.rs, src/ast/specialize.rs, src/eval.rs).

Same operator surface as the reference: `*` geometric, `^` outer, `&` inner, `<<` / `>>`
contractions, `+ - unary-`, `.g(k) .rev() .ginvol() .conj() .scal() .norm_sq() .sinv() .vinv()
.sqrt() .pow() .exp() .log()`.  Phases 1-3 run in the C++ host code of libgaast_hip.so;
`SpecializedAst.eval*` is phase 4 on the GPU through the C ABI -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .algebra import as_algebra
from .grade_set import GradeSet
from .graded import DeviceMV, GradeMapMV, _mask_of

DEFAULT_MATERIALIZE_LIMIT = 1 << 22   # larger comp-mul lists stay compact descriptors


class Input:
    """Placeholder for a multivector bound at evaluation time (one per batch item)."""

    def __init__(self, slot, grades, dim):
        self.slot, self.dim = int(slot), int(dim)
        self.mask = grades.mask if isinstance(grades, GradeSet) else _mask_of(grades)


class Expr:
    def __init__(self, ptr, keep=()):
        if not ptr:
            raise _lib.GaastError(_lib.STATUS_NAMES and 6, _lib.lib().gaast_expr_last_error().decode())
        self._p = C.c_void_p(ptr)
        self._keep = tuple(keep)

    def __del__(self):
        try:
            _lib.lib().gaast_expr_release(self._p)
        except Exception:
            pass

    # -- leaves ---------------------------------------------------------------------------
    @staticmethod
    def _lift(x):
        if isinstance(x, Expr):
            return x
        if isinstance(x, (int, float)):
            return Expr(_lib.lib().gaast_expr_from_f64(float(x)))   # From<f64>/From<i64>, expr.rs:231-246
        return mv(x)

    @staticmethod
    def basis_vectors(dim):                                          # expr.rs:148-157
        return [Expr(_lib.lib().gaast_expr_basis_vector(dim, i)) for i in range(dim)]

    def _bin(self, rhs, fn, *extra):
        rhs = Expr._lift(rhs)
        return Expr(fn(self._p, rhs._p, *extra), keep=(self, rhs))

    def _un(self, fn, *extra):
        return Expr(fn(self._p, *extra), keep=(self,))

    # -- products, expr.rs:166-197 ---------------------------------------------------------
    def __mul__(self, rhs):
        return self._bin(rhs, _lib.lib().gaast_expr_product, _lib.PROD_GEOMETRIC)

    def __rmul__(self, lhs):                                         # expr.rs:256-262
        return Expr._lift(lhs)._bin(self, _lib.lib().gaast_expr_product, _lib.PROD_GEOMETRIC)

    def __xor__(self, rhs):
        return self._bin(rhs, _lib.lib().gaast_expr_product, _lib.PROD_OUTER)

    def __and__(self, rhs):
        return self._bin(rhs, _lib.lib().gaast_expr_product, _lib.PROD_INNER)

    def __lshift__(self, rhs):
        return self._bin(rhs, _lib.lib().gaast_expr_product, _lib.PROD_LCONTRACT)

    def __rshift__(self, rhs):
        return self._bin(rhs, _lib.lib().gaast_expr_product, _lib.PROD_RCONTRACT)

    def product(self, rhs, grades_to_produce):                       # expr.rs:123-144
        """Custom product: grades_to_produce(k1, k2) -> iterable of grades / GradeSet."""
        def cb(k1, k2, _user):
            g = grades_to_produce(k1, k2)
            return g.mask if isinstance(g, GradeSet) else _mask_of(g)
        cfn = _lib.SELECT_FN(cb)
        rhs = Expr._lift(rhs)
        return Expr(_lib.lib().gaast_expr_product_custom(self._p, rhs._p, cfn, None), keep=(self, rhs, cfn))

    # -- sums, expr.rs:200-229 -------------------------------------------------------------
    def __add__(self, rhs):
        return self._bin(rhs, _lib.lib().gaast_expr_add)

    def __radd__(self, lhs):                                         # expr.rs:249-255
        return Expr._lift(lhs)._bin(self, _lib.lib().gaast_expr_add)

    def __neg__(self):
        return self._un(_lib.lib().gaast_expr_neg)

    def __sub__(self, rhs):
        return self._bin(rhs, _lib.lib().gaast_expr_sub)

    def __truediv__(self, s):                                        # expr.rs:263-270
        return self._un(_lib.lib().gaast_expr_div_scalar, float(s))

    # -- unary ops & shortcuts, expr.rs:276-372 ----------------------------------------------
    def rev(self):
        return self._un(_lib.lib().gaast_expr_rev)

    def ginvol(self):
        return self._un(_lib.lib().gaast_expr_ginvol)

    def exp(self):
        return self._un(_lib.lib().gaast_expr_exp)

    def log(self):
        return self._un(_lib.lib().gaast_expr_log)

    def pow(self, p):
        return self._bin(p, _lib.lib().gaast_expr_pow)

    def sqrt(self):
        return self._un(_lib.lib().gaast_expr_sqrt)

    def g(self, k):
        return self._un(_lib.lib().gaast_expr_g, int(k))

    def gselect(self, grades):
        mask = grades.mask if isinstance(grades, GradeSet) else _mask_of(grades)
        return self._un(_lib.lib().gaast_expr_gselect_mask, mask)

    def conj(self):
        return self._un(_lib.lib().gaast_expr_conj)

    def scal(self, rhs):
        return self._bin(rhs, _lib.lib().gaast_expr_scal)

    def norm_sq(self):
        return self._un(_lib.lib().gaast_expr_norm_sq)

    def sinv(self):
        return self._un(_lib.lib().gaast_expr_sinv)

    def vinv(self):
        return self._un(_lib.lib().gaast_expr_vinv)

    # -- phases 2-3 ----------------------------------------------------------------------------
    def specialize(self, alg, dtype=_lib.F64, flags=0, materialize_limit=DEFAULT_MATERIALIZE_LIMIT):
        """Expr::specialize(&alg) (specialize.rs:36-50)."""
        return SpecializedAst(self, as_algebra(alg), dtype, flags, materialize_limit)


def mv(x):
    """mv(x) (expr.rs:162-164).  x: a GradeMapMV (value fixed now, shared by all batch items),
    a float (scalar MV) or an Input placeholder (bound per item at evaluation time)."""
    L = _lib.lib()
    if isinstance(x, Input):
        return Expr(L.gaast_expr_input(x.slot, x.mask, x.dim))
    if isinstance(x, (int, float)):
        x = GradeMapMV({0: [float(x)]}, dim=0)     # f64 as a scalar MV, graded.rs:145-166
    if isinstance(x, GradeMapMV):
        row = np.ascontiguousarray(x.row(), dtype=np.float64)
        return Expr(L.gaast_expr_const(x.grade_set().mask, x.dim, row.ctypes.data_as(C.POINTER(C.c_double)), row.size))
    raise TypeError(type(x))


class SpecializedAst:
    """SpecializedAst (specialize.rs:10-24) + its evaluation on the GPU (eval.rs:12-19)."""

    def __init__(self, expr, alg, dtype, flags, materialize_limit):
        L = _lib.lib()
        self.alg, self.dtype, self.flags = alg, dtype, flags
        self._expr = expr
        self._p = C.c_void_p(L.gaast_expr_specialize(expr._p, alg.vec_space_dim(), alg._c_diag(), materialize_limit))
        if not self._p:
            raise _lib.GaastError(1, L.gaast_expr_last_error().decode())
        self._prog = None
        self._n_slots = None
        self._out_info = None
        self._input_descs = None

    def __del__(self):
        try:
            if self._prog:
                _lib.lib().gaast_hip_program_destroy(self._prog)
            if self._p:
                _lib.lib().gaast_spec_free(self._p)
        except Exception:
            pass

    # -- public read API of the reference (specialize.rs:17-24, base_types.rs:124-146) -----------
    def root_id(self):
        return _lib.lib().gaast_spec_root(self._p)

    def num_nodes(self):
        return _lib.lib().gaast_spec_num_nodes(self._p)

    def get_node(self, idx):
        info = _lib.SpecNodeInfo()
        st = _lib.lib().gaast_spec_node(self._p, idx, C.byref(info))
        if st:
            raise IndexError(idx)
        return info

    def nodes(self):
        return [self.get_node(i) for i in range(self.num_nodes())]

    def comp_muls(self, idx):
        info = self.get_node(idx)
        p = _lib.lib().gaast_spec_comp_muls(self._p, idx)
        if not p:
            return None
        return [(p[i].left_grade, p[i].left_index, p[i].right_grade, p[i].right_index,
                 p[i].result_grade, p[i].result_index, p[i].coeff) for i in range(info.n_comp_muls)]

    def num_user_inputs(self):
        return _lib.lib().gaast_spec_num_user_inputs(self._p)

    def program_desc(self):
        d = _lib.ProgramDesc()
        _lib.check(_lib.lib().gaast_spec_program_desc(self._p, self.dtype, self.flags, C.byref(d)))
        return d

    def serialize(self):
        """The flat program as bytes (program wire format, include/gaast_expr.h)."""
        d = self.program_desc()
        n = _lib.lib().gaast_program_serialize(C.byref(d), None, 0)
        buf = (C.c_ubyte * n)()
        assert _lib.lib().gaast_program_serialize(C.byref(d), buf, n) == n
        return bytes(buf)

    # -- phase 4 on the device ------------------------------------------------------------------
    def program(self):
        """gaast_hip_program_create for this AST (built once, reused by every eval)."""
        if self._prog is None:
            _lib.init_device()
            d = self.program_desc()
            h = C.c_void_p()
            _lib.check(_lib.lib().gaast_hip_program_create(C.byref(d), C.byref(h)))
            self._prog = h
        return self._prog

    def output_info(self):
        if self._out_info is None:
            mask, rl = C.c_uint64(), C.c_int64()
            _lib.check(_lib.lib().gaast_hip_program_output_info(self.program(), C.byref(mask), C.byref(rl)))
            self._out_info = (mask.value, rl.value)
        return self._out_info

    def domain_errors(self):
        """GAAST_FLAG_EXP_LOG extension: items refused by the exp / log domain check since the last call (synchronises)."""
        n = C.c_int64()
        _lib.check(_lib.lib().gaast_hip_program_domain_errors(self.program(), C.byref(n)))
        return n.value

    def jit_source(self):
        """GAAST_FLAG_DEBUG_KEEP_JIT_SOURCE: the HIP source hiprtc compiled for this program."""
        return _lib.lib().gaast_hip_program_jit_source(self.program()).decode()

    def _bind(self, inputs):
        """inputs -> (handle array, n_slots, objects to keep alive); host arrays are uploaded."""
        L = _lib.lib()
        if self._n_slots is None:
            self._n_slots = L.gaast_spec_num_inputs(self._p)
        n_slots = self._n_slots
        handles = (C.c_void_p * max(1, n_slots))()
        keep = []
        for slot in range(min(len(inputs), n_slots)):
            x = inputs[slot]
            if x is None:
                continue
            if not isinstance(x, DeviceMV):
                if self._input_descs is None:    # only a host array needs the slot's layout
                    d = self.program_desc()
                    self._input_descs = [(d.inputs[i].storage_dim, d.inputs[i].grade_mask) for i in range(d.n_inputs)]
                sdim, smask = self._input_descs[slot]
                x = DeviceMV.from_rows(sdim, GradeSet(smask), x, self.dtype)
            keep.append(x)
            handles[slot] = x._h
        return handles, n_slots, keep

    def launches(self):
        p = self.program()
        return [_lib.lib().gaast_hip_program_launch_name(p, i).decode()
                for i in range(_lib.lib().gaast_hip_program_num_launches(p))]

    def eval_batch(self, inputs=(), batch=1, out=None):
        """One evaluation per batch item.  inputs[slot]: DeviceMV (batch items, or 1 = shared),
        a [batch, row_len] array, or None for unused slots.  Returns the output DeviceMV."""
        prog = self.program()
        handles, n_slots, keep = self._bind(inputs)
        if out is None:
            mask, _ = self.output_info()
            out = DeviceMV.alloc(self.get_node(self.root_id()).vec_space_dim, GradeSet(mask), batch, self.dtype)
        _lib.check(_lib.lib().gaast_hip_eval(prog, handles, n_slots, batch, out._h))
        out._keep_inputs = keep
        return out

    def eval_gather(self, inputs, out, gathered, counts, root=0, n_chunks=4):
        """gaast_hip_eval_gather: evaluate this rank's shard (counts[rank] items) in n_chunks chunks, chunk k travelling
        to `root` over RCCL while chunk k + 1 is computed.  `gathered` (root only) receives all ranks' rows in item order."""
        prog = self.program()
        handles, n_slots, keep = self._bind(inputs)
        cnt = (C.c_int64 * len(counts))(*counts)
        _lib.check(_lib.lib().gaast_hip_eval_gather(prog, handles, n_slots, out._h, gathered._h if gathered is not None else None,
                                                    cnt, root, n_chunks))
        out._keep_inputs = keep
        return out

    def eval(self):
        """eval::<GradeMapMV>() (eval.rs:12-19) for an AST whose inputs are all fixed values."""
        out = self.eval_batch((), 1)
        _lib.check(_lib.lib().gaast_hip_synchronize())
        return out.item(0)


class ProgramImage:
    """A program decoded from its wire format; evaluates like a SpecializedAst."""

    def __init__(self, data: bytes):
        self._buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
        self._img = C.c_void_p(_lib.lib().gaast_program_deserialize(self._buf, len(data)))
        if not self._img:
            raise _lib.GaastError(1, "malformed program image")
        self.desc = _lib.lib().gaast_program_image_desc(self._img).contents
        self._prog = None

    def __del__(self):
        try:
            if self._prog:
                _lib.lib().gaast_hip_program_destroy(self._prog)
            _lib.lib().gaast_program_image_free(self._img)
        except Exception:
            pass

    def eval_batch(self, inputs, batch):
        L = _lib.lib()
        _lib.init_device()
        if self._prog is None:
            h = C.c_void_p()
            _lib.check(L.gaast_hip_program_create(C.byref(self.desc), C.byref(h)))
            self._prog = h
        mask, rl = C.c_uint64(), C.c_int64()
        _lib.check(L.gaast_hip_program_output_info(self._prog, C.byref(mask), C.byref(rl)))
        handles = (C.c_void_p * max(1, self.desc.n_inputs))()
        keep = []
        for slot, x in enumerate(inputs):
            if x is None:
                continue
            if not isinstance(x, DeviceMV):
                ind = self.desc.inputs[slot]
                x = DeviceMV.from_rows(ind.storage_dim, GradeSet(ind.grade_mask), x, self.desc.dtype)
            keep.append(x)
            handles[slot] = x._h
        out = DeviceMV.alloc(self.desc.nodes[self.desc.root].vec_space_dim, GradeSet(mask.value), batch, self.desc.dtype)
        _lib.check(L.gaast_hip_eval(self._prog, handles, self.desc.n_inputs, batch, out._h))
        out._keep_inputs = keep
        return out
