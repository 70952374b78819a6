"""ctypes loader for libgaast_hip.so (the C ABI of include/gaast_hip.h and gaast_expr.h).

There is no fallback: if the shared library is missing or no gfx950 GPU is visible, the
functions that need them raise.  Build with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C gaast_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAAST_HIP_LIB: another build of the same library (kernel A/B measurements: make KFLAGS=... OUTDIR=...)
LIB_PATH = os.environ.get("GAAST_HIP_LIB") or os.path.join(_HERE, "lib", "libgaast_hip.so")

GAAST_OK = 0
STATUS_NAMES = {1: "INVALID_PROGRAM", 2: "MISSING_GRADE", 3: "UNIMPLEMENTED", 4: "HIP", 5: "RCCL",
                6: "INVALID_ARGUMENT", 7: "NO_DEVICE", 8: "OVERFLOW"}
F64, F32 = 0, 1
(OP_INPUT, OP_ADD, OP_PRODUCT, OP_NEG, OP_EXP, OP_LOG, OP_PROJ, OP_REVERSE, OP_GINVOL, OP_SINV,
 OP_SSQRT) = range(11)
OP_NAMES = ["GradedObj", "Addition", "Product", "Negation", "Exponential", "Logarithm",
            "GradeProjection", "Reverse", "GradeInvolution", "ScalarInversion", "ScalarSqrt"]
PROD_EXPLICIT, PROD_GEOMETRIC, PROD_OUTER, PROD_INNER, PROD_LCONTRACT, PROD_RCONTRACT = -1, 0, 1, 2, 3, 4
FLAG_DEBUG_OVERFLOW, FLAG_NO_FUSION, FLAG_EXACT_ORDER, FLAG_NO_MFMA, FLAG_NO_JIT, FLAG_SPINOR_GEMM = 1, 2, 4, 8, 16, 32
FLAG_DEBUG_JIT_FAILS, FLAG_DEBUG_KEEP_JIT_SOURCE, FLAG_EXP_LOG, FLAG_NO_COALESCE, FLAG_DEBUG_LDS_12K = 0x40, 0x80, 0x100, 0x200, 0x400
FLAG_DEBUG_NO_CHAIN = 0x800
COMM_ID_BYTES = 128


class GaastError(RuntimeError):
    """Non-zero gaast_status.  Where the reference would panic, `.status` says which panic."""

    def __init__(self, status, msg):
        super().__init__(f"gaast status {STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status
        self.status_name = STATUS_NAMES.get(status, str(status))


class CompMul(C.Structure):
    _fields_ = [("left_grade", C.c_uint32), ("left_index", C.c_uint32), ("right_grade", C.c_uint32),
                ("right_index", C.c_uint32), ("result_grade", C.c_uint32), ("result_index", C.c_uint32),
                ("coeff", C.c_double)]


class NodeDesc(C.Structure):
    _fields_ = [("opcode", C.c_int32), ("child0", C.c_int32), ("child1", C.c_int32),
                ("minimal_grade_mask", C.c_uint64), ("vec_space_dim", C.c_int32),
                ("input_slot", C.c_int32), ("product_kind", C.c_int32), ("n_comp_muls", C.c_uint64),
                ("comp_muls", C.POINTER(CompMul))]


class InputDesc(C.Structure):
    _fields_ = [("grade_mask", C.c_uint64), ("storage_dim", C.c_int32), ("is_const", C.c_int32),
                ("const_row", C.POINTER(C.c_double))]


class ProgramDesc(C.Structure):
    _fields_ = [("vec_space_dim", C.c_int32), ("metric_diag", C.POINTER(C.c_double)),
                ("dtype", C.c_int32), ("n_nodes", C.c_int32), ("nodes", C.POINTER(NodeDesc)),
                ("root", C.c_int32), ("n_inputs", C.c_int32), ("inputs", C.POINTER(InputDesc)),
                ("flags", C.c_uint32)]


class SpecNodeInfo(C.Structure):
    _fields_ = [("opcode", C.c_int32), ("child0", C.c_int32), ("child1", C.c_int32),
                ("maximal_grade_mask", C.c_uint64), ("minimal_grade_mask", C.c_uint64),
                ("vec_space_dim", C.c_int32), ("num_uses", C.c_int32), ("input_slot", C.c_int32),
                ("product_kind", C.c_int32), ("n_comp_muls", C.c_uint64)]


SELECT_FN = C.CFUNCTYPE(C.c_uint64, C.c_int64, C.c_int64, C.c_void_p)

# every symbol the two public headers declare: name -> (restype, argtypes)
_vp, _i64, _u64, _dbl, _ci, _sz = C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_int, C.c_size_t
_pd = C.POINTER(C.c_double)
SIGNATURES = {
    # ---- include/gaast_hip.h ----
    "gaast_hip_init": (_ci, [C.POINTER(_ci), _ci]),
    "gaast_hip_shutdown": (_ci, []),
    "gaast_hip_set_stream": (_ci, [_vp]),
    "gaast_hip_synchronize": (_ci, []),
    "gaast_hip_last_error": (C.c_char_p, []),
    "gaast_hip_version": (C.c_char_p, []),
    "gaast_hip_program_create": (_ci, [C.POINTER(ProgramDesc), C.POINTER(_vp)]),
    "gaast_hip_program_destroy": (_ci, [_vp]),
    "gaast_hip_program_output_info": (_ci, [_vp, C.POINTER(_u64), C.POINTER(_i64)]),
    "gaast_hip_program_num_launches": (_ci, [_vp]),
    "gaast_hip_program_launch_name": (C.c_char_p, [_vp, _ci]),
    "gaast_hip_mv_alloc": (_ci, [_ci, _u64, _i64, _ci, C.POINTER(_vp)]),
    "gaast_hip_mv_wrap": (_ci, [_vp, _ci, _u64, _i64, _ci, _i64, C.POINTER(_vp)]),
    "gaast_hip_mv_free": (_ci, [_vp]),
    "gaast_hip_mv_info": (_ci, [_vp, C.POINTER(_ci), C.POINTER(_u64), C.POINTER(_i64), C.POINTER(_ci),
                                C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_vp)]),
    "gaast_hip_mv_upload": (_ci, [_vp, _ci, _vp, _i64]),
    "gaast_hip_mv_download": (_ci, [_vp, _ci, _vp, _i64]),
    "gaast_hip_mv_upload_rows": (_ci, [_vp, _vp, _i64]),
    "gaast_hip_mv_download_rows": (_ci, [_vp, _vp, _i64]),
    "gaast_hip_mv_zero": (_ci, [_vp]),
    "gaast_hip_eval": (_ci, [_vp, C.POINTER(_vp), _ci, _i64, _vp]),
    "gaast_hip_program_jit_source": (C.c_char_p, [_vp]),
    "gaast_hip_program_domain_errors": (_ci, [_vp, C.POINTER(_i64)]),
    "gaast_hip_comm_set_library": (_ci, [C.c_char_p]),
    "gaast_hip_comm_unique_id": (_ci, [_vp]),
    "gaast_hip_comm_init": (_ci, [_vp, _ci, _ci]),
    "gaast_hip_comm_destroy": (_ci, []),
    "gaast_hip_comm_info": (_ci, [C.POINTER(_ci), C.POINTER(_ci)]),
    "gaast_hip_comm_count_ranks": (_ci, [C.POINTER(_ci)]),
    "gaast_hip_gather_rows": (_ci, [_vp, _vp, C.POINTER(_i64), _ci]),
    "gaast_hip_eval_gather": (_ci, [_vp, C.POINTER(_vp), _ci, _vp, _vp, C.POINTER(_i64), _ci, _ci]),
    # ---- include/gaast_expr.h ----
    "gaast_expr_last_error": (C.c_char_p, []),
    "gaast_gs_single": (_u64, [_i64]),
    "gaast_gs_range": (_u64, [_ci, _ci]),
    "gaast_gs_mul": (_u64, [_u64, _u64]),
    "gaast_gs_select": (_u64, [_ci, _i64, _i64]),
    "gaast_gs_parts_contributing_to_product": (None, [_u64, _ci, _u64, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "gaast_n_choose_k": (_u64, [_u64, _u64]),
    "gaast_component_to_blade": (_u64, [_ci, _ci, _u64]),
    "gaast_blade_to_component": (_u64, [_ci, _u64, C.POINTER(_ci)]),
    "gaast_blades_gp": (_dbl, [_ci, _pd, _u64, _u64, C.POINTER(_u64)]),
    "gaast_expr_retain": (_vp, [_vp]),
    "gaast_expr_release": (None, [_vp]),
    "gaast_expr_input": (_vp, [_ci, _u64, _ci]),
    "gaast_expr_const": (_vp, [_u64, _ci, _pd, _sz]),
    "gaast_expr_from_f64": (_vp, [_dbl]),
    "gaast_expr_basis_vector": (_vp, [_ci, _ci]),
    "gaast_expr_product": (_vp, [_vp, _vp, _ci]),
    "gaast_expr_product_custom": (_vp, [_vp, _vp, SELECT_FN, _vp]),
    "gaast_expr_add": (_vp, [_vp, _vp]),
    "gaast_expr_neg": (_vp, [_vp]),
    "gaast_expr_sub": (_vp, [_vp, _vp]),
    "gaast_expr_div_scalar": (_vp, [_vp, _dbl]),
    "gaast_expr_rev": (_vp, [_vp]),
    "gaast_expr_ginvol": (_vp, [_vp]),
    "gaast_expr_exp": (_vp, [_vp]),
    "gaast_expr_log": (_vp, [_vp]),
    "gaast_expr_pow": (_vp, [_vp, _vp]),
    "gaast_expr_sqrt": (_vp, [_vp]),
    "gaast_expr_g": (_vp, [_vp, _i64]),
    "gaast_expr_gselect_mask": (_vp, [_vp, _u64]),
    "gaast_expr_conj": (_vp, [_vp]),
    "gaast_expr_scal": (_vp, [_vp, _vp]),
    "gaast_expr_norm_sq": (_vp, [_vp]),
    "gaast_expr_sinv": (_vp, [_vp]),
    "gaast_expr_vinv": (_vp, [_vp]),
    "gaast_expr_specialize": (_vp, [_vp, _ci, _pd, _u64]),
    "gaast_spec_free": (None, [_vp]),
    "gaast_spec_num_nodes": (_ci, [_vp]),
    "gaast_spec_root": (_ci, [_vp]),
    "gaast_spec_node": (_ci, [_vp, _ci, C.POINTER(SpecNodeInfo)]),
    "gaast_spec_comp_muls": (C.POINTER(CompMul), [_vp, _ci]),
    "gaast_spec_program_desc": (_ci, [_vp, _ci, C.c_uint32, C.POINTER(ProgramDesc)]),
    "gaast_program_serialize": (_sz, [C.POINTER(ProgramDesc), _vp, _sz]),
    "gaast_program_deserialize": (_vp, [_vp, _sz]),
    "gaast_program_image_desc": (C.POINTER(ProgramDesc), [_vp]),
    "gaast_program_image_free": (None, [_vp]),
    "gaast_spec_num_inputs": (_ci, [_vp]),
    "gaast_spec_num_user_inputs": (_ci, [_vp]),
}

_lib = None
_device_ready = False


def lib():
    """The loaded shared library (host-side functions work without a GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it first (make -C gaast_amd/csrc). "
                              "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != GAAST_OK:
        raise GaastError(status, lib().gaast_hip_last_error().decode())


def init_device(device_id=None):
    """gaast_hip_init for this process (LOCAL_RANK picks the GPU when device_id is None)."""
    global _device_ready
    if _device_ready:
        return
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", "0"))
    ids = (C.c_int * 1)(device_id)
    check(lib().gaast_hip_init(ids, 1))
    _device_ready = True
