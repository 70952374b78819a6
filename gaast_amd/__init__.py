"""gaast_amd -- MI355X-native evaluator for gaast's specialized geometric-algebra ASTs.

Host side mirrors the reference's public surface (Expr, mv, GradeSet, GradeMapMV,
OrthoEuclidN, SpecializedAst); `SpecializedAst.eval*` runs phase 4 on the GPU through the C
ABI of include/gaast_hip.h (hand-written gfx950 kernels).  There is no CPU evaluation path.
"""
from ._lib import (F32, F64, FLAG_DEBUG_OVERFLOW, FLAG_EXACT_ORDER, FLAG_NO_FUSION, FLAG_NO_JIT, FLAG_NO_MFMA, FLAG_SPINOR_GEMM,
                   FLAG_DEBUG_JIT_FAILS, FLAG_DEBUG_KEEP_JIT_SOURCE, FLAG_EXP_LOG, FLAG_NO_COALESCE, FLAG_DEBUG_NO_CHAIN, GaastError,
                   init_device, lib)
from .algebra import MetricAlgebra, OrthoEuclidN, n_choose_k
from .ast import Expr, Input, ProgramImage, SpecializedAst, mv
from .grade_set import GradeSet
from .graded import DeviceMV, GradeMapMV, grade_map_mv

__all__ = ["Expr", "Input", "ProgramImage", "SpecializedAst", "mv", "GradeSet", "GradeMapMV", "grade_map_mv", "DeviceMV",
           "MetricAlgebra", "OrthoEuclidN", "n_choose_k", "GaastError", "init_device", "lib",
           "F32", "F64", "FLAG_DEBUG_OVERFLOW", "FLAG_EXACT_ORDER", "FLAG_NO_FUSION", "FLAG_NO_MFMA", "FLAG_NO_JIT", "FLAG_SPINOR_GEMM",
           "FLAG_DEBUG_JIT_FAILS", "FLAG_DEBUG_KEEP_JIT_SOURCE", "FLAG_EXP_LOG", "FLAG_NO_COALESCE", "FLAG_DEBUG_NO_CHAIN"]
