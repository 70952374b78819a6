"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes as C
import os
import re

import pytest

import gaast_amd
from gaast_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(gaast_[a-z0-9_]+)\s*\(", src)) - {"gaast_select_fn"}


@pytest.mark.parametrize("header", ["gaast_hip.h", "gaast_expr.h"])
def test_every_declared_symbol_is_exported(header):
    L = C.CDLL(_lib.LIB_PATH)
    names = _declared(header)
    assert len(names) > 15
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, f"declared in {header} but not exported: {missing}"


def test_python_binding_covers_the_headers():
    declared = _declared("gaast_hip.h") | _declared("gaast_expr.h")
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    e1, e2, _ = gaast_amd.Expr.basis_vectors(3)
    with pytest.raises(gaast_amd.GaastError) as ei:
        (e1 ^ e2).specialize(3).eval()
    assert ei.value.status_name == "NO_DEVICE"
