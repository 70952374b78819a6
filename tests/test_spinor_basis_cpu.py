"""Host-side index bookkeeping of the opt-in matrix-representation kernel, checked without a GPU:
the C++ header the library uses (gaast_amd/csrc/device/spinor_basis.hpp) is compiled into a small
harness with g++ and run over every (alpha, lambda) pair of m = 3..6; the numpy prototype of the whole
one-plane algorithm (tools/proto/spinor_single_plane.py) is run over every signature of n = 6."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spinor_basis_header_exhaustive(tmp_path):
    exe = tmp_path / "spinor_basis_check"
    src = os.path.join(ROOT, "tests", "cpp", "spinor_basis_check.cpp")
    inc = os.path.join(ROOT, "gaast_amd", "csrc", "device")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", inc, src, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("OK "), out


def test_one_plane_algorithm_prototype_every_signature_of_n6():
    sys.path.insert(0, os.path.join(ROOT, "tools", "proto"))
    import itertools
    from helpers import gp_bits
    from spinor_single_plane import product_single_plane
    rng = np.random.default_rng(2)
    cases = set()
    for signs in itertools.product([1.0, -1.0], repeat=6):
        A, B = rng.uniform(-1, 1, 64), rng.uniform(-1, 1, 64)
        got, info = product_single_plane(6, list(signs), A, B)
        assert np.abs(got - gp_bits(6, list(signs), A, B)).max() < 1e-12, signs
        cases.add(info[2:])
    # lambda = 0 with / without alpha, lambda on the second bit, lambda on the top bit with / without alpha
    assert cases == {(-1, False), (-1, True), (1, True), (2, False), (2, True)}
