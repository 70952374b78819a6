"""The C ABI's struct layout, three ways: what gcc makes of include/gaast_hip.h (tests/cpp/abi_layout.c), the ctypes
mirrors the Python host uses (gaast_amd/_lib.py) and the constants committed next to the Rust bindings
(integration/rust/abi_layout.json + the compile-time assertions of ffi.rs).  A drift between any two fails here.
Also: the self-proving property the Rust shim relies on to send GAAST_PROD_GEOMETRIC instead of a 537 MB list."""
import ctypes as C
import itertools
import json
import os
import re
import subprocess

import numpy as np
import pytest

import gaast_amd as ga
from gaast_amd import _lib
from math import comb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIRRORS = {"gaast_comp_mul": _lib.CompMul, "gaast_node_desc": _lib.NodeDesc, "gaast_input_desc": _lib.InputDesc,
           "gaast_program_desc": _lib.ProgramDesc}


@pytest.fixture(scope="module")
def c_layout(tmp_path_factory):
    exe = tmp_path_factory.mktemp("abi") / "abi_layout"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "abi_layout.c"), "-o", str(exe)], check=True)
    return json.loads(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)


def test_ctypes_mirrors_match_the_header(c_layout):
    assert set(c_layout) == set(MIRRORS)
    for name, cls in MIRRORS.items():
        want = c_layout[name]
        assert C.sizeof(cls) == want["sizeof"], name
        fields = [f[0] for f in cls._fields_]
        assert fields == [k for k in want if k != "sizeof"], name          # same fields, same order
        for f in fields:
            assert getattr(cls, f).offset == want[f], (name, f)


def test_committed_rust_constants_match_the_header(c_layout):
    committed = json.load(open(os.path.join(ROOT, "integration", "rust", "abi_layout.json")))
    assert committed == c_layout
    ffi = open(os.path.join(ROOT, "integration", "rust", "src", "ffi.rs")).read()
    for rust, cname in (("GaastCompMul", "gaast_comp_mul"), ("GaastNodeDesc", "gaast_node_desc"),
                        ("GaastInputDesc", "gaast_input_desc"), ("GaastProgramDesc", "gaast_program_desc")):
        m = re.search(r"size_of::<%s>\(\) == (\d+)" % rust, ffi)
        assert m and int(m.group(1)) == c_layout[cname]["sizeof"], rust
        # #[repr(C)] field order in ffi.rs == the header's
        body = re.search(r"pub struct %s \{(.*?)\n\}" % rust, ffi, re.S).group(1)
        assert re.findall(r"pub (\w+):", body) == [k for k in c_layout[cname] if k != "sizeof"], rust
    for const, (cname, field) in {"LAYOUT_COMP_MUL_COEFF_OFFSET": ("gaast_comp_mul", "coeff"),
                                  "LAYOUT_NODE_DESC_MINIMAL_GRADE_MASK_OFFSET": ("gaast_node_desc", "minimal_grade_mask"),
                                  "LAYOUT_NODE_DESC_N_COMP_MULS_OFFSET": ("gaast_node_desc", "n_comp_muls"),
                                  "LAYOUT_NODE_DESC_COMP_MULS_OFFSET": ("gaast_node_desc", "comp_muls"),
                                  "LAYOUT_INPUT_DESC_CONST_ROW_OFFSET": ("gaast_input_desc", "const_row"),
                                  "LAYOUT_PROGRAM_DESC_NODES_OFFSET": ("gaast_program_desc", "nodes"),
                                  "LAYOUT_PROGRAM_DESC_INPUTS_OFFSET": ("gaast_program_desc", "inputs"),
                                  "LAYOUT_PROGRAM_DESC_FLAGS_OFFSET": ("gaast_program_desc", "flags")}.items():
        m = re.search(r"pub const %s: usize = (\d+);" % const, ffi)
        assert m and int(m.group(1)) == c_layout[cname][field], const


def test_rust_bindings_declare_entry_points_the_library_exports():
    ffi = open(os.path.join(ROOT, "integration", "rust", "src", "ffi.rs")).read()
    names = re.findall(r"pub fn (gaast_hip_\w+)\(", ffi)
    assert len(names) >= 18
    L = C.CDLL(_lib.LIB_PATH)
    assert all(hasattr(L, n) for n in names)
    for n in names:     # same number of parameters as the Python signature table (which the header test pins)
        args = re.search(r"pub fn %s\((.*?)\)" % n, ffi, re.S).group(1)
        n_args = len([a for a in args.split(",") if a.strip()])
        assert n_args == len(_lib.SIGNATURES[n][1]), n


def _pair_count(n, lmask, rmask, omask):
    """eval_hip.rs::geometric_pair_count"""
    total = 0
    for kl in range(n + 1):
        for kr in range(n + 1):
            if not ((lmask >> kl) & 1 and (rmask >> kr) & 1):
                continue
            for j in range(min(kl, kr) + 1):
                g = kl + kr - 2 * j
                if kr - j > n - kl or not (omask >> g) & 1:
                    continue
                total += comb(n, kl) * comb(kl, j) * comb(n - kl, kr - j)
    return total


@pytest.mark.parametrize("n,metric", [(4, [1.0, 1.0, 1.0, 1.0]), (5, [1.0, 1.0, 1.0, 1.0, -1.0]), (4, [0.0, 1.0, 1.0, 1.0])])
def test_a_list_as_long_as_the_geometric_products_is_the_geometric_products(n, metric):
    """What lets the Rust shim send product_kind = GEOMETRIC although grades_to_produce is opaque to it: among the
    reference's five products (expr.rs:180-197) only a list that equals the geometric product's, entry for entry, has
    the geometric product's length for the same operand / result grade sets."""
    rng = np.random.default_rng(n)
    ops = {"gp": lambda a, b: a * b, "outer": lambda a, b: a ^ b, "inner": lambda a, b: a & b,
           "lc": lambda a, b: a << b, "rc": lambda a, b: a >> b}
    checked = 0
    for _ in range(40):
        ga_ = sorted(set(rng.integers(0, n + 1, rng.integers(1, n + 2)).tolist()))
        gb_ = sorted(set(rng.integers(0, n + 1, rng.integers(1, n + 2)).tolist()))
        want_g = sorted(set(rng.integers(0, n + 1, rng.integers(1, n + 2)).tolist()))
        a, b = ga.mv(ga.Input(0, ga_, n)), ga.mv(ga.Input(1, gb_, n))
        lists = {}
        for name, op in ops.items():
            spec = op(a, b).gselect(want_g).specialize(ga.MetricAlgebra(metric), materialize_limit=1 << 30)
            prod = [i for i, nd in enumerate(spec.nodes()) if nd.opcode == _lib.OP_PRODUCT]
            if not prod:
                continue
            nd = spec.get_node(prod[0])
            l, r = spec.get_node(nd.child0), spec.get_node(nd.child1)
            lists[name] = (spec.comp_muls(prod[0]) or [], l.minimal_grade_mask, r.minimal_grade_mask, nd.minimal_grade_mask)
        for name, (lst, lm, rm, om) in lists.items():
            count_says_geometric = _pair_count(n, lm, rm, om) == len(lst)
            # the geometric product's list for the SAME grade sets
            spec = (ga.mv(ga.Input(0, [k for k in range(n + 1) if (lm >> k) & 1], n)) *
                    ga.mv(ga.Input(1, [k for k in range(n + 1) if (rm >> k) & 1], n))
                    ).gselect([k for k in range(n + 1) if (om >> k) & 1]).specialize(ga.MetricAlgebra(metric), materialize_limit=1 << 30)
            prod = [i for i, nd in enumerate(spec.nodes()) if nd.opcode == _lib.OP_PRODUCT]
            gp_list = (spec.comp_muls(prod[0]) or []) if prod else []
            gp_node = spec.get_node(prod[0]) if prod else None
            same_sets = gp_node is not None and (spec.get_node(gp_node.child0).minimal_grade_mask, spec.get_node(gp_node.child1).minimal_grade_mask,
                                                 gp_node.minimal_grade_mask) == (lm, rm, om)
            if not same_sets:
                continue      # the geometric product would prune the operands differently: not comparable
            assert count_says_geometric == (lst == gp_list), (name, ga_, gb_, want_g)
            checked += 1
    assert checked > 30
