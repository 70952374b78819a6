"""Fuzz: random programs over the reference's operator surface, phases 1-3 of the product (C++ host
mirror) against the oracle -- node for node, grade set for grade set, comp-mul entry for entry; programs
that make the reference panic (oracle: OraclePanic) must make the product fail too.  No GPU needed."""
import pytest

import gaast_amd as ga
from fuzz import random_program, realise
from helpers import HipBackend, OracleBackend, assert_same_ast
from oracle import pyoracle as og

N_PROGRAMS = 300


def _specialize_both(seed):
    n, metric, spec = random_program(seed)
    o = h = None
    oerr = herr = None
    try:
        o = realise(spec, OracleBackend(), n).specialize(og.as_algebra(metric))
    except og.OraclePanic as e:
        oerr = e
    try:
        h = realise(spec, HipBackend(), n).specialize(metric, materialize_limit=0)
    except (ga.GaastError, RuntimeError) as e:
        herr = e
    return spec, o, h, oerr, herr


def test_random_programs_specialize_like_the_oracle():
    ok = panics = 0
    for seed in range(N_PROGRAMS):
        spec, o, h, oerr, herr = _specialize_both(seed)
        if oerr is not None or herr is not None:
            assert oerr is not None and herr is not None, f"seed {seed}: oracle {oerr!r} vs product {herr!r}\n{spec}"
            panics += 1
            continue
        try:
            assert_same_ast(o, h)
        except AssertionError as e:
            raise AssertionError(f"seed {seed}: {e}\n{spec}")
        ok += 1
    assert ok >= N_PROGRAMS // 2, (ok, panics)      # the generator must mostly produce valid programs
