"""Committed golden vectors (tests/golden/golden_eval.json, produced by the oracle with
tests/golden/gen_golden.py).  CPU: the oracle still reproduces them bit for bit.  GPU: the HIP
path reproduces them bit for bit without the oracle in the loop."""
import json
import os

import numpy as np
import pytest

import gaast_amd as ga
from golden_programs import PROGRAMS
from helpers import HipBackend, OracleBackend, hip_eval_batch, oracle_eval_batch
from oracle import pyoracle as og


def _load(golden_dir):
    with open(os.path.join(golden_dir, "golden_eval.json")) as f:
        return json.load(f)


def _rows(case):
    ins = {int(s): np.array([[float.fromhex(x) for x in row] for row in rows]) for s, rows in case["inputs"].items()}
    want = np.array([[float.fromhex(x) for x in row] for row in case["expected"]])
    return ins, want


@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_oracle_reproduces_golden(name, golden_dir):
    case = _load(golden_dir)[name]
    ins, want = _rows(case)
    got, mask = oracle_eval_batch(PROGRAMS[name]["build"], PROGRAMS[name]["metric"], ins, case["batch"])
    assert mask == case["out_mask"]
    assert np.array_equal(got, want) and np.array_equal(np.signbit(got), np.signbit(want))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 16, 2])   # specialised kernel; LDS interpreter; one launch per eval.rs arm
@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_hip_reproduces_golden(name, flags, golden_dir):
    case = _load(golden_dir)[name]
    ins, want = _rows(case)
    got, mask, spec = hip_eval_batch(PROGRAMS[name]["build"], PROGRAMS[name]["metric"], ins, case["batch"], flags=flags)
    assert mask == case["out_mask"]
    if any("fused multiply-adds under shared operands" in l for l in spec.launches()) and PROGRAMS[name].get("shared"):
        # round 4: without GAAST_FLAG_EXACT_ORDER the specialised kernel of an arithmetic-bound program runs its contracted variant
        # when an operand is shared by all items (one rounding per term instead of two: the tolerance contract, not the bits) ...
        assert flags == 0
        assert np.all(np.abs(got - want) <= 64 * np.finfo(np.float64).eps * np.abs(want).max()), spec.launches()
        # ... and with the flag the same program reproduces the reference's bits
        got, mask, spec = hip_eval_batch(PROGRAMS[name]["build"], PROGRAMS[name]["metric"], ins, case["batch"], flags=ga.FLAG_EXACT_ORDER)
        assert not any("fused multiply-adds" in l for l in spec.launches()), spec.launches()
    assert np.array_equal(got, want) and np.array_equal(np.signbit(got), np.signbit(want)), spec.launches()


def test_sum_of_different_grade_sets_is_a_reference_panic():
    """specialize.rs:75-79 hands `wanted` to both sides of an Addition unchanged, so a side is
    asked for grades it cannot produce and the assert of specialize.rs:113-117 fires -- unless
    the extra grades lie beyond the BitVec length of its maximal set (grade_set.rs:149-151).
    e1 + e1^e2 panics (the bivector is asked for grade 1); the product mirrors both outcomes."""
    e1, e2, _ = og.Expr.basis_vectors(3)
    with pytest.raises(og.OraclePanic):
        (e1 + (e1 ^ e2)).specialize(3)
    h1, h2, _ = ga.Expr.basis_vectors(3)
    with pytest.raises(ga.GaastError):
        (h1 + (h1 ^ h2)).specialize(3)
    # blind spot: {0,2} + {0} is accepted (the scalar side, a BitVec of length 1, never sees grade 2)
    def build(B):
        a, b, _ = B.basis_vectors(3)
        return (a ^ b) * (a ^ b) + (a & a)
    o = build(OracleBackend()).specialize(3)
    h = build(HipBackend()).specialize(3)
    assert [n.minimal for n in o.nodes()] == [n.minimal_grade_mask for n in h.nodes()]
