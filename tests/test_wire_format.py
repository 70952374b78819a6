"""Program wire format: the flat program survives a byte round trip."""
import ctypes as C

import numpy as np
import pytest

import gaast_amd as ga
from gaast_amd import _lib
from exprs import CASES
from helpers import HipBackend, full_grades, oracle_eval_batch, rows_of


def _desc_fields(d):
    nodes = [(n.opcode, n.child0, n.child1, n.minimal_grade_mask, n.vec_space_dim, n.input_slot, n.product_kind,
              n.n_comp_muls, bytes(C.string_at(n.comp_muls, n.n_comp_muls * 32)) if n.comp_muls else None)
             for n in (d.nodes[i] for i in range(d.n_nodes))]
    ins = []
    for i in range(d.n_inputs):
        x = d.inputs[i]
        rl = ga.graded.row_len(x.storage_dim, x.grade_mask)
        ins.append((x.grade_mask, x.storage_dim, x.is_const, [x.const_row[c] for c in range(rl)] if x.is_const else None))
    return (d.vec_space_dim, [d.metric_diag[i] for i in range(d.vec_space_dim)], d.dtype, d.root, d.flags, nodes, ins)


@pytest.mark.parametrize("name", sorted(CASES))
def test_roundtrip_is_lossless(name):
    alg, build = CASES[name]
    spec = build(HipBackend(), np.random.default_rng(7)).specialize(alg, flags=ga.FLAG_EXACT_ORDER)
    data = spec.serialize()
    img = ga.ProgramImage(data)
    assert _desc_fields(img.desc) == _desc_fields(spec.program_desc())
    assert ga.ProgramImage(data) is not None and len(data) > 40


def test_malformed_images_are_rejected():
    spec = (ga.mv(ga.Input(0, [1], 3)) * ga.mv(ga.Input(1, [1], 3))).specialize(3)
    data = spec.serialize()
    for bad in (b"", data[:-1], data + b"\0", b"X" + data[1:], data[:12] + b"\xff\xff\xff\x7f" + data[16:]):
        with pytest.raises(ga.GaastError):
            ga.ProgramImage(bad)


@pytest.mark.gpu
def test_deserialized_program_evaluates_bit_exact():
    batch = 200
    rng = np.random.default_rng(5)
    build = lambda B: (lambda r, x: r * x * r.rev())(B.input(0, [0, 2, 4], 5), B.input(1, [1], 5))
    cga = [1.0, 1.0, 1.0, 1.0, -1.0]
    rows = {0: rows_of(5, [0, 2, 4], batch, rng), 1: rows_of(5, [1], batch, rng)}
    want, _ = oracle_eval_batch(build, cga, rows, batch)
    data = build(HipBackend()).specialize(cga).serialize()
    out = ga.ProgramImage(data).eval_batch([rows[0], rows[1]], batch)
    ga.lib().gaast_hip_synchronize()
    assert np.array_equal(out.download_rows(), want)
