"""The C ABI exercised the way a foreign host (the Rust shim of INTEGRATION.md) would use it:
flat program structs built by hand, explicit comp-mul lists, wrapped device memory, strides,
shared (batch-1) operands, f32 programs, malformed programs."""
import ctypes as C

import numpy as np
import pytest

import gaast_amd as ga
from gaast_amd import _lib
from helpers import (HipBackend, OracleBackend, bits_to_row, full_grades, gp_bits, hip_eval_batch, n_choose_k,
                     oracle_eval_batch, row_to_bits, rows_of)

pytestmark = pytest.mark.gpu


def _raw_program(spec, kind_override=None, flags=0, dtype=_lib.F64):
    """Re-create the program of `spec` through raw structs, as a host that only sees the public
    read API would: explicit lists, product_kind = EXPLICIT (the closure is opaque to it)."""
    L = _lib.lib()
    d = spec.program_desc()
    nodes = (_lib.NodeDesc * d.n_nodes)()
    keep = []
    for i in range(d.n_nodes):
        src = d.nodes[i]
        nodes[i].opcode, nodes[i].child0, nodes[i].child1 = src.opcode, src.child0, src.child1
        nodes[i].minimal_grade_mask, nodes[i].vec_space_dim = src.minimal_grade_mask, src.vec_space_dim
        nodes[i].input_slot = src.input_slot
        nodes[i].product_kind = _lib.PROD_EXPLICIT if kind_override is None else kind_override
        nodes[i].n_comp_muls = src.n_comp_muls
        if src.opcode == _lib.OP_PRODUCT and src.n_comp_muls:
            arr = (_lib.CompMul * src.n_comp_muls)()
            C.memmove(arr, src.comp_muls, C.sizeof(_lib.CompMul) * src.n_comp_muls)
            keep.append(arr)
            nodes[i].comp_muls = arr
    out = _lib.ProgramDesc()
    out.vec_space_dim, out.metric_diag, out.dtype = d.vec_space_dim, d.metric_diag, dtype
    out.n_nodes, out.nodes, out.root = d.n_nodes, nodes, d.root
    out.n_inputs, out.inputs, out.flags = d.n_inputs, d.inputs, flags
    keep.append(nodes)
    return out, keep


def _run_raw(desc, inputs, batch, out_dim):
    L = _lib.lib()
    _lib.init_device()
    prog = C.c_void_p()
    _lib.check(L.gaast_hip_program_create(C.byref(desc), C.byref(prog)))
    mask, rl = C.c_uint64(), C.c_int64()
    _lib.check(L.gaast_hip_program_output_info(prog, C.byref(mask), C.byref(rl)))
    out = ga.DeviceMV.alloc(out_dim, ga.GradeSet(mask.value), batch, desc.dtype)
    handles = (C.c_void_p * max(1, len(inputs)))(*[m._h for m in inputs])
    _lib.check(L.gaast_hip_eval(prog, handles, len(inputs), batch, out._h))
    _lib.check(L.gaast_hip_synchronize())
    rows = out.download_rows()
    names = [L.gaast_hip_program_launch_name(prog, i).decode() for i in range(L.gaast_hip_program_num_launches(prog))]
    _lib.check(L.gaast_hip_program_destroy(prog))
    return rows, names


def test_explicit_lists_from_a_foreign_host_are_exact():
    """BASELINE config 5 through hand-built structs (product_kind unknown): bit-exact."""
    batch = 300
    rng = np.random.default_rng(5)
    build = lambda B: (lambda r, x: r * x * r.rev())(B.input(0, [0, 2, 4], 5), B.input(1, [1], 5))
    cga = [1.0, 1.0, 1.0, 1.0, -1.0]
    rows = {0: rows_of(5, [0, 2, 4], batch, rng), 1: rows_of(5, [1], batch, rng)}
    want, _ = oracle_eval_batch(build, cga, rows, batch)
    spec = build(HipBackend()).specialize(cga, materialize_limit=0)
    desc, keep = _raw_program(spec)
    ins = [ga.DeviceMV.from_rows(5, [0, 2, 4], rows[0]), ga.DeviceMV.from_rows(5, [1], rows[1])]
    got, names = _run_raw(desc, ins, batch, 5)
    assert np.array_equal(got, want), names


def test_explicit_dense_list_is_recognised_only_when_it_is_the_geometric_product():
    """A foreign host cannot name the product (closure opaque): the library compares the list with
    the geometric product's; an exact match runs on the dense kernel, anything else stays on the
    exact kernels."""
    n, batch = 6, 17
    rng = np.random.default_rng(6)
    build = lambda B: B.input(0, full_grades(n), n) * B.input(1, full_grades(n), n)
    rows = {0: rows_of(n, full_grades(n), batch, rng), 1: rows_of(n, full_grades(n), batch, rng)}
    want, _ = oracle_eval_batch(build, n, rows, batch)
    spec = build(HipBackend()).specialize(n, materialize_limit=0)
    ins = [ga.DeviceMV.from_rows(n, full_grades(n), rows[s]) for s in range(2)]
    desc, keep = _raw_program(spec)                                   # kind unknown, list untouched
    got, names = _run_raw(desc, ins, batch, n)
    assert any("dense" in x for x in names) and np.allclose(got, want, rtol=0, atol=1e-12)
    desc, keep = _raw_program(spec)
    muls = desc.nodes[2].comp_muls                                    # swap two entries of different outputs:
    tmp = _lib.CompMul()                                              # same sums, but no longer THE list
    C.memmove(C.byref(tmp), C.byref(muls[0]), C.sizeof(_lib.CompMul))
    C.memmove(C.byref(muls[0]), C.byref(muls[1]), C.sizeof(_lib.CompMul))
    C.memmove(C.byref(muls[1]), C.byref(tmp), C.sizeof(_lib.CompMul))
    assert (muls[0].result_grade, muls[0].result_index) != (muls[1].result_grade, muls[1].result_index)
    got, names = _run_raw(desc, ins, batch, n)
    # (64 rows of 64 terms: the specialised list kernel; without GAAST_FLAG_EXACT_ORDER its sums use fused multiply-adds since round 4:
    #  the tolerance contract of every re-ordered kernel, here far inside 1e-12; the reference's bits under the flag, below)
    assert not any("dense" in x for x in names) and np.allclose(got, want, rtol=0, atol=1e-12)
    desc.flags = _lib.FLAG_EXACT_ORDER
    got, names = _run_raw(desc, ins, batch, n)
    assert not any("dense" in x for x in names) and np.array_equal(got, want)
    desc, keep = _raw_program(spec, flags=_lib.FLAG_EXACT_ORDER)      # the host insists on the reference order
    got, names = _run_raw(desc, ins, batch, n)
    assert not any("dense" in x for x in names) and np.array_equal(got, want)


def test_wrapped_memory_with_row_stride_and_shared_operand():
    """Rows inside a wider caller-owned buffer (row_stride > row_len); left operand shared (batch 1)."""
    import torch
    n, batch = 7, 19
    rng = np.random.default_rng(7)
    a = rows_of(n, full_grades(n), 1, rng)
    b = rows_of(n, full_grades(n), batch, rng)
    build = lambda B: B.input(0, full_grades(n), n) * B.input(1, full_grades(n), n)
    want, _ = oracle_eval_batch(build, n, {0: a, 1: b}, batch)
    ga.init_device()
    L = _lib.lib()
    N = 1 << n
    big_b = torch.full((batch, N + 24), 777.0, dtype=torch.float64, device="cuda")
    big_o = torch.full((batch, N + 8), -5.0, dtype=torch.float64, device="cuda")
    big_b[:, :N] = torch.from_numpy(b).cuda()
    hb, ho = C.c_void_p(), C.c_void_p()
    mask = (1 << (n + 1)) - 1
    _lib.check(L.gaast_hip_mv_wrap(C.c_void_p(big_b.data_ptr()), n, mask, batch, _lib.F64, N + 24, C.byref(hb)))
    _lib.check(L.gaast_hip_mv_wrap(C.c_void_p(big_o.data_ptr()), n, mask, batch, _lib.F64, N + 8, C.byref(ho)))
    wb = ga.DeviceMV(hb, n, mask, batch, _lib.F64, keep=big_b)
    wo = ga.DeviceMV(ho, n, mask, batch, _lib.F64, keep=big_o)
    spec = build(HipBackend()).specialize(n)
    spec.eval_batch([ga.DeviceMV.from_rows(n, full_grades(n), a), wb], batch, out=wo)
    torch.cuda.synchronize()
    L.gaast_hip_synchronize()
    res = big_o.cpu().numpy()
    assert np.allclose(res[:, :N], want, rtol=0, atol=1e-12)
    assert np.all(res[:, N:] == -5.0), "wrote outside the rows"
    assert np.all(big_b.cpu().numpy()[:, N:] == 777.0)


@pytest.mark.parametrize("flags,kernel", [(0, "ast_jit"), (ga.FLAG_NO_JIT, "ast_fused"), (ga.FLAG_NO_FUSION, "k_product_csr<float>")])
@pytest.mark.parametrize("name", ["cfg1", "sandwich", "kat_projection"])
def test_f32_extension_of_the_exact_kernels_is_bit_exact(name, flags, kernel):
    """f32 is an extension of the build (the reference is f64-only, graded.rs:46): the exact kernels execute the reference's
    statements in its order on binary32 values, so they must equal the oracle's F32 MODE (every operand and result rounded to
    binary32, same order, no contraction; pinned by the reference's four eval KATs) BIT FOR BIT -- specialised kernel,
    LDS interpreter and the unfused one-kernel-per-arm plan alike.  A wrong sign on a small term cannot hide in a tolerance."""
    from oracle import pyoracle as og
    batch = 257
    rng = np.random.default_rng(9)
    if name == "cfg1":
        build = lambda B: (B.input(0, full_grades(3), 3) + B.input(1, full_grades(3), 3) * B.input(2, full_grades(3), 3)).g(2)
        alg, rows = 3, {s: rows_of(3, full_grades(3), batch, rng, np.float32) for s in range(3)}
    elif name == "sandwich":
        build = lambda B: (lambda r, x: r * x * r.rev())(B.input(0, [0, 2, 4], 5), B.input(1, [1], 5))
        alg = [1.0, 1.0, 1.0, 1.0, -1.0]
        rows = {0: rows_of(5, [0, 2, 4], batch, rng, np.float32), 1: rows_of(5, [1], batch, rng, np.float32)}
    else:   # the projection KAT of eval.rs:152-163 on batched inputs: &, *, rev, norm_sq, sinv (binary32 division), a shared node
        build = lambda B: (lambda v, bv: (v & bv) & bv.vinv())(B.input(0, [1], 3), B.input(1, [2], 3))
        alg, rows = 3, {0: rows_of(3, [1], batch, rng, np.float32), 1: rows_of(3, [2], batch, rng, np.float32)}
    rows64 = {s: r.astype(np.float64) for s, r in rows.items()}
    want, wmask = oracle_eval_batch(build, alg, rows64, batch, mode=og.EVAL_F32)
    want64, _ = oracle_eval_batch(build, alg, rows64, batch)
    got, mask, spec = hip_eval_batch(build, alg, rows, batch, dtype=ga.F32, flags=flags)
    assert mask == wmask and any(kernel in l for l in spec.launches()), spec.launches()
    assert got.dtype == np.float32 and np.array_equal(got.astype(np.float64), want), float(np.abs(got - want).max())
    # the F32 mode is doing something: it differs from the rounded f64 evaluation somewhere in the batch
    assert not np.array_equal(want, want64.astype(np.float32).astype(np.float64))


def test_malformed_programs_are_rejected_not_executed():
    L = _lib.lib()
    _lib.init_device()
    spec = (ga.mv(ga.Input(0, [1], 3)) * ga.mv(ga.Input(1, [1], 3))).specialize(3)
    desc, keep = _raw_program(spec)
    desc.nodes[2].child0 = 2                      # not post-order: a node referring to itself
    prog = C.c_void_p()
    assert L.gaast_hip_program_create(C.byref(desc), C.byref(prog)) == 1   # INVALID_PROGRAM
    desc, keep = _raw_program(spec)
    desc.nodes[2].comp_muls[0].left_index = 99   # index beyond C(3,1)
    assert L.gaast_hip_program_create(C.byref(desc), C.byref(prog)) == 1
    desc, keep = _raw_program(spec)
    desc.nodes[0].input_slot = 7                 # slot out of range
    assert L.gaast_hip_program_create(C.byref(desc), C.byref(prog)) == 1
    # wrong grade set / dtype / batch of a bound input -> INVALID_ARGUMENT at eval
    with pytest.raises(ga.GaastError) as ei:
        spec.eval_batch([np.zeros((4, 3)), ga.DeviceMV.alloc(3, [2], 4)], 4)
    assert ei.value.status_name == "INVALID_ARGUMENT"
    with pytest.raises(ga.GaastError):
        spec.eval_batch([np.zeros((4, 3)), np.zeros((3, 3))], 4)


def test_q2_missing_grade_is_a_status():
    """A GradeProjection inside an Addition chain whose shared child wants more grades than the
    result buffer holds: the reference panics in grade_slice_mut (graded.rs:192-194)."""
    def build(B):
        a = B.input(0, [1], 3)
        b = B.input(1, [1], 3)
        p = a * b                      # {0,2}, shared
        return p.g(0) + (p * p).g(0)   # root {0}; p is asked for {0,2} by the product, {0} by the projection
    rows = {0: np.array([[1.0, 2.0, 3.0]]), 1: np.array([[0.5, -1.0, 2.0]])}
    try:
        want, _ = oracle_eval_batch(build, 3, rows, 1)
        got, _, _ = hip_eval_batch(build, 3, rows, 1)
        assert np.array_equal(got, want)
    except Exception as ref:
        from oracle import pyoracle as og
        assert isinstance(ref, og.OraclePanic) and ref.code == 1
        with pytest.raises(ga.GaastError) as ei:
            hip_eval_batch(build, 3, rows, 1)
        assert ei.value.status_name == "MISSING_GRADE"


def test_a_program_beyond_the_back_end_is_refused_whole_at_program_create():
    """n = 14: three middle grades times a full multivector stage 206 KB of operands per item (f64) -- more than the
    160 KiB of LDS the list kernels have, and no dense kernel exists beyond n = 13.  The program is valid in the
    reference; gaast_hip_program_create says UNIMPLEMENTED (not INVALID_PROGRAM, and not at eval time with half of
    the launches already queued)."""
    n = 14
    a, b = ga.mv(ga.Input(0, [6, 7, 8], n)), ga.mv(ga.Input(1, full_grades(n), n))
    spec = (a * b).specialize(n)
    with pytest.raises(ga.GaastError) as ei:
        spec.program()
    assert ei.value.status_name == "UNIMPLEMENTED" and "LDS" in str(ei.value)
    # the full product in f32 fits the LDS (128 KiB): it runs on the matrix-core kernel (test_gpu_dense_oracle.py); kept off
    # it, its 4^14-entry list exceeds the table budget of the list kernels
    full = (ga.mv(ga.Input(0, full_grades(n), n)) * b).specialize(n, dtype=ga.F32, flags=ga.FLAG_NO_MFMA)
    with pytest.raises(ga.GaastError) as ei:
        full.program()
    assert ei.value.status_name == "UNIMPLEMENTED" and "table budget" in str(ei.value)
    # ... while a product of the same algebra that does fit runs, bit-exact
    rng = np.random.default_rng(14)
    build = lambda B: B.input(0, [1], n) * B.input(1, [1, 2], n)
    rows = {0: rows_of(n, [1], 5, rng), 1: rows_of(n, [1, 2], 5, rng)}
    want, _ = oracle_eval_batch(build, n, rows, 5)
    got, _, _ = hip_eval_batch(build, n, rows, 5)
    assert np.array_equal(got, want)


def test_wrapped_single_row_with_zero_stride_is_well_formed():
    """gaast_hip_mv_wrap(batch = 1, row_stride = 0): the stride of a single row is normalised to the row length, so zero
    fill, upload and download (2-D copies with pitch >= width) work"""
    import torch
    ga.init_device()
    L = _lib.lib()
    t = torch.full((1, 8), 3.0, dtype=torch.float64, device="cuda")
    h = C.c_void_p()
    _lib.check(L.gaast_hip_mv_wrap(C.c_void_p(t.data_ptr()), 3, 0xF, 1, _lib.F64, 0, C.byref(h)))
    stride = C.c_int64()
    _lib.check(L.gaast_hip_mv_info(h, None, None, None, None, None, C.byref(stride), None))
    assert stride.value == 8
    _lib.check(L.gaast_hip_mv_zero(h))
    _lib.check(L.gaast_hip_synchronize())
    assert float(t.abs().sum()) == 0.0
    row = np.arange(8, dtype=np.float64)
    _lib.check(L.gaast_hip_mv_upload_rows(h, row.ctypes.data_as(C.c_void_p), 8))
    assert np.array_equal(t.cpu().numpy()[0], row)
    _lib.check(L.gaast_hip_mv_free(h))


def test_entry_points_make_the_library_device_current_on_any_thread():
    """HIP's current device is per host thread: an eval driven from a second thread (a Rust worker, torch's autograd
    thread) runs on the library's device and gives the same bits"""
    import threading
    rng = np.random.default_rng(3)
    build = lambda B: (B.input(0, full_grades(3), 3) + B.input(1, full_grades(3), 3) * B.input(2, full_grades(3), 3)).g(2)
    rows = {s: rows_of(3, full_grades(3), 100, rng) for s in range(3)}
    want, _ = oracle_eval_batch(build, 3, rows, 100)
    box = {}

    def work():
        try:
            box["got"] = hip_eval_batch(build, 3, rows, 100)[0]
        except Exception as e:      # pragma: no cover
            box["err"] = e
    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert "err" not in box and np.array_equal(box["got"], want)


def test_eval_gather_chunk_schedule_on_a_one_rank_communicator():
    """gaast_hip_eval_gather == gaast_hip_eval + gaast_hip_gather_rows for every chunking (more chunks than items, one
    chunk, ragged chunks), into a separate destination and into `out` itself; argument errors are statuses."""
    L = _lib.lib()
    ga.init_device()
    rng = np.random.default_rng(77)
    build = lambda B: (lambda r, x: r * x * r.rev())(B.input(0, [0, 2, 4], 5), B.input(1, [1], 5))
    cga = [1.0, 1.0, 1.0, 1.0, -1.0]
    spec = build(HipBackend()).specialize(cga)
    # without a communicator: a status, not a crash
    cnt1 = (C.c_int64 * 1)(5)
    o5 = ga.DeviceMV.alloc(5, [1, 3, 5], 5)
    assert L.gaast_hip_gather_rows(o5._h, o5._h, cnt1, 0) == 5          # GAAST_ERR_RCCL
    idbuf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
    _lib.check(L.gaast_hip_comm_unique_id(idbuf))
    _lib.check(L.gaast_hip_comm_init(idbuf, 0, 1))
    try:
        assert L.gaast_hip_comm_init(idbuf, 0, 1) == 5                  # a second communicator is refused
        n = C.c_int()
        _lib.check(L.gaast_hip_comm_count_ranks(C.byref(n)))
        assert n.value == 1
        for batch, chunks in ((1, 4), (3, 7), (64, 1), (257, 4), (1000, 3)):
            rows = {0: rows_of(5, [0, 2, 4], batch, rng), 1: rows_of(5, [1], batch, rng)}
            want, _ = oracle_eval_batch(build, cga, rows, batch)
            ins = [ga.DeviceMV.from_rows(5, [0, 2, 4], rows[0]), ga.DeviceMV.from_rows(5, [1], rows[1])]
            out = ga.DeviceMV.alloc(5, [1, 3, 5], batch)
            gathered = ga.DeviceMV.alloc(5, [1, 3, 5], batch)
            spec.eval_gather(ins, out, gathered, [batch], root=0, n_chunks=chunks)
            _lib.check(L.gaast_hip_synchronize())
            assert np.array_equal(gathered.download_rows(), want) and np.array_equal(out.download_rows(), want)
            out2 = ga.DeviceMV.alloc(5, [1, 3, 5], batch)               # `out` is the root's range of `gathered`
            spec.eval_gather(ins, out2, out2, [batch], root=0, n_chunks=chunks)
            _lib.check(L.gaast_hip_synchronize())
            assert np.array_equal(out2.download_rows(), want)
        # argument checks
        bad = (C.c_int64 * 1)(2000)
        handles = (C.c_void_p * 2)(ins[0]._h, ins[1]._h)
        assert L.gaast_hip_eval_gather(spec.program(), handles, 2, out._h, gathered._h, bad, 0, 4) == 6       # counts[rank] > batch
        ok = (C.c_int64 * 1)(1000)
        assert L.gaast_hip_eval_gather(spec.program(), handles, 2, out._h, gathered._h, ok, 0, 0) == 6        # n_chunks < 1
        assert L.gaast_hip_eval_gather(spec.program(), handles, 2, out._h, gathered._h, ok, 1, 4) == 6        # root out of range
        assert L.gaast_hip_eval_gather(spec.program(), handles, 2, out._h, None, ok, 0, 4) == 6               # root without destination
    finally:
        _lib.check(L.gaast_hip_comm_destroy())
    r, w = C.c_int(), C.c_int()
    assert L.gaast_hip_comm_info(C.byref(r), C.byref(w)) == 5


@pytest.mark.parametrize("case", ["cl41_span", "r5_lines"])
@pytest.mark.parametrize("layout", ["contiguous", "strided_aligned", "strided_odd", "misaligned_base"])
def test_specialised_kernels_row_io_forms_are_bit_exact(case, layout):
    """The hiprtc kernels pick their row I/O per operand at run time: one coalesced span through LDS (contiguous, 16-byte
    aligned rows), a cache line of every row at a time (long rows; any 16-byte-multiple stride), or every lane its own row
    (odd strides, misaligned bases, the last partial wave).  Same bits in every form; the batch ends in a partial wave."""
    import torch
    ga.init_device()
    batch = 64 * 5 + 37
    rng = np.random.default_rng(31)
    if case == "cl41_span":
        n, alg = 5, [1.0, 1.0, 1.0, 1.0, -1.0]
        grades = [[0, 2, 4], [1]]
        build = lambda B: (lambda r, x: r * x * r.rev())(B.input(0, [0, 2, 4], n), B.input(1, [1], n))
        flags = 0
    else:
        n, alg = 5, 5
        grades = [full_grades(5), full_grades(5)]
        build = lambda B: B.input(0, full_grades(n), n) * B.input(1, full_grades(n), n)
        flags = ga.FLAG_EXACT_ORDER
    rows = {s: rows_of(n, g, batch, rng) for s, g in enumerate(grades)}
    want, wmask = oracle_eval_batch(build, alg, rows, batch)
    spec = build(HipBackend()).specialize(alg, flags=flags)
    assert any("ast_jit" in l for l in spec.launches()), spec.launches()
    pad = {"contiguous": 0, "strided_aligned": 6, "strided_odd": 3, "misaligned_base": 0}[layout]   # doubles of padding per row
    shift = 1 if layout == "misaligned_base" else 0                                                  # base pointer off by 8 bytes
    keep, ins = [], []
    for s, g in enumerate(grades):
        rl = rows[s].shape[1]
        flat = torch.full((batch * (rl + pad) + 2,), 777.0, dtype=torch.float64, device="cuda")
        view = flat[shift:shift + batch * (rl + pad)].view(batch, rl + pad)
        view[:, :rl] = torch.from_numpy(rows[s]).cuda()
        h = C.c_void_p()
        _lib.check(_lib.lib().gaast_hip_mv_wrap(C.c_void_p(view.data_ptr()), n, ga.graded._mask_of(g), batch, _lib.F64, rl + pad, C.byref(h)))
        keep.append((flat, view))
        ins.append(h)
    orl = want.shape[1]
    oflat = torch.full((batch * (orl + pad) + 2,), -5.0, dtype=torch.float64, device="cuda")
    oview = oflat[shift:shift + batch * (orl + pad)].view(batch, orl + pad)
    ho = C.c_void_p()
    _lib.check(_lib.lib().gaast_hip_mv_wrap(C.c_void_p(oview.data_ptr()), n, wmask, batch, _lib.F64, orl + pad, C.byref(ho)))
    handles = (C.c_void_p * 2)(*ins)
    _lib.check(_lib.lib().gaast_hip_eval(spec.program(), handles, 2, batch, ho))
    _lib.check(_lib.lib().gaast_hip_synchronize())
    got = oview[:, :orl].cpu().numpy()
    assert np.array_equal(got, want)
    if pad:
        assert torch.all(oview[:, orl:] == -5.0)       # nothing written between the rows
    assert float(oflat[0]) == -5.0 or shift == 0        # ... nor before a shifted base
    for h in ins + [ho]:
        _lib.check(_lib.lib().gaast_hip_mv_free(h))
