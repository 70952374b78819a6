"""Shared test plumbing: one expression source, two back ends.

`ORACLE` wraps oracle/pyoracle.py (the CPU restatement of the reference: the checker) and
`HIP` wraps gaast_amd (the product).  An expression is written once as a function of a
back end `B` using the reference's operator surface, e.g.

    lambda B: (B.value(a) + B.value(b) * B.value(c)).g(2)

and then specialised / evaluated by both.
"""
from __future__ import annotations

import numpy as np

import gaast_amd as ga
from oracle import pyoracle as og


def n_choose_k(n, k):
    return og.lib().og_n_choose_k(n, k)


def full_grades(n):
    return list(range(n + 1))


def random_mv(rng, n, grades, lo=-1.0, hi=1.0):
    """dict grade -> f64 array of C(n,k) uniform values"""
    return {k: rng.uniform(lo, hi, n_choose_k(n, k)) for k in grades}


def rows_of(n, grades, batch, rng, dtype=np.float64):
    """[batch, row_len] item-major rows for the given grades"""
    rl = sum(n_choose_k(n, k) for k in sorted(grades))
    return rng.uniform(-1.0, 1.0, (batch, rl)).astype(dtype)


def split_row(n, grades, row):
    out, pos = {}, 0
    for k in sorted(grades):
        g = n_choose_k(n, k)
        out[k] = np.asarray(row[pos:pos + g], dtype=np.float64)
        pos += g
    return out


class OracleBackend:
    name = "oracle"

    def __init__(self):
        self.inputs = {}  # slot -> (GradeMapMV, grades, dim)

    def basis_vectors(self, n):
        return og.Expr.basis_vectors(n)

    def value(self, grades, dim=None):
        return og.mv(og.GradeMapMV({k: np.asarray(v, dtype=np.float64) for k, v in grades.items()}))

    def scalar(self, x):
        return og.Expr._lift(float(x))

    def input(self, slot, grades, dim):
        """per-item input: bound to a mutable GradeMapMV that eval_batch rewrites"""
        if slot not in self.inputs:
            m = og.GradeMapMV({k: np.zeros(n_choose_k(dim, k)) for k in grades})
            self.inputs[slot] = (m, sorted(grades), dim)
        return og.mv(self.inputs[slot][0])


class HipBackend:
    name = "hip"

    def basis_vectors(self, n):
        return ga.Expr.basis_vectors(n)

    def value(self, grades, dim=None):
        return ga.mv(ga.GradeMapMV(grades, dim=dim))

    def scalar(self, x):
        return ga.Expr._lift(float(x))

    def input(self, slot, grades, dim):
        return ga.mv(ga.Input(slot, grades, dim))


def oracle_eval_batch(build, alg, input_rows, batch, mode=og.EVAL_RELEASE):
    """Run the oracle once per item.  input_rows: {slot: [batch,row] or [1,row] (shared)}.
    Returns (rows [batch, out_len] f64, out_mask)."""
    B = OracleBackend()
    spec = build(B).specialize(alg)
    root = spec.nodes()[spec.root()]
    dim = root.vec_space_dim
    out_len = sum(n_choose_k(dim, k) for k in range(64) if (root.minimal >> k) & 1)
    slots = sorted(B.inputs)
    mvs = [B.inputs[s][0] for s in slots]
    data = []
    for s in slots:
        rows = np.asarray(input_rows[s], dtype=np.float64)
        if rows.shape[0] == 1 and batch > 1:
            rows = np.repeat(rows, batch, axis=0)
        data.append(rows)
    out = spec.eval_batch(mvs, data, batch, out_len, mode)
    return out, root.minimal


def hip_eval_batch(build, alg, input_rows, batch, dtype=ga.F64, flags=0, **kw):
    """Same through the C ABI on the GPU.  Returns (rows [batch,out_len], out_mask, spec)."""
    spec = build(HipBackend()).specialize(alg, dtype=dtype, flags=flags, **kw)
    n_slots = spec.num_user_inputs()
    ins = [input_rows.get(s) for s in range(n_slots)]
    out = spec.eval_batch(ins, batch)
    ga.lib().gaast_hip_synchronize()
    return out.download_rows(), out.mask, spec


def assert_same_ast(ospec, hspec):
    """Phases 1-3 of the product agree with the oracle node for node, entry for entry."""
    onodes, hnodes = ospec.nodes(), hspec.nodes()
    assert len(onodes) == len(hnodes)
    assert ospec.root() == hspec.root_id()
    for i, (o, h) in enumerate(zip(onodes, hnodes)):
        assert o.kind == h.opcode, f"node {i}: kind {og.NODE_KINDS[o.kind]} vs {h.opcode}"
        assert (o.child0, o.child1) == (h.child0, h.child1), f"node {i}: children"
        assert o.maximal == h.maximal_grade_mask, f"node {i}: maximal {o.maximal:b} vs {h.maximal_grade_mask:b}"
        assert o.minimal == h.minimal_grade_mask, f"node {i}: minimal {o.minimal:b} vs {h.minimal_grade_mask:b}"
        assert o.vec_space_dim == h.vec_space_dim
        assert o.num_uses == h.num_uses, f"node {i}: num_uses"
        assert o.n_comp_muls == h.n_comp_muls, f"node {i}: {o.n_comp_muls} vs {h.n_comp_muls} comp muls"
        if o.n_comp_muls:
            hm = hspec.comp_muls(i)
            if hm is not None:
                assert ospec.comp_muls(i) == hm, f"node {i}: comp-mul lists differ"


# ----------------------------------------------------------------------------------------------
# Independent second method: the product as a twisted XOR-convolution in blade
# bitmask space, no table, no per-grade indexing.  Used to cross-check the oracle on CPU and as
# the size-independent checker for dimensions where the oracle's 4^n table is too slow to build.
# ----------------------------------------------------------------------------------------------
_POP16 = np.array([bin(i).count("1") for i in range(1 << 16)], dtype=np.int64)


def blades_in_row_order(n, grades):
    """bitmask of every component of a graded row (grades ascending, colex = numeric order)."""
    out = []
    for k in sorted(grades):
        out.extend(m for m in range(1 << n) if _POP16[m] == k)
    return np.array(out, dtype=np.int64)


def row_to_bits(n, grades, row):
    v = np.zeros(1 << n, dtype=np.float64)
    v[blades_in_row_order(n, grades)] = np.asarray(row, dtype=np.float64)
    return v


def bits_to_row(n, grades, bits):
    return bits[blades_in_row_order(n, grades)]


def gp_bits(n, metric_diag, A, B, absolute=False):
    """C[a^b] += s(a,b) * prod(metric over a&b) * A[a] * B[b] over all blade pairs.
    absolute=True returns sum |term| per output blade (for error bounds)."""
    N = 1 << n
    a = np.arange(N, dtype=np.int64)[:, None]
    b = np.arange(N, dtype=np.int64)[None, :]
    par = np.zeros((N, N), dtype=np.int64)
    for s in range(1, n):
        par += _POP16[(a >> s) & b]
    coef = np.where(par & 1, -1.0, 1.0)
    shared = a & b
    for i, g in enumerate(metric_diag):
        if g != 1.0:
            coef = coef * np.where((shared >> i) & 1, float(g), 1.0)
    terms = coef * np.asarray(A, dtype=np.float64)[:, None] * np.asarray(B, dtype=np.float64)[None, :]
    if absolute:
        terms = np.abs(terms)
    return np.bincount((a ^ b).ravel(), weights=terms.ravel(), minlength=N)


def abs_terms_bound(n, A_bits, B_bits, metric=None):
    """sum over all blade pairs landing on each output blade of |A[a]| |B[b]| (the XOR-convolution of |A| and |B|) by a
    Walsh-Hadamard transform: O(n 2^n) instead of the 4^n table of gp_bits(absolute=True).  For metrics with entries
    of modulus <= 1 it bounds sum |term| from above (null vectors only remove terms)."""
    def wht(v):
        v = np.array(v, dtype=np.float64)
        h = 1
        while h < v.size:
            v = v.reshape(-1, 2, h)
            v = np.concatenate([v[:, 0, :] + v[:, 1, :], v[:, 0, :] - v[:, 1, :]], axis=1).reshape(-1)
            h *= 2
        return v
    N = 1 << n
    A_bits, B_bits = np.abs(np.asarray(A_bits, dtype=np.float64)), np.abs(np.asarray(B_bits, dtype=np.float64))
    if metric is None:
        return np.abs(wht(wht(A_bits) * wht(B_bits)) / N)
    # general diagonal metric: |term| = |A_S| |B_U| prod_{i in S & U} |g_i| = (w_S |A_S|) (w_U |B_U|) / w_T with
    # w_S = prod_{i in S} sqrt|g_i| (1 for a null vector, whose terms only vanish: still an upper bound)
    w = np.ones(N)
    for i, g in enumerate(metric):
        if g != 0.0:
            w = np.where((np.arange(N) >> i) & 1, w * np.sqrt(abs(float(g))), w)
    return np.abs(wht(wht(A_bits * w) * wht(B_bits * w)) / N) / w


# ---- C hosts and the test transport (tests/cpp) ---------------------------------------------------------------------
def build_c_host(name, out_dir):
    """gcc a plain C host of the ABI (tests/cpp/<name>.c) against include/gaast_hip.h and libgaast_hip.so"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "gaast_amd", "lib")
    exe = os.path.join(str(out_dir), name)
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                    "-I", os.path.join(root, "tests", "cpp"), os.path.join(root, "tests", "cpp", name + ".c"),
                    "-L", libdir, "-lgaast_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def build_rccl_stub(out_dir):
    """The TEST-ONLY transport with the nccl* entry points the library resolves (tests/cpp/rccl_stub.c): lets several
    ranks share one GPU, which RCCL refuses.  Returns the path to hand to gaast_hip_comm_set_library."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(str(out_dir), "librccl_stub.so")
    subprocess.run(["gcc", "-std=gnu11", "-shared", "-fPIC", "-O1", "-Wall", "-Wextra", "-Werror", "-fvisibility=hidden",
                    "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(root, "tests", "cpp", "rccl_stub.c"),
                    "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-Bsymbolic", "-Wl,-rpath,/opt/rocm/lib", "-o", so], check=True)
    return so
