#!/usr/bin/env python3
"""Headline benchmark: batched evaluation of one SpecializedAst on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload r12|r12s|r12d|r12ds|r12x|r8|r8d|r8s|r8x|cl41|cl41g1|cl41s|gpN{f32,f64}[s|x]]

A "step" is one pass of the hot path (gaast_hip_eval) over one batch of synthetic input
multivectors already resident in HBM.  Default workload = BASELINE.json configs[2], the one the
north-star target is quoted on: full multivector x full multivector geometric product in R^12
(4096 components, 4^12 component multiplies), f32, 65,536 input sets per GPU.  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank evaluates its own shard of
the batch (independent items, no data-path collective): weak scaling.  The RCCL gather of
the result shards to rank 0 is timed separately and reported beside the main number.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel, timed with HIP
events on the launch stream inside the timed region; `cpu_baseline` is the CPU oracle (the C
restatement of the reference's src/eval.rs loop) timed on this host on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
PEAK_FP64_TFLOPS = 78.6     # datasheet (64 cycles per v_mfma_f64_16x16x4_f64 at 2.4 GHz); not in the local guide
MEASURED_FP64_MFMA_TFLOPS = 74.3   # tools/microbench/mfma_f64_rate.hip on this pool (profiles/r03_microbench_mfma_f64_rate.txt)
POWER_CAP_W, POWER_IDLE_W, POWER_W_PER_TFLOPS, POWER_W_PER_TBPS = 1400.0, 250.0, 7.2, 129.0   # measured, round 3 (DESIGN.md 4.1 "A power roofline")
# the other single-GPU BASELINE configurations the default line also times, then (round 4) one workload per kernel family added
# this round: n = 6 on the matrix cores, the projected rotor sandwich (two lists, one launch), the versor inverse and
# d = (a + b * c).g(2) beyond R^3 (whole-AST programs in one launch)
SIDE_CONFIGS = ("r8", "cl41", "r12d", "gp6f32", "sand9g1", "vinv12", "cfg1_8")
PEAK_HBM_GBPS = 8000.0     # spec; 6290 measured float4 copy
GATHER_WATCHDOG_S = 240     # the multi-GPU gather legs give up after this long (the throughput line is printed regardless)
GATHER_FAILED_EXIT = 3      # ... and the job then ends with this status: a hung or failed exchange is a FAILED run


def workload_spec(name):
    """(n, metric, dtype name, builder, input grade lists, label)"""
    import gaast_amd as ga
    if name == "r12":
        n = 12
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F32, dtname="f32", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=65536,
                    label="R^12 full MV x MV geometric product, f32 (BASELINE configs[2])")
    if name == "r12s":
        # same products through the opt-in matrix-representation kernel (GAAST_FLAG_SPINOR_GEMM): not the
        # reference's summation order, so it is a separate workload and never the default line
        n = 12
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F32, dtname="f32", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=65536, flags=ga.FLAG_SPINOR_GEMM,
                    flops_item=2 * 3 * 32 * 64 * 64 + 2 * 384 * (256 + 64),   # executed: HALF of the complex product, mirrored
                    label="R^12 full MV x MV geometric product, f32, opt-in 64x64 complex matrix representation")
    if name == "r66s":
        # Cl(6,6) = M(64, R): the matrix representation is real (lambda = 0), one real 64 x 64 product per item
        n = 12
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0, -1.0] * 6, dtype=ga.F32, dtname="f32", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=65536, flags=ga.FLAG_SPINOR_GEMM,
                    flops_item=2 * 64 ** 3 + 2 * 384 * (256 + 64),
                    label="Cl(6,6) full MV x MV geometric product, f32, opt-in 64x64 REAL matrix representation")
    if name == "r12d":
        n = 12
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=16384,
                    label="R^12 full MV x MV geometric product, f64 (the reference's value type)")
    if name == "r12ds":
        n = 12
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=16384, flags=ga.FLAG_SPINOR_GEMM,
                    flops_item=2 * 3 * 32 * 64 * 64 + 2 * 384 * (128 + 64),   # executed: HALF of the complex product, mirrored
                    label="R^12 full MV x MV geometric product, f64, opt-in 64x64 complex matrix representation")
    if name == "r12x":
        n = 12
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=512, flags=ga.FLAG_EXACT_ORDER,
                    label="R^12 full MV x MV geometric product, f64, reference summation order (bit-exact kernel)")
    if name == "r8d":
        n = 8
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=1 << 19,
                    label="R^8 full MV x MV geometric product, f64 (the reference's value type)")
    if name == "r8x":
        n = 8
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=1 << 16, flags=ga.FLAG_EXACT_ORDER,
                    label="R^8 full MV x MV geometric product, f64, reference summation order (bit-exact kernel)")
    if name == "r8s":
        n = 8
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F32, dtname="f32", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=1 << 20, flags=ga.FLAG_SPINOR_GEMM,
                    flops_item=2 * 3 * 16 ** 3 + 2 * 64 * (64 + 16),
                    label="R^8 full MV x MV geometric product, f32, opt-in 16x16 complex matrix representation")
    if name == "r8":
        n = 8
        full = list(range(n + 1))
        return dict(n=n, metric=[1.0] * n, dtype=ga.F32, dtname="f32", inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=1 << 20,
                    label="R^8 full MV x MV geometric product, f32 (BASELINE configs[1])")
    if name == "cl41":
        return dict(n=5, metric=[1.0, 1.0, 1.0, 1.0, -1.0], dtype=ga.F64, dtname="f64",
                    inputs=[[0, 2, 4], [1]], build=lambda r, x: r * x * r.rev(), entries=80 + 256,
                    default_batch=1 << 22,
                    label="R^{4,1} rotor sandwich R X ~R, f64 (BASELINE configs[4])")
    if name == "cl41g1":    # the same sandwich projected on grade 1 (tables 80 + 80, 5 result components)
        return dict(n=5, metric=[1.0, 1.0, 1.0, 1.0, -1.0], dtype=ga.F64, dtname="f64",
                    inputs=[[0, 2, 4], [1]], build=lambda r, x: (r * x * r.rev()).g(1), entries=80 + 80,
                    default_batch=1 << 22,
                    label="R^{4,1} rotor sandwich (R X ~R).g(1), f64")
    if name == "cl41s":     # one rotor shared by every item (batch-1 input): 40 B in, 128 B out per item
        return dict(n=5, metric=[1.0, 1.0, 1.0, 1.0, -1.0], dtype=ga.F64, dtname="f64",
                    inputs=[[0, 2, 4], [1]], shared=[0], build=lambda r, x: r * x * r.rev(), entries=80 + 256,
                    default_batch=1 << 22,
                    label="R^{4,1} rotor sandwich R X ~R with one shared rotor, f64")
    import re
    m = re.fullmatch(r"sand(\d+)(g1)?(x)?", name)   # sand8 / sand9 / sand10: BASELINE configs[4]'s pipeline where it no longer fits one fused launch
    if m:
        # the rotor sandwich R X ~R (R even, X grade 1; f64) at n = 8 (R^8), 9 (R^{6,3}), 10 (R^10): 80 / 256-entry tables become
        # n 2^(n-1) + 4^(n-1) entries; sandNg1 projects on grade 1, sandNx keeps the reference's summation order
        n = int(m.group(1))
        metric = [1.0] * 6 + [-1.0] * 3 if n == 9 else [1.0] * n
        even = [k for k in range(n + 1) if k % 2 == 0]
        half = 1 << (n - 1)
        g1 = bool(m.group(2))
        return dict(n=n, metric=metric, dtype=ga.F64, dtname="f64", inputs=[even, [1]],
                    build=(lambda r, x: (r * x * r.rev()).g(1)) if g1 else (lambda r, x: r * x * r.rev()),
                    entries=n * half + (n * half if g1 else half * half), default_batch=max(1024, min(1 << 22, (1 << 30) // (half * 8))),
                    flags=ga.FLAG_EXACT_ORDER if m.group(3) else 0,
                    label=f"rotor sandwich {'(R X ~R).g(1)' if g1 else 'R X ~R'} in {'R^{6,3}' if n == 9 else 'R^%d' % n}, R even ({half} components), X grade 1, f64"
                          + (", reference summation order" if m.group(3) else ""))
    # ---- whole-AST programs beyond a bare product or the sandwich (round 4; eval.rs has ten arms) ----
    m = re.fullmatch(r"unary(\d+)", name)   # the element-wise arms on rows too big to fuse: GradedObj, Negation, Reverse, GradeInvolution, Addition, then a scaling product
    if m:
        from math import comb
        n = int(m.group(1))
        k = n // 2
        return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[[k], [k], [0]],
                    build=lambda a, b, s: (-(a.rev()) + b.ginvol()).rev() * s, entries=comb(n, k), default_batch=max(1024, min(1 << 20, (1 << 28) // (comb(n, k) * 8))),
                    label=f"(-(a.rev()) + b.ginvol()).rev() * s in R^{n}: a, b grade {k} ({comb(n, k)} components), s scalar, f64 -- one kernel per eval.rs arm")
    m = re.fullmatch(r"(vinv|proj|cfg1_)(\d+)", name)
    if m:
        from math import comb
        kind, n = m.group(1), int(m.group(2))
        full = list(range(n + 1))
        even = [k for k in full if k % 2 == 0]
        half = 1 << (n - 1)
        if kind == "vinv":      # the versor inverse a.rev() * a.norm_sq().sinv() of an even multivector (expr.rs:363-371): Reverse, Product -> grade 0, ScalarUnaryOp, Product
            return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[even], build=lambda a: a.vinv(), entries=2 * half,
                        default_batch=max(1024, min(1 << 22, (1 << 30) // (half * 8))),
                        label=f"versor inverse a.rev() * a.norm_sq().sinv() in R^{n}, a even ({half} components), f64")
        if kind == "proj":      # the projection known-answer test of eval.rs:152-163 on batched inputs: (v & bv) & bv.vinv(), v a vector, bv a bivector
            return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[[1], [2]], build=lambda v, bv: (v & bv) & bv.vinv(),
                        entries=3 * n * (n - 1), default_batch=1 << 22,
                        label=f"projection (v & bv) & bv.vinv() in R^{n} (eval.rs:152-163 at scale), f64")
        # README.md:20-22 / BASELINE configs[0] at scale: d = (a + b * c).g(2), a, b, c full multivectors; of a only grade 2 is read
        return dict(n=n, metric=[1.0] * n, dtype=ga.F64, dtname="f64", inputs=[full, full, full], build=lambda a, b, c: (a + b * c).g(2),
                    entries=comb(n, 2) << n, default_batch=max(1024, min(1 << 20, (1 << 29) // ((1 << n) * 8))),
                    read_len=comb(n, 2) + 2 * (1 << n),
                    label=f"d = (a + b * c).g(2) in R^{n}, full operands (README.md:20-22), f64")
    m = re.fullmatch(r"gp(\d+)(f32|f64)(deg|gen)", name)   # the cold instantiations of the dense kernels: a null vector / a general diagonal metric
    if m:
        n, dt, var = int(m.group(1)), m.group(2), m.group(3)
        full = list(range(n + 1))
        sz = 4 if dt == "f32" else 8
        metric = [0.0] + [1.0] * (n - 2) + [-1.0] if var == "deg" else [2.0 ** ((i % 5) - 2) * (-1.0 if i % 3 == 1 else 1.0) for i in range(n)]
        return dict(n=n, metric=metric, dtype=ga.F32 if dt == "f32" else ga.F64, dtname=dt, inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=max(64, min(1 << 22, (1 << 30) // ((1 << n) * sz))),
                    label=f"R^{n} full MV x MV geometric product, {dt}, " + ("one null vector (PGA style), one -1" if var == "deg" else "general diagonal metric (rescaled basis)"))
    m = re.fullmatch(r"gp(\d+)(f32|f64)(s|x|ee|eo|oe|oo)?", name)   # e.g. gp10f32, gp10f32s (matrix representation), gp9f64x (exact order), gp12f32ee (even x even)
    if m:
        n, dt, var = int(m.group(1)), m.group(2), m.group(3)
        full = list(range(n + 1))
        sz = 4 if dt == "f32" else 8
        if var in ("ee", "eo", "oe", "oo"):
            # parity-pure operands (rotor composition, the second product of a sandwich): 2^(n-1) components each, a quarter of
            # the 4^n table -- one product in the even subalgebra Cl(n - 1)
            grades = {"e": [k for k in full if k % 2 == 0], "o": [k for k in full if k % 2 == 1]}
            batch = max(64, min(1 << 22, (1 << 30) // ((1 << (n - 1)) * sz)))
            names = {"e": "even", "o": "odd"}
            return dict(n=n, metric=[1.0] * n, dtype=ga.F32 if dt == "f32" else ga.F64, dtname=dt, inputs=[grades[var[0]], grades[var[1]]],
                        build=lambda a, b: a * b, entries=4 ** n // 4, default_batch=batch,
                        label=f"R^{n} {names[var[0]]} x {names[var[1]]} geometric product ({1 << (n - 1)} components each, 4^{n}/4 table entries), {dt}")
        batch = max(64, min(1 << 22, (1 << 30) // ((1 << n) * sz)))          # 1 GiB per operand
        if var == "x":
            batch = max(64, batch >> 6)
        flags = ga.FLAG_SPINOR_GEMM if var == "s" else ga.FLAG_EXACT_ORDER if var == "x" else 0
        return dict(n=n, metric=[1.0] * n, dtype=ga.F32 if dt == "f32" else ga.F64, dtname=dt, inputs=[full, full],
                    build=lambda a, b: a * b, entries=4 ** n, default_batch=batch, flags=flags,
                    label=f"R^{n} full MV x MV geometric product, {dt}" +
                          (", opt-in matrix representation" if var == "s" else ", reference summation order" if var == "x" else ""))
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(wl, budget_s=12.0):
    """Time the oracle (literal eval.rs loop, 56-byte AoS table, 1 thread) on a bounded sample."""
    import numpy as np
    from oracle import pyoracle as og
    n = wl["n"]
    rng = np.random.default_rng(0)
    L = og.lib()
    note = ""
    grades_a = wl["inputs"][0]
    scale = 1.0
    if wl["entries"] > (1 << 22):
        # The full table is 4^n x 56 B (940 MB at n = 12).  The reference orders it by left grade
        # first (specialize.rs:162-183), so restricting the LEFT operand to grades 0..4 makes the
        # oracle build exactly the leading slice of that table; time scales with entries.
        grades_a = [k for k in grades_a if k <= 4]
        sub = sum(L.og_n_choose_k(n, k) for k in grades_a) * (1 << n)
        scale = wl["entries"] / sub
        note = f"leading {sub} of {wl['entries']} table entries (left grades 0..4), time scaled x{scale:.3f}; "
    def val(grades):
        return og.GradeMapMV({k: rng.uniform(-1, 1, L.og_n_choose_k(n, k)) for k in grades})
    a, b = val(grades_a), val(wl["inputs"][1])
    t0 = time.time()
    spec = wl["build"](og.mv(a), og.mv(b)).specialize(og.as_algebra(wl["metric"] if any(m != 1.0 for m in wl["metric"]) else n))
    t_spec = time.time() - t0
    t0 = time.time()
    spec.eval()
    t1 = time.time() - t0
    items = max(2, min(1 << 20, int(budget_s / max(t1, 1e-6))))
    t0 = time.time()
    for _ in range(items):
        spec.eval()
    dt = (time.time() - t0) / items
    per_item = dt * scale
    # all host cores: one forked worker per core, each evaluating the same specialized AST on its own
    # copy of the inputs (the reference is single-threaded and !Send, so this is its embarrassingly
    # parallel ceiling: independent evaluations in independent processes)
    cores = len(os.sched_getaffinity(0))
    try:  # a container's CPU share (cgroup v2) can be far below the affinity mask
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    all_cores = None
    if cores > 1:
        per_worker = max(2, int(0.5 * budget_s / dt))      # about half the single-core leg's wall time
        pipes, pids = [], []
        t0 = time.time()
        for _ in range(cores):
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:
                os.close(r)
                code = 1
                try:
                    for _ in range(per_worker):
                        spec.eval()
                    os.write(w, b"k")
                    code = 0
                finally:
                    os._exit(code)
            os.close(w)
            pipes.append(r)
            pids.append(pid)
        ok = True
        for r, pid in zip(pipes, pids):
            ok = (os.read(r, 1) == b"k") and ok
            os.close(r)
            os.waitpid(pid, 0)
        wall = time.time() - t0
        if ok:
            all_cores = {"value": cores * per_worker / (wall * scale), "unit": "products/s", "cores": cores,
                         "sample": f"{cores} forked workers x {per_worker} evaluations, {wall:.1f} s wall"}
    # `cpu-packed-1t` (the same loop over 16-byte packed entries on flat rows: how much of the reference's
    # time is table traffic) and `cpu-packed-allcores` (the batch split over threads: the fair CPU ceiling)
    packed = None
    pk = spec.packed_root_product()
    if pk is not None:
        hnd, n_ent, pll, prl, pol = pk
        t_item = max(1e-7, n_ent * 1.2e-9)
        pb1 = max(2, min(1 << 16, int(0.25 * budget_s / t_item)))
        dbl = C.POINTER(C.c_double)
        def run(pbatch, threads):
            la = np.ascontiguousarray(rng.uniform(-1, 1, (pbatch, pll)))
            ra = np.ascontiguousarray(rng.uniform(-1, 1, (pbatch, prl)))
            oa = np.empty((pbatch, pol))
            sec = L.og_packed_eval_batch(hnd, n_ent, la.ctypes.data_as(dbl), pll, ra.ctypes.data_as(dbl), prl,
                                         oa.ctypes.data_as(dbl), pol, pbatch, threads)
            return pbatch / (sec * scale)
        v1 = run(pb1, 1)
        packed = {"cpu-packed-1t": {"value": v1, "unit": "products/s", "cores": 1,
                                    "sample": note + f"{pb1} items, {n_ent} entries of 16 bytes (u32 left, u32 right, u32 out, f32 coeff) on flat rows, same order and roundings"}}
        if cores > 1:
            pbn = pb1 * cores
            packed["cpu-packed-allcores"] = {"value": run(pbn, cores), "unit": "products/s", "cores": cores,
                                             "sample": note + f"{pbn} items over {cores} threads (contiguous item ranges)"}
        L.og_packed_free(hnd)
    out = {"value": 1.0 / per_item, "unit": "products/s", "cores": 1, "kind": "port",
           "sample": note + f"{items} evaluations of the oracle's eval.rs loop (f64, 56-byte AoS entries, "
                            f"per-eval allocations included), {dt * 1e3:.2f} ms each; table build {t_spec:.1f} s excluded; "
                            f"host has {cores} cores"}
    if all_cores:
        out["all_cores"] = all_cores
    if packed:
        out["variants"] = packed      # cpu-ref-1t is `value`; these are the packed-table variants
    return out


def _load_traffic_table():
    """profiles/traffic.json: HBM bytes per launch measured with rocprofv3 --pmc (tools/pmc_traffic.py writes it from
    the raw counter CSVs): {"<workload>:<batch>": {"kernel": ..., "bytes": ..., "source": ...}}"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def _traffic_for(workload, batch, kernel_names):
    """(bytes per launch or None, source or None, stale flag).  A counter recorded for another kernel than the one this
    run launched is NOT reported: traffic = None, traffic_stale = True."""
    import gaast_amd as ga
    ent = _load_traffic_table().get(f"{workload}:{batch}")
    if not ent:
        return None, None, False
    if ent["kernel"] not in kernel_names or ent.get("library") != ga.lib().gaast_hip_version().decode():
        return None, ent.get("source"), True
    return ent["bytes"], ent.get("source"), False


def roofline_of(wl, workload, spec, batch, kernel_ms, in_len, out_len):
    """The `roofline` object of one workload: ALGORITHMIC bytes / flops per launch (every input / output component
    touched once; one multiply + one add per comp-mul entry) over the kernel's average launch duration."""
    import gaast_amd as ga
    dtype = wl["dtype"]
    sz = 4 if dtype == ga.F32 else 8
    bytes_item = (in_len + out_len) * sz
    flops_item = wl.get("flops_item", 2 * wl["entries"])
    launches = spec.launches()
    dense = any("product_dense" in l or "product_spinor" in l for l in launches)
    peak_tf = PEAK_FP32_TFLOPS if dtype == ga.F32 else PEAK_FP64_TFLOPS
    ach_tf = flops_item * batch / (kernel_ms * 1e-3) * 1e-12
    ach_gb = bytes_item * batch / (kernel_ms * 1e-3) * 1e-9
    # the BINDING roof is the one that allows fewer items per second: dense products are matrix / vector FMA bound from
    # n = 7 on; at n = 6 (768 B and 8,192 flop per item in f32) the HBM roof binds
    hbm_bound_dense = PEAK_HBM_GBPS * 1e9 / bytes_item < peak_tf * 1e12 / flops_item
    if dense and not hbm_bound_dense:
        roof = {"bound": "mfma", "achieved": ach_tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach_tf / peak_tf,
                "traffic": None,
                "note": ("dense product: fp32 vector FMA peak == fp32 MFMA peak (157.3 TFLOP/s); " if dtype == ga.F32 else
                         "dense product: fp64 vector FMA peak == fp64 MFMA peak (78.6 TFLOP/s datasheet; 74-77 measured: "
                         "profiles/r03_microbench_mfma_f64_rate.txt); ") +
                        f"algorithmic HBM {ach_gb:.1f} GB/s = {ach_gb / PEAK_HBM_GBPS:.4f} of 8 TB/s"}
        if dtype != ga.F32:
            roof["frac_of_measured_peak"] = ach_tf / MEASURED_FP64_MFMA_TFLOPS
        else:
            # informational: the FP32 rate 1,400 W buy at this kernel's HBM bytes per flop (DESIGN.md 4.1 "A power roofline":
            # 250 W + 7.2 W per TFLOP/s + 129 W per TB/s, from rocm-smi readings under load, profiles/r03_clocks_and_power_under_load.txt)
            cap_tf = (POWER_CAP_W - POWER_IDLE_W) / (POWER_W_PER_TFLOPS + POWER_W_PER_TBPS * (bytes_item / flops_item))
            roof["power_capped_frac_ceiling"] = min(1.0, cap_tf / peak_tf)
    else:
        roof = {"bound": "hbm", "achieved": ach_gb, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": ach_gb / PEAK_HBM_GBPS,
                "traffic": None, "note": f"{len(launches)} launches per evaluation; {ach_tf:.2f} TFLOP/s = {ach_tf / peak_tf:.4f} of the {'FP32' if dtype == ga.F32 else 'FP64'} roof"}
    roof["kernel"] = [l for l in launches if "product" in l][-1] if any("product" in l for l in launches) else launches[-1]
    traffic, source, stale = _traffic_for(workload, batch, [roof["kernel"]])
    roof["traffic"] = traffic
    roof["traffic_source"] = source
    roof["traffic_stale"] = stale
    roof["algorithmic_bytes_per_launch"] = bytes_item * batch
    roof["kernel_ms"] = kernel_ms
    roof["flops_per_item"] = flops_item
    roof["bytes_per_item"] = bytes_item
    return roof


def side_config(name, dev, stream, steps, warmup):
    """One of the other single-GPU BASELINE configurations, measured in the same process after the headline: the same
    timing discipline (inputs resident, HIP events on the launch stream around every step), a few steps."""
    import torch
    import gaast_amd as ga
    wl = workload_spec(name)
    n, dtype, batch = wl["n"], wl["dtype"], wl["default_batch"]
    tdt = torch.float32 if dtype == ga.F32 else torch.float64
    exprs = [ga.mv(ga.Input(s, g, n)) for s, g in enumerate(wl["inputs"])]
    t0 = time.time()
    spec = wl["build"](*exprs).specialize(ga.MetricAlgebra(wl["metric"]), dtype=dtype, flags=wl.get("flags", 0))
    spec.program()
    t_spec = time.time() - t0
    out_mask, out_len = spec.output_info()
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    ins, in_t = [], []
    for slot, g in enumerate(wl["inputs"]):
        rl = ga.graded.row_len(n, ga.graded._mask_of(g))
        nb = 1 if slot in wl.get("shared", []) else batch
        t = torch.empty((nb, rl), device=dev, dtype=tdt)
        for lo in range(0, nb, 1 << 16):
            t[lo:lo + (1 << 16)].uniform_(-1, 1, generator=gen)
        in_t.append(t)
        ins.append(ga.DeviceMV.wrap_tensor(t, n, g))
    out_t = torch.empty((batch, out_len), device=dev, dtype=tdt)
    out = ga.DeviceMV.wrap_tensor(out_t, n, ga.GradeSet(out_mask))
    for _ in range(warmup):
        spec.eval_batch(ins, batch, out=out)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)
        spec.eval_batch(ins, batch, out=out)
        e1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / steps
    in_len = wl.get("read_len", sum(t.shape[1] for t in in_t if t.shape[0] == batch))
    res = {"workload": wl["label"], "key": name, "dim": n, "dtype": wl["dtname"], "batch": batch, "steps": steps, "warmup": warmup,
           "value": batch * steps / wall, "unit": "products/s" if len(wl["inputs"]) == 2 and wl["entries"] == 4 ** n else "evaluations/s",
           "ms_per_step": wall / steps * 1e3, "launches_per_eval": spec.launches(), "specialize_s": t_spec,
           "roofline": roofline_of(wl, name, spec, batch, kernel_ms, in_len, out_len)}
    del ins, in_t, out, out_t, spec
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="r12")
    ap.add_argument("--batch", type=int, default=0, help="input sets per GPU (default: the workload's; with --gpus N > 1 "
                    "and the default workload: BASELINE configs[3], 1,048,576 input sets over all GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather-chunks", type=int, default=4)
    ap.add_argument("--no-alt", action="store_true", help="skip the opt-in matrix-representation side measurement")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-item latency side measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the other single-GPU BASELINE configurations (r8, cl41, r12d) "
                    "the default line also times")
    ap.add_argument("--flags", type=lambda x: int(x, 0), default=0, help="extra GAAST_FLAG_* bits for the program (A/B measurements)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher: start the N ranks ourselves, BEFORE anything touches a GPU
    # (this parent never imports torch; it waits for the ranks and exits with their status)
    if args.gpus > 1 and "RANK" not in os.environ:
        from gaast_amd.launch import self_launch
        raise SystemExit(self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    # stdout carries exactly ONE JSON line: whatever the libraries underneath print there (gloo / RCCL banners) goes
    # to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # CPU baseline first (rank 0, N = 1 only): its all-cores leg forks workers, which must happen before
    # this process initialises the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(workload_spec(args.workload))

    import torch
    import torch.distributed as dist
    import gaast_amd as ga
    from gaast_amd.sharding import max_over_ranks, shard_range

    # GAAST_BENCH_REHEARSAL=1: rehearse the N > 1 control flow on a box with fewer GPUs than ranks (ranks
    # share devices, collectives go through gloo on host copies).  Never set by the driver; numbers
    # from such a run are not measurements.
    rehearsal = world > 1 and os.environ.get("GAAST_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = None
    lib_comm = False
    lib_comm_error = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ga.init_device(local_rank)
    L = ga.lib()
    stream = torch.cuda.current_stream()
    ga._lib.check(L.gaast_hip_set_stream(C.c_void_p(stream.cuda_stream)))
    if world > 1:
        # how many ranks the collective library really joined (an all-reduce of ones)
        ones = torch.ones(1, dtype=torch.int64, device="cpu" if rehearsal else dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        # rehearsal only: the test transport (tests/cpp/rccl_stub.c, built by the test) stands in for RCCL, which refuses
        # ranks that share a device -- so that the LIBRARY's gather runs, not a torch.distributed stand-in
        transport = os.environ.get("GAAST_BENCH_REHEARSAL_TRANSPORT") if rehearsal else None
        if transport:
            ga._lib.check(L.gaast_hip_comm_set_library(transport.encode()))
        if not rehearsal or transport:
            # the library's own communicator (what a Rust host would use: include/gaast_hip.h, multi-GPU): the
            # 128-byte id travels over the rendezvous torch.distributed already provides
            # (a failure here is reported in the JSON and the gather falls back to torch.distributed: the throughput
            # line must not depend on it)
            cdev = "cpu" if rehearsal else dev
            idbuf = (C.c_ubyte * ga._lib.COMM_ID_BYTES)()
            ok = 1
            if rank == 0:
                try:
                    ga._lib.check(L.gaast_hip_comm_unique_id(idbuf))
                except ga.GaastError as e:
                    ok, lib_comm_error = 0, str(e)
            idt = torch.tensor(list(bytes(idbuf)) + [ok], dtype=torch.uint8, device=cdev)
            dist.broadcast(idt, src=0)
            vals = idt.cpu().tolist()
            if vals[-1]:
                idbuf = (C.c_ubyte * ga._lib.COMM_ID_BYTES)(*vals[:-1])
                try:
                    ga._lib.check(L.gaast_hip_comm_init(idbuf, rank, world))
                    nr = C.c_int()
                    ga._lib.check(L.gaast_hip_comm_count_ranks(C.byref(nr)))
                    joined = int(nr.value == world)
                except ga.GaastError as e:
                    joined, lib_comm_error = 0, str(e)
            else:
                joined = 0
            # every rank must agree on which path the gather takes
            agree = torch.tensor([joined], dtype=torch.int64, device=cdev)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            lib_comm = bool(agree.item())
            if not lib_comm and lib_comm_error is None:
                lib_comm_error = "a rank could not join the library communicator"

    wl = workload_spec(args.workload)
    n, dtype = wl["n"], wl["dtype"]
    label = wl["label"]
    scaling = "weak"
    if args.batch:
        batch = args.batch
        global_batch = batch * world
        first = rank * batch
    elif world > 1 and args.workload == "r12":
        # BASELINE configs[3]: 1,048,576 input sets, contiguous shards of ceil(B / g) items per GPU
        global_batch = 1 << 20
        first, stop = shard_range(global_batch, rank, world)
        batch = stop - first
        label = f"R^12 full MV x MV geometric product, f32, 1,048,576 input sets sharded over {world} GPUs (BASELINE configs[3])"
        scaling = "strong"
    else:
        batch = wl["default_batch"]
        global_batch = batch * world
        first = rank * batch
    counts = [batch] * world
    if world > 1:
        ct = torch.tensor([batch], dtype=torch.int64, device="cpu" if rehearsal else dev)
        cl = [torch.zeros_like(ct) for _ in range(world)]
        dist.all_gather(cl, ct)
        counts = [int(c.item()) for c in cl]
        global_batch = sum(counts)
    tdt = torch.float32 if dtype == ga.F32 else torch.float64

    # one SpecializedAst (phases 1-3 on the host, once), one device program
    exprs = [ga.mv(ga.Input(s, g, n)) for s, g in enumerate(wl["inputs"])]
    t0 = time.time()
    spec = wl["build"](*exprs).specialize(ga.MetricAlgebra(wl["metric"]), dtype=dtype, flags=wl.get("flags", 0) | args.flags)
    spec.program()
    t_spec = time.time() - t0
    out_mask, out_len = spec.output_info()

    # synthetic inputs, generated on the device: the shard of rank r is seeded by (seed, r)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3 + rank)
    ins, in_t = [], []
    for slot, g in enumerate(wl["inputs"]):
        rl = ga.graded.row_len(n, ga.graded._mask_of(g))
        nb = 1 if slot in wl.get("shared", []) else batch          # a batch-1 input is shared by every item
        t = torch.empty((nb, rl), device=dev, dtype=tdt)
        for lo in range(0, nb, 1 << 16):                           # in slices: no multi-GiB temporaries
            t[lo:lo + (1 << 16)].uniform_(-1, 1, generator=gen)
        in_t.append(t)
        ins.append(ga.DeviceMV.wrap_tensor(t, n, g))
    # the root's result rows live inside the gathered buffer (no local copy at gather time)
    gathered_t = gathered = None
    want_gather = world > 1 and not args.no_gather
    if want_gather and rank == 0 and lib_comm:
        gathered_t = torch.empty((global_batch, out_len), device=dev, dtype=tdt)
        gathered = ga.DeviceMV.wrap_tensor(gathered_t, n, ga.GradeSet(out_mask))
        out_t = gathered_t[:batch]
    else:
        out_t = torch.empty((batch, out_len), device=dev, dtype=tdt)
    out = ga.DeviceMV.wrap_tensor(out_t, n, ga.GradeSet(out_mask))

    def step():
        spec.eval_batch(ins, batch, out=out)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        """wall seconds of `steps` calls between fences, max over ranks"""
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        if lib_comm:
            ga._lib.check(L.gaast_hip_synchronize())
        fence()
        w = time.perf_counter() - t0
        return max_over_ranks(w, device=None if rehearsal else dev) if world > 1 else w

    for _ in range(args.warmup):
        step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)
        step()
        e1.record(stream)
    fence()
    wall = time.perf_counter() - t0
    if world > 1:
        wall = max_over_ranks(wall, device=None if rehearsal else dev)
    step_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    kernel_ms = sum(step_ms) / len(step_ms)

    # the same products through the opt-in matrix-representation kernel (not the reference's summation
    # order, so never `value`): reported beside the headline, with its distance from the default path
    alt = None
    if args.workload in ("r12", "r12d", "r8", "r8d") and rank == 0 and world == 1 and not args.no_alt:
        spec_alt = wl["build"](*exprs).specialize(ga.MetricAlgebra(wl["metric"]), dtype=dtype, flags=ga.FLAG_SPINOR_GEMM)
        out_alt_t = torch.empty_like(out_t)
        out_alt = ga.DeviceMV.wrap_tensor(out_alt_t, n, ga.GradeSet(out_mask))
        for _ in range(2):
            spec_alt.eval_batch(ins, batch, out=out_alt)
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(stream)
        for _ in range(args.steps):
            spec_alt.eval_batch(ins, batch, out=out_alt)
        a1.record(stream)
        torch.cuda.synchronize()
        alt_ms = a0.elapsed_time(a1) / args.steps
        sl = slice(0, min(batch, 4096))
        diff = (out_alt_t[sl].double() - out_t[sl].double()).abs().max().item()
        scale = (in_t[0][sl].double().norm(dim=1) * in_t[1][sl].double().norm(dim=1)).max().item()
        # multiply-adds the kernel executes: three real D x D products (Gauss); from D = 32 on only the half of the rows the
        # representation's real structure does not determine (Euclidean signatures: lambda != 0)
        D_alt = 1 << (n // 2)
        alt_flops = 3 * 2 * D_alt ** 3 // (2 if D_alt >= 32 else 1)
        sz_a = 4 if dtype == ga.F32 else 8
        alt = {"kernel": [l for l in spec_alt.launches() if "product" in l][-1], "value": batch / (alt_ms * 1e-3),
               "unit": "products/s", "kernel_ms": alt_ms, "algorithmic_GBps": batch * 3 * (1 << n) * sz_a / (alt_ms * 1e-3) * 1e-9,
               "mfma_TFLOPs": batch * alt_flops / (alt_ms * 1e-3) * 1e-12,
               "frac_of_hbm_roof": batch * 3 * (1 << n) * sz_a / (alt_ms * 1e-3) * 1e-9 / PEAK_HBM_GBPS,
               "frac_of_mfma_roof": batch * alt_flops / (alt_ms * 1e-3) * 1e-12 / (PEAK_FP32_TFLOPS if dtype == ga.F32 else PEAK_FP64_TFLOPS),
               "max_abs_diff_vs_default_path": diff, "diff_over_eps_normA_normB": diff / ((2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52) * scale),
               "note": f"opt-in GAAST_FLAG_SPINOR_GEMM: {1 << (n // 2)}x{1 << (n // 2)} complex matrix representation, 3 real MFMA products per item"
                       f"{' (half of each: the rest is its mirror image)' if D_alt >= 32 else ''}; mfma_TFLOPs counts what is executed; "
                       "norm-wise error bound, not the reference's summation order"}
        del out_alt_t, out_alt, spec_alt

    # what the reference actually does -- ONE input set per eval(): latency of a batch-1 evaluation through the
    # C ABI (program built once; launch + kernel + synchronize), beside the batched throughput
    latency = None
    if rank == 0 and world == 1 and not args.no_latency:
        one_in = [ga.DeviceMV.wrap_tensor(t[:1], n, g) for t, g in zip(in_t, wl["inputs"])]
        one_out_t = torch.empty((1, out_len), device=dev, dtype=tdt)
        one_out = ga.DeviceMV.wrap_tensor(one_out_t, n, ga.GradeSet(out_mask))
        for _ in range(3):
            spec.eval_batch(one_in, 1, out=one_out)
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            spec.eval_batch(one_in, 1, out=one_out)
            torch.cuda.synchronize()
        latency = {"batch": 1, "ms_per_eval": (time.perf_counter() - t0) / reps * 1e3, "program_create_s": t_spec,
                   "note": "one gaast_hip_eval of one input set + synchronize (the reference evaluates one input set per "
                           "eval()); the program is built once per SpecializedAst"}

    # the other single-GPU BASELINE configurations (configs[1] R^8 f32, configs[4] R^{4,1} sandwich f64, and R^12 in the
    # reference's value type), each with its own roofline object: a few steps each, after the headline's buffers are idle
    side = []
    if args.workload == "r12" and rank == 0 and world == 1 and not args.no_configs and not args.batch and not args.flags:
        for name in SIDE_CONFIGS:
            try:
                side.append(side_config(name, dev, stream, steps=20, warmup=5))
            except Exception as e:      # never at the expense of the headline line
                side.append({"key": name, "error": f"{type(e).__name__}: {e}"})

    def emit(gather):
        """rank 0: the ONE JSON line"""
        items_total = global_batch * args.steps
        value = items_total / wall
        in_len = sum(t.shape[1] for t in in_t if t.shape[0] == batch)      # shared inputs are read once, not per item
        in_len = wl.get("read_len", in_len)                                # (a + b * c).g(2): of `a` only the wanted grade is read
        launches = spec.launches()
        roof = roofline_of(wl, args.workload, spec, batch, kernel_ms, in_len, out_len)
        res = {
            "metric": "full-MV geometric products/sec (dim n); achieved HBM GB/s vs roofline",
            "value": value, "unit": "products/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": wl["dtname"], "data": "synthetic",
            "global_batch": global_batch,
            **({"rccl_ranks": rccl_ranks} if world > 1 else {}),
            **({"rehearsal": "ranks share GPUs, gloo / test-transport collectives: control-flow check only, not a measurement"} if rehearsal else {}),
            "config": {"workload": label, "dim": n, "batch_per_gpu": batch, "global_batch": global_batch,
                       "shards": counts, "launches_per_eval": launches, "specialize_s": t_spec},
            "roofline": roof,
        }
        if side:
            res["configs"] = side
        if alt is not None:
            res["matrix_representation"] = alt
        if latency is not None:
            res["single_item_latency"] = latency
        if gather is not None:
            res["gather"] = gather
        if cpu is not None:
            res["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    # the one exchange of the path: result rows to rank 0.  Measured twice over the same K steps:
    #   blocking   eval, then gaast_hip_gather_rows (one transfer per peer, all links at once)
    #   overlapped gaast_hip_eval_gather: the shard in `chunks` chunks, chunk k travels while chunk k + 1 computes
    gather = None
    if want_gather:
        sz = 4 if dtype == ga.F32 else 8
        chunks = max(1, args.gather_chunks)
        cnt = (C.c_int64 * world)(*counts)
        if lib_comm:
            def blocking():
                spec.eval_batch(ins, batch, out=out)
                ga._lib.check(L.gaast_hip_gather_rows(out._h, gathered._h if gathered is not None else None, cnt, 0))

            def overlapped():
                spec.eval_gather(ins, out, gathered, counts, root=0, n_chunks=chunks)
            path = ("gaast_hip_eval_gather / gaast_hip_gather_rows: RCCL send/recv, one direct transfer per peer" if not rehearsal else
                    "gaast_hip_eval_gather / gaast_hip_gather_rows over the TEST transport (tests/cpp/rccl_stub.c: ranks share a GPU)")
        else:
            # rehearsal (ranks share a GPU, RCCL refuses that: gloo on host copies) or no library communicator
            # (torch.distributed's own RCCL on device tensors): the same chunk schedule with torch collectives
            from gaast_amd.sharding import chunk_span
            per = max(counts)
            host = rehearsal
            def _gather_span(lo, hi, rows, async_op):
                send = torch.zeros((rows, out_len), dtype=tdt, device="cpu" if host else dev)
                send[:hi - lo] = out_t[lo:hi].cpu() if host else out_t[lo:hi]
                bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
                return dist.gather(send, bufs, dst=0, async_op=async_op), bufs

            def blocking():
                spec.eval_batch(ins, batch, out=out)
                torch.cuda.synchronize()
                _gather_span(0, batch, per, False)

            def overlapped():
                pend = []
                for c in range(chunks):
                    lo, hi = chunk_span(batch, chunks, c)
                    sub_in = [ga.DeviceMV.wrap_tensor(t[lo:hi] if t.shape[0] == batch else t, n, g) for t, g in zip(in_t, wl["inputs"])]
                    sub_out = ga.DeviceMV.wrap_tensor(out_t[lo:hi], n, ga.GradeSet(out_mask))
                    if hi > lo:
                        spec.eval_batch(sub_in, hi - lo, out=sub_out)
                    torch.cuda.synchronize()
                    pend.append(_gather_span(lo, hi, -(-per // chunks), True))
                for h, _ in pend:
                    h.wait()
            path = ("torch.distributed gather of host copies (rehearsal: ranks share a GPU)" if rehearsal else
                    "torch.distributed gather on device tensors (library communicator unavailable: " + str(lib_comm_error) + ")")
        # If the gather legs hang or fail, the throughput line measured above is still printed (gather.error says what
        # happened) -- and the job ends with status GATHER_FAILED_EXIT, so that the launcher and the driver see a failure.
        import threading
        state = {"done": False}

        def give_up():
            if state["done"]:
                return
            if rank == 0:
                emit({"path": path, "library_communicator": lib_comm, "chunks": chunks,
                      "error": f"gather legs did not finish within {GATHER_WATCHDOG_S} s; throughput above is unaffected"})
            os._exit(GATHER_FAILED_EXIT)
        watchdog = threading.Timer(GATHER_WATCHDOG_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            if rehearsal and os.environ.get("GAAST_BENCH_FAIL_GATHER") == "1":     # test hook: the failure path
                raise RuntimeError("injected gather failure (GAAST_BENCH_FAIL_GATHER)")
            blocking()
            overlapped()
            w_block = timed(blocking, args.steps)
            w_over = timed(overlapped, args.steps)
            items_total = global_batch * args.steps
            gather = {"path": path, "library_communicator": lib_comm, "chunks": chunks, "bytes_per_rank": out_len * sz * batch,
                      "bytes_total": out_len * sz * global_batch,
                      "ms_per_step_eval_only": wall / args.steps * 1e3,
                      "ms_per_step_blocking_gather": w_block / args.steps * 1e3,
                      "ms_per_step_overlapped_gather": w_over / args.steps * 1e3,
                      "value_without_gather": items_total / wall,
                      "value_with_blocking_gather": items_total / w_block,
                      "value_with_gather": items_total / w_over,
                      "ms": max(0.0, (w_block - wall) / args.steps * 1e3)}
            if rank == 0 and gathered_t is not None:
                gather["gathered_rows"] = int(gathered_t.shape[0])
                # rows of the last shard, as the root received them, against what a second, blocking gather delivers
                gather["last_shard_checksum"] = float(gathered_t[-counts[-1]:].double().sum().item())
        except Exception as e:      # a failed collective on this rank: say so, keep the throughput line
            gather = {"path": path, "library_communicator": lib_comm, "chunks": chunks, "error": f"{type(e).__name__}: {e}"}
        state["done"] = True
        watchdog.cancel()

    if rank == 0:
        emit(gather)
    if gather is not None and "error" in gather:
        sys.stderr.write(f"rank {rank}: gather failed: {gather['error']}\n")
        sys.stderr.flush()
        os._exit(GATHER_FAILED_EXIT)          # some rank may be stuck in a collective: no further rendezvous
    if world > 1:
        dist.barrier()
        if lib_comm:
            L.gaast_hip_comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
