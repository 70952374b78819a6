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
PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBPS = 8000.0     # spec; 6290 measured float4 copy

# HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x 2 x 1024 +
# WRITE_SIZE x 1024, the gfx950 correction of MI355X_MICROARCH.md): (workload, batch) -> bytes
PMC_TRAFFIC = {("r12s", 65536): (2 * 1.04883e6 * 1024 + 1.04858e6 * 1024, "profiles/r01_r12s_pmc_counters.csv"),
               ("r12", 65536): (2 * 1.04889e6 * 1024 + 1.04858e6 * 1024, "profiles/r01_r12_mfma_pmc_counters.csv"),
               ("r8", 1 << 20): (2 * 1.04869e6 * 1024 + 1.04858e6 * 1024, "profiles/r01_r8_pmc_counters.csv"),
               ("cl41g1", 1 << 22): (2 * 344100 * 1024 + 163853 * 1024, "profiles/r01_cl41g1_pmc_counters.csv"),
               ("cl41", 1 << 22): (2 * 344497.1 * 1024 + 524352.1 * 1024, "profiles/r01_cl41_pmc_counters.csv")}


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
                    flops_item=2 * 3 * 64 ** 3 + 2 * 384 * (256 + 64),
                    label="R^12 full MV x MV geometric product, f32, opt-in 64x64 complex matrix representation")
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
                    flops_item=2 * 3 * 64 ** 3 + 2 * 384 * (128 + 64),
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
    m = re.fullmatch(r"gp(\d+)(f32|f64)(s|x)?", name)   # e.g. gp10f32, gp10f32s (matrix representation), gp9f64x (exact order)
    if m:
        n, dt, var = int(m.group(1)), m.group(2), m.group(3)
        full = list(range(n + 1))
        sz = 4 if dt == "f32" else 8
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
    out = {"value": 1.0 / per_item, "unit": "products/s", "cores": 1, "kind": "port",
           "sample": note + f"{items} evaluations of the oracle's eval.rs loop (f64, 56-byte AoS entries, "
                            f"per-eval allocations included), {dt * 1e3:.2f} ms each; table build {t_spec:.1f} s excluded; "
                            f"host has {cores} cores"}
    if all_cores:
        out["all_cores"] = all_cores
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="r12")
    ap.add_argument("--batch", type=int, default=0, help="input sets per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the opt-in matrix-representation side measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import gaast_amd as ga

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # CPU baseline first (rank 0, N = 1 only): its all-cores leg forks workers, which must happen before
    # this process initialises the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(workload_spec(args.workload))
    # GAAST_BENCH_REHEARSAL=1: rehearse the N > 1 control flow on a box with fewer GPUs than ranks (ranks
    # share devices, collectives go through gloo on host copies).  Never set by the driver; numbers
    # from such a run are not measurements.
    rehearsal = world > 1 and os.environ.get("GAAST_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ga.init_device(local_rank)
    stream = torch.cuda.current_stream()
    ga._lib.check(ga.lib().gaast_hip_set_stream(C.c_void_p(stream.cuda_stream)))

    wl = workload_spec(args.workload)
    n, dtype = wl["n"], wl["dtype"]
    batch = args.batch or wl["default_batch"]
    tdt = torch.float32 if dtype == ga.F32 else torch.float64

    # one SpecializedAst (phases 1-3 on the host, once), one device program
    exprs = [ga.mv(ga.Input(s, g, n)) for s, g in enumerate(wl["inputs"])]
    t0 = time.time()
    spec = wl["build"](*exprs).specialize(ga.MetricAlgebra(wl["metric"]), dtype=dtype, flags=wl.get("flags", 0))
    spec.program()
    t_spec = time.time() - t0
    out_mask, out_len = spec.output_info()

    # synthetic inputs, generated on the device: item i of rank r is seeded by (seed, r)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3 + rank)
    ins, in_t = [], []
    for slot, g in enumerate(wl["inputs"]):
        rl = ga.graded.row_len(n, ga.graded._mask_of(g))
        nb = 1 if slot in wl.get("shared", []) else batch          # a batch-1 input is shared by every item
        t = torch.rand((nb, rl), generator=gen, device=dev, dtype=tdt) * 2 - 1
        in_t.append(t)
        ins.append(ga.DeviceMV.wrap_tensor(t, n, g))
    out_t = torch.empty((batch, out_len), device=dev, dtype=tdt)
    out = ga.DeviceMV.wrap_tensor(out_t, n, ga.GradeSet(out_mask))

    def step():
        spec.eval_batch(ins, batch, out=out)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
        from gaast_amd.sharding import max_over_ranks
        wall = max_over_ranks(wall, device=None if rehearsal else dev)
    step_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    kernel_ms = sum(step_ms) / len(step_ms)

    # the same products through the opt-in matrix-representation kernel (not the reference's summation
    # order, so never `value`): reported beside the headline, with its distance from the default path
    alt = None
    if args.workload in ("r12", "r12d", "r8") and rank == 0 and world == 1 and not args.no_alt:
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
        alt = {"kernel": [l for l in spec_alt.launches() if "product" in l][-1], "value": batch / (alt_ms * 1e-3),
               "unit": "products/s", "kernel_ms": alt_ms, "algorithmic_GBps": batch * 3 * (1 << n) * (4 if dtype == ga.F32 else 8) / (alt_ms * 1e-3) * 1e-9,
               "mfma_TFLOPs": batch * (3 * 2 * (1 << (n // 2)) ** 3) / (alt_ms * 1e-3) * 1e-12,
               "max_abs_diff_vs_default_path": diff, "diff_over_eps_normA_normB": diff / ((2.0 ** -23 if dtype == ga.F32 else 2.0 ** -52) * scale),
               "note": f"opt-in GAAST_FLAG_SPINOR_GEMM: {1 << (n // 2)}x{1 << (n // 2)} complex matrix representation, 3 real MFMA products per item; "
                       "norm-wise error bound, not the reference's summation order"}
        del out_alt_t, out_alt, spec_alt

    # final gather of the result shards to rank 0 over RCCL (xGMI), outside the timed region
    gather_ms = None
    if world > 1 and not args.no_gather:
        send = out_t.cpu() if rehearsal else out_t
        glist = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        fence()
        g0 = time.perf_counter()
        dist.gather(send, glist, dst=0)
        fence()
        gather_ms = (time.perf_counter() - g0) * 1e3
        del glist

    if rank == 0:
        items_total = batch * world * args.steps
        value = items_total / wall
        sz = 4 if dtype == ga.F32 else 8
        in_len = sum(t.shape[1] for t in in_t if t.shape[0] == batch)      # shared inputs are read once, not per item
        bytes_item = (in_len + out_len) * sz          # every input/output component touched once
        # one multiply + one add per comp-mul entry (the matrix-representation kernel: what it executes)
        flops_item = wl.get("flops_item", 2 * wl["entries"])
        launches = spec.launches()
        dense = any("product_dense" in l or "product_spinor" in l for l in launches)
        peak_tf = PEAK_FP32_TFLOPS if dtype == ga.F32 else PEAK_FP64_TFLOPS
        ach_tf = flops_item * batch / (kernel_ms * 1e-3) * 1e-12
        ach_gb = bytes_item * batch / (kernel_ms * 1e-3) * 1e-9
        if dense:
            roof = {"bound": "mfma", "achieved": ach_tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": ach_tf / peak_tf,
                    "traffic": None,
                    "note": ("dense product: fp32 vector FMA peak == fp32 MFMA peak (157.3 TFLOP/s); " if dtype == ga.F32 else
                             "dense product: fp64 vector FMA peak == fp64 MFMA peak (78.6 TFLOP/s); ") +
                            f"algorithmic HBM {ach_gb:.1f} GB/s = {ach_gb / PEAK_HBM_GBPS:.4f} of 8 TB/s"}
        else:
            roof = {"bound": "hbm", "achieved": ach_gb, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": ach_gb / PEAK_HBM_GBPS,
                    "traffic": None, "note": f"{len(launches)} launches per evaluation; {ach_tf:.2f} TFLOP/s"}
        pmc = PMC_TRAFFIC.get((args.workload, batch))
        roof["traffic"] = pmc[0] if pmc else None
        roof["traffic_source"] = pmc[1] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)" if pmc else None
        roof["algorithmic_bytes_per_launch"] = bytes_item * batch
        roof["kernel"] = [l for l in launches if "product" in l][-1] if any("product" in l for l in launches) else launches[-1]
        roof["kernel_ms"] = kernel_ms
        roof["flops_per_item"] = flops_item
        roof["bytes_per_item"] = bytes_item
        res = {
            "metric": "full-MV geometric products/sec (dim n); achieved HBM GB/s vs roofline",
            "value": value, "unit": "products/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": wl["dtname"], "data": "synthetic",
            **({"rehearsal": "ranks share GPUs, gloo collectives: control-flow check only, not a measurement"} if rehearsal else {}),
            "config": {"workload": wl["label"], "dim": n, "batch_per_gpu": batch, "global_batch": batch * world,
                       "launches_per_eval": launches, "specialize_s": t_spec},
            "roofline": roof,
        }
        if alt is not None:
            res["matrix_representation"] = alt
        if gather_ms is not None:
            out_bytes = out_len * sz * batch
            res["gather"] = {"ms": gather_ms, "bytes_per_rank": out_bytes,
                             "value_with_gather": items_total / (wall + gather_ms * 1e-3 * args.steps)}
        if cpu is not None:
            res["cpu_baseline"] = cpu
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
