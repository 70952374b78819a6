"""Diagnostic: phase cycles of k_gp_mfma16 from a build with s_memtime stamps (gaast_hip_debug_phases; scratch build only).
Usage: GAAST_HIP_LIB=<stamped build> python tools/phase_probe.py [n]"""
import ctypes, sys
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import gaast_amd as ga

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = (1 << 20) >> (2 * (n - 8))
rng = np.random.default_rng(0)
full = list(range(n + 1))
a, b = ga.mv(ga.Input(0, full, n)), ga.mv(ga.Input(1, full, n))
spec = (a * b).specialize([1.0] * n, dtype=ga.F32)
A = rng.uniform(-1, 1, (batch, 1 << n)).astype(np.float32)
B = rng.uniform(-1, 1, (batch, 1 << n)).astype(np.float32)
ins = [ga.DeviceMV.from_rows(n, full, X, ga.F32) for X in (A, B)]
out = spec.eval_batch(ins, batch)
lib = ga.lib()
lib.gaast_hip_synchronize()
buf = (ctypes.c_ulonglong * 8)()
f = lib.gaast_hip_debug_phases
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
f(buf, 1)
for _ in range(5):
    out = spec.eval_batch(ins, batch)
lib.gaast_hip_synchronize()
f(buf, 1)
waves = buf[5]
groups = 5 * ((batch + 3) // 4) * (1 if n == 8 else 2)
names = ["stage (scatter into LDS)", "issue next fetch", "product loop", "result rows", "whole kernel per wave"]
print(f"n={n} batch={batch} waves={waves} group-waves={groups}")
for q in range(4):
    print(f"  {names[q]:28s} {buf[q] / groups:10.0f} s_memtime ticks per group (100 MHz ticks x clock ratio: see guide)")
print(f"  {names[4]:28s} {buf[4] / waves:10.0f} per wave; per group {buf[4] / groups:10.0f}")
