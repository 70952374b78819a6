"""Diagnostic: error of the dense kernel vs the oracle in units of eps * sum|terms| per component."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gaast_amd as ga
from helpers import *

def gp(n):
    return lambda B: B.input(0, full_grades(n), n) * B.input(1, full_grades(n), n)

for metric, dtype in [([1.0]*6, ga.F64), ([1.0]*8, ga.F64), ([1.0]*8, ga.F32), ([1.0, 1.0, 1.0, 1.0, 0.0, 1.0, -1.0, 0.0], ga.F64),
                      ([1.0, 1.0, 1.0, 1.0, -1.0, 1.0, -1.0], ga.F64), ([1.0]*10, ga.F32)]:
    n = len(metric); batch = 16
    rng = np.random.default_rng(3)
    npdt = np.float32 if dtype == ga.F32 else np.float64
    rows = {0: rows_of(n, full_grades(n), batch, rng, npdt), 1: rows_of(n, full_grades(n), batch, rng, npdt)}
    want, _ = oracle_eval_batch(gp(n), metric, rows, batch)
    got, mask, spec = hip_eval_batch(gp(n), metric, rows, batch, dtype=dtype)
    eps = 2.0**-24 if dtype == ga.F32 else 2.0**-53
    worst = 0
    for i in range(batch):
        S = gp_bits(n, np.abs(metric), np.abs(row_to_bits(n, full_grades(n), rows[0][i])), np.abs(row_to_bits(n, full_grades(n), rows[1][i])), absolute=True)
        S = bits_to_row(n, full_grades(n), S)
        err = np.abs(got[i].astype(np.float64) - want[i])
        ratio = err / (eps * np.maximum(S, 1e-300))
        j = int(np.argmax(ratio))
        worst = max(worst, ratio[j])
    print(metric, "f32" if dtype == ga.F32 else "f64", "max err/(eps*sum|t|) =", worst, spec.launches())
