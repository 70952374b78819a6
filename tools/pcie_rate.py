#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline workload: host rows -> gaast_hip_mv_upload_rows -> gaast_hip_eval ->
gaast_hip_mv_download_rows, pageable host memory, 8,192 R^12 f32 products per call (384 MiB over the link).  Never the bench's
`value` (that one starts with the rows resident in HBM)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaast_amd as ga

n, batch = 12, 8192
full = list(range(n + 1))
spec = (ga.mv(ga.Input(0, full, n)) * ga.mv(ga.Input(1, full, n))).specialize(n, dtype=ga.F32)
rng = np.random.default_rng(0)
a = rng.uniform(-1, 1, (batch, 1 << n)).astype(np.float32)
b = rng.uniform(-1, 1, (batch, 1 << n)).astype(np.float32)
da, db = ga.DeviceMV.alloc(n, full, batch, ga.F32), ga.DeviceMV.alloc(n, full, batch, ga.F32)
out = ga.DeviceMV.alloc(n, full, batch, ga.F32)
for rep in range(3):
    t0 = time.perf_counter()
    da.upload_rows(a)
    db.upload_rows(b)
    t1 = time.perf_counter()
    spec.eval_batch([da, db], batch, out=out)
    ga.lib().gaast_hip_synchronize()
    t2 = time.perf_counter()
    rows = out.download_rows()
    t3 = time.perf_counter()
    print(f"rep {rep}: upload {(t1 - t0) * 1e3:.1f} ms ({2 * a.nbytes / (t1 - t0) * 1e-9:.1f} GB/s), eval {(t2 - t1) * 1e3:.1f} ms, "
          f"download {(t3 - t2) * 1e3:.1f} ms ({rows.nbytes / (t3 - t2) * 1e-9:.1f} GB/s): {batch / (t3 - t0):.4g} products/s PCIe-inclusive")
