"""bench.py end to end on the GPU (small batch): exactly one JSON line on stdout with the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


@pytest.mark.parametrize("args,bound", [(["--workload", "r12", "--batch", "512"], "mfma"),
                                        (["--workload", "cl41", "--batch", "65536"], "hbm")])
def test_bench_prints_one_json_line(args, bound):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"] + args, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, run.stdout[-2000:]
    d = json.loads(lines[0])
    assert KEYS <= set(d), sorted(KEYS - set(d))
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == bound and r["unit"] == ("TFLOP/s" if bound == "mfma" else "GB/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
