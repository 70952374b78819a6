"""bench.py without a GPU: every workload name resolves, the generic gpN names parse, and the CPU baseline
leg (the oracle on a bounded sample, single core + all cores) returns the fields the contract asks for."""
import json

import pytest

import bench


@pytest.mark.parametrize("name", ["r12", "r12s", "r12d", "r12ds", "r12x", "r8", "r8d", "r8s", "r8x", "cl41", "cl41g1", "cl41s",
                                  "gp6f32", "gp10f64", "gp9f32s", "gp8f64x"])
def test_workloads_resolve(name):
    wl = bench.workload_spec(name)
    assert wl["n"] >= 1 and len(wl["metric"]) == wl["n"] and wl["default_batch"] > 0
    assert wl["dtname"] in ("f32", "f64") and callable(wl["build"]) and wl["entries"] > 0
    assert isinstance(wl["label"], str) and wl["label"]


def test_unknown_workload_is_an_error():
    with pytest.raises(SystemExit):
        bench.workload_spec("nope")


def test_cpu_baseline_fields():
    base = bench.cpu_baseline(bench.workload_spec("cl41"), budget_s=0.5)
    assert base["kind"] == "port" and base["cores"] == 1 and base["unit"] == "products/s" and base["value"] > 0
    assert isinstance(base["sample"], str) and "evaluations" in base["sample"]
    if "all_cores" in base:
        assert base["all_cores"]["cores"] >= 2 and base["all_cores"]["value"] > 0
    json.dumps(base)   # must be serialisable into the one JSON line


def test_pmc_traffic_table_points_at_committed_profiles():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for (workload, batch), (traffic, source) in bench.PMC_TRAFFIC.items():
        assert traffic > 0 and os.path.exists(os.path.join(root, source)), source
        assert bench.workload_spec(workload)["default_batch"] == batch
