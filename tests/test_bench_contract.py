"""bench.py without a GPU: every workload name resolves, the generic gpN names parse, and the CPU baseline
leg (the oracle on a bounded sample, single core + all cores) returns the fields the contract asks for."""
import json

import pytest

import bench


@pytest.mark.parametrize("name", ["r12", "r12s", "r66s", "r12d", "r12ds", "r12x", "r8", "r8d", "r8s", "r8x", "cl41", "cl41g1", "cl41s",
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


def test_traffic_is_reported_only_for_the_kernel_and_library_it_was_measured_with(monkeypatch, tmp_path):
    """roofline.traffic comes from profiles/traffic.json (tools/pmc_traffic.py); a figure recorded for another kernel, or
    with other kernel sources than the loaded library's, is withheld and flagged stale."""
    import gaast_amd
    rev = gaast_amd.lib().gaast_hip_version().decode()
    table = {"r12:65536": {"kernel": "product_dense_mfma[gp n=12]", "library": rev, "bytes": 3.2e9, "source": "x"},
             "r8:1048576": {"kernel": "product_dense_mfma[gp n=8]", "library": "some other build", "bytes": 3.2e9, "source": "y"}}
    monkeypatch.setattr(bench, "_load_traffic_table", lambda: table)
    assert bench._traffic_for("r12", 65536, ["product_dense_mfma[gp n=12]"]) == (3.2e9, "x", False)
    assert bench._traffic_for("r12", 65536, ["product_dense[gp n=12]"]) == (None, "x", True)        # another kernel ran
    assert bench._traffic_for("r8", 1 << 20, ["product_dense_mfma[gp n=8]"]) == (None, "y", True)   # kernels rebuilt since
    assert bench._traffic_for("r12", 512, ["product_dense_mfma[gp n=12]"]) == (None, None, False)   # never measured


def test_committed_traffic_table_is_well_formed():
    for key, ent in bench._load_traffic_table().items():
        workload, batch = key.split(":")
        assert bench.workload_spec(workload)["n"] >= 1 and int(batch) > 0
        assert ent["bytes"] > 0 and ent["kernel"] and ent["library"] and ent["source"]
        assert abs(ent["bytes"] - (2 * ent["fetch_size_kib"] + ent["write_size_kib"]) * 1024) < 1


def test_pmc_traffic_tool_parses_rocprofv3_counter_csv(tmp_path, monkeypatch):
    import csv, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = ["Correlation_Id", "Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    for d, counter, vals in (("f", "FETCH_SIZE", [1000.0, 1002.0, 1001.0]), ("w", "WRITE_SIZE", [500.0, 500.0, 500.0])):
        os.makedirs(tmp_path / d / "host")
        with open(tmp_path / d / "host" / "1_counter_collection.csv", "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(hdr)
            wr.writerow([0, 0, "other_kernel", counter, 7.0])
            for i, v in enumerate(vals):
                wr.writerow([i + 1, i + 1, "void gaast::k_gp_mfma32<false, 256>(gaast::DenseArgs<float>)", counter, v])
    # run the tool on a scratch copy of the repo layout (it writes profiles/traffic.json next to itself)
    scratch = tmp_path / "repo"
    os.makedirs(scratch / "tools")
    os.makedirs(scratch / "profiles")
    (scratch / "tools" / "pmc_traffic.py").write_text(open(os.path.join(root, "tools", "pmc_traffic.py")).read())
    run = subprocess.run([sys.executable, str(scratch / "tools" / "pmc_traffic.py"), "--workload", "r12", "--batch", "65536",
                          "--launch-name", "product_dense_mfma[gp n=12]", "--kernel", "k_gp_mfma32", "--fetch", str(tmp_path / "f"),
                          "--write", str(tmp_path / "w"), "--revision", "rev-x"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    ent = json.load(open(scratch / "profiles" / "traffic.json"))["r12:65536"]
    assert ent["bytes"] == (2 * 1001.0 + 500.0) * 1024 and ent["library"] == "rev-x"


def test_packed_cpu_baseline_reproduces_the_oracle_bit_for_bit():
    """BASELINE.md section 2's `cpu-packed-*` variants run the reference's loop (eval.rs:77-83) over 16-byte packed entries on
    flat rows: same order, same roundings -- so their rows must equal the literal oracle's, bit for bit (1 and 3 threads)."""
    import ctypes as C

    import numpy as np

    from oracle import pyoracle as og
    n = 5
    rng = np.random.default_rng(1)
    L = og.lib()
    full = list(range(n + 1))
    val = lambda: og.GradeMapMV({k: rng.uniform(-1, 1, L.og_n_choose_k(n, k)) for k in full})
    a, b = val(), val()
    spec = (og.mv(a) * og.mv(b)).specialize(og.as_algebra([1.0, 1.0, -1.0, 1.0, -1.0]))
    hnd, n_ent, ll, rl, ol = spec.packed_root_product()
    assert n_ent == 4 ** n and ll == rl == ol == 2 ** n
    batch = 7
    la, ra = rng.uniform(-1, 1, (batch, ll)), rng.uniform(-1, 1, (batch, rl))
    want = spec.eval_batch([a, b], [la, ra], batch, ol)
    dbl = C.POINTER(C.c_double)
    for threads in (1, 3):
        got = np.empty((batch, ol))
        L.og_packed_eval_batch(hnd, n_ent, la.ctypes.data_as(dbl), ll, ra.ctypes.data_as(dbl), rl, got.ctypes.data_as(dbl), ol, batch, threads)
        assert np.array_equal(got, want)
    L.og_packed_free(hnd)
