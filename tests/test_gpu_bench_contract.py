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


@pytest.mark.parametrize("with_transport", [True, False])
def test_bench_self_launches_two_ranks_and_gathers(tmp_path, with_transport):
    """`python bench.py --gpus 2` with NO launcher: the parent starts the ranks before touching a GPU.  Rehearsal mode
    (both ranks on this box's one GPU) exercises the real N > 1 control flow: shards, the chunked overlapped gather,
    max-over-ranks timing, one JSON line from rank 0.  With the test transport the gather is the LIBRARY's
    (gaast_hip_eval_gather / gaast_hip_gather_rows, `library_communicator: true`); without it the torch.distributed
    stand-in over gloo."""
    from helpers import build_rccl_stub
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GAAST_BENCH_REHEARSAL"] = "1"
    if with_transport:
        env["GAAST_BENCH_REHEARSAL_TRANSPORT"] = build_rccl_stub(tmp_path)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "300"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, run.stdout[-2000:]
    d = json.loads(lines[0])
    assert KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["global_batch"] == 600 and d["rccl_ranks"] == 2 and "rehearsal" in d
    assert d["config"]["shards"] == [300, 300]
    g = d["gather"]
    assert "error" not in g, g
    assert g["chunks"] == 4 and g["value_with_gather"] > 0 and g["value_with_blocking_gather"] > 0
    assert g["bytes_per_rank"] == 300 * 4096 * 4
    assert g["library_communicator"] is with_transport
    if with_transport:
        assert g["gathered_rows"] == 600


def test_a_failed_gather_ends_the_job_non_zero(tmp_path):
    """A hung or failed exchange is a FAILED run: the throughput line is still printed (gather.error), the status is
    GATHER_FAILED_EXIT.  Provoked with a transport file that does not exist on rank 1 only... every rank then agrees to
    fall back -- so instead the failure is injected where the legs run: GAAST_BENCH_FAIL_GATHER=1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["GAAST_BENCH_REHEARSAL"] = "1"
    env["GAAST_BENCH_FAIL_GATHER"] = "1"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--batch", "64"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert run.returncode == 3, (run.returncode, run.stderr[-2000:])
    d = json.loads([l for l in run.stdout.splitlines() if l.strip()][0])
    assert d["value"] > 0 and "error" in d["gather"]


def test_default_line_carries_the_other_single_gpu_configs():
    """The DEFAULT run (no --workload, no --batch) also times BASELINE configs[1] (R^8 f32), configs[4] (R^{4,1} sandwich,
    f64) and R^12 in f64, each with its own roofline object, so that they stop resting on builder-run profiles -- and, since
    round 4, one workload per kernel family added that round (n = 6 on the matrix cores, the projected sandwich as one launch of
    two lists, the versor inverse and d = (a + b * c).g(2) beyond R^3 as one launch each)."""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-alt", "--no-latency"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    d = json.loads([l for l in run.stdout.splitlines() if l.strip()][0])
    assert d["config"]["batch_per_gpu"] == 65536 and d["roofline"]["kernel"].endswith(":: k_gp_mfma32p<false,12>")
    side = {c["key"]: c for c in d["configs"]}
    assert set(side) == {"r8", "cl41", "r12d", "gp6f32", "sand9g1", "vinv12", "cfg1_8"}
    for key, bound, kern in (("r8", "mfma", "k_gp_mfma16x4<float,false,8,"), ("cl41", "hbm", "ast_jit"), ("r12d", "mfma", "k_gp_mfma16x4<double,false,12,"),
                             ("gp6f32", "hbm", "k_gp_mfma6<float,"), ("sand9g1", "hbm", "gaast_chain<double>"), ("vinv12", "hbm", "k_reduce_scale<double>"),
                             ("cfg1_8", "hbm", "gaast_chain<double>[one list")):
        c = side[key]
        assert "error" not in c, c
        r = c["roofline"]
        assert r["bound"] == bound and kern in r["kernel"] and 0.1 < r["frac"] < 1.0 and c["value"] > 0
        assert len(c["launches_per_eval"]) == 1 if key in ("sand9g1", "vinv12", "cfg1_8") else True
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_bench_default_multi_gpu_workload_is_config_4(monkeypatch):
    """Without --batch, N > 1 runs BASELINE configs[3]: 1,048,576 input sets cut into contiguous shards."""
    from gaast_amd.sharding import shard_range
    spans = [shard_range(1 << 20, r, 8) for r in range(8)]
    assert all(hi - lo == 131072 for lo, hi in spans)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "global_batch = 1 << 20" in src and "BASELINE configs[3]" in src
