"""Host-side logic (phases 1-3 mirror, launch-plan builder with every kernel table, wire format) compiled
with g++ -fsanitize=address,undefined and driven through the C API -- the CPU-only sanitizer run the
GPU pool cannot offer.  See tests/cpp/host_sanitize_driver.cpp for what is exercised."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_is_clean_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "gaast_amd", "csrc")
    exe = tmp_path / "host_asan"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(csrc, "device"), "-I", os.path.join(csrc, "common"), "-I", os.path.join(csrc, "host"),
           os.path.join(ROOT, "tests", "cpp", "host_sanitize_driver.cpp")]
    cmd += [os.path.join(csrc, f) for f in ("host/expr.cpp", "host/c_api_host.cpp", "host/wire.cpp", "device/plan.cpp")]
    subprocess.run(cmd + ["-o", str(exe)], check=True, cwd=csrc)
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert run.stdout.strip().endswith("ALL OK")
