"""Register budgets of the hot kernels, read from the gfx950 code object inside libgaast_hip.so (no GPU needed).

Occupancy decides these kernels: k_gp_mfma32p<false, 12> (the headline, BASELINE configs[2]) must fit TWO waves per SIMD
(vgpr + agpr <= 256 of the 512 registers per lane); round 3 once lost 18 % of the headline to 13 extra registers from an
unrelated code path (267 -> one wave per SIMD).  No kernel may spill."""
import re
import struct
import subprocess

import pytest

from gaast_amd import _lib

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    data = open(_lib.LIB_PATH, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle in the library"
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off, co = i + 32, None
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            co = data[i + o:i + o + sz]
    assert co, "no gfx950 code object"
    path = tmp_path_factory.mktemp("co") / "gaast.co"
    path.write_bytes(co)
    notes = subprocess.run([READELF, "--notes", str(path)], capture_output=True, text=True, check=True).stdout
    out = {}
    for block in notes.split("  - .agpr_count:")[1:]:
        agpr = int(block.split()[0])
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1))
        sspill = int(re.search(r"\.sgpr_spill_count:\s+(\d+)", block).group(1))
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", block).group(1))
        out[name] = dict(vgpr=vgpr, agpr=agpr, spill=spill, sgpr_spill=sspill, lds=lds)
    names = subprocess.run([CXXFILT], input="\n".join(out), capture_output=True, text=True, check=True).stdout.split("\n")
    return {d.replace("void gaast::", "").split("(")[0]: v for d, v in zip(names, out.values())}


def _find(kernels, prefix):
    hits = {k: v for k, v in kernels.items() if k.startswith(prefix)}
    assert hits, (prefix, sorted(kernels)[:20])
    return hits


def test_spills_stay_out_of_the_hot_kernels(kernels):
    """a few registers spill in cold instantiations of the one-item-per-workgroup kernel at its register caps -- the 1,024-thread
    n = 12 f64 ones (128 registers per thread) for degenerate metrics / a rescaled basis / a chained list, and chained ones at
    n = 9 ... 11; the plain (false, false) instantiations of non-degenerate metrics never spill, and no other kernel does"""
    bad = {k: v["spill"] for k, v in kernels.items() if v["spill"]}
    def cold(k):
        if not k.startswith("k_gp_mfma16x4<"):
            return False
        args = [a.strip() for a in k[len("k_gp_mfma16x4<"):-1].split(",")]     # T, DEGENERATE, N, MODE, SCALED, CHAINED
        return args[1] == "true" or args[4] == "true" or args[5] == "true"
    assert all(cold(k) for k in bad), bad
    assert all(v <= 32 for v in bad.values()), bad


@pytest.mark.parametrize("prefix,waves", [
    ("k_gp_mfma32p<false, 12, false, false>", 2),                 # the headline: two waves per SIMD
    ("k_gp_mfma32p<false, 10, false, false>", 2), ("k_gp_mfma32p<false, 11, false, false>", 2),
    ("k_gp_mfma16x4<float, false, 8, 2, false, false>", 6),       # BASELINE configs[1]
    ("k_gp_mfma16x4<double, false, 12, 2, false, false>", 4),     # r12d: 16 waves per item, one item per CU
    ("k_gp_spinor12s<5, true>", 2),
    ("k_gp_mfma7<float, 2, false, false>", 8), ("k_gp_mfma7<double, 2, false, false>", 6),   # one wave per item: the waves of a SIMD are its latency hiding
    # round 4
    ("k_gp_mfma6<float, false, true>", 8), ("k_gp_mfma6<double, false, true>", 5),           # n = 6: one wave per item, four items in flight each
    ("k_gp_mfma16x4<double, false, 8, 2, false, true>", 2),      # sand9: the list's 72 operand addresses in registers beside the unrolled matrix loop
    ("k_gp_mfma7<double, 2, false, true>", 4), ("k_gp_mfma7<float, 2, false, true>", 4),     # sand8: amdgpu_waves_per_eu(4)
])
def test_hot_kernels_keep_their_occupancy(kernels, prefix, waves):
    for name, k in _find(kernels, prefix).items():
        regs = -(-k["vgpr"] // 8) * 8          # .vgpr_count is the unified total (arch + accumulation registers); granule: 8
        assert k["spill"] == 0, (name, k)
        assert regs * waves <= 512, (name, k, f"{regs} registers: fewer than {waves} waves per SIMD")


def test_library_revision_matches_the_sources_it_was_built_from():
    """gaast_hip_version() carries an md5 of everything that decides which kernel code runs (csrc/Makefile: KREV); PMC traffic figures are
    keyed to it (profiles/traffic.json).  The string is compiled into runtime.o, which a change in plan.cpp alone once failed to rebuild:
    the library must report the revision of the sources in the tree (default build switches)."""
    import glob
    import hashlib
    import os
    import gaast_amd
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaast_amd", "csrc")
    files = sorted(glob.glob(os.path.join(root, "device", "*.hip.hpp"))) + [os.path.join(root, "device", f) for f in
                                                                              ("runtime.hip", "plan.cpp", "plan.hpp", "spinor_basis.hpp")]
    h = hashlib.md5()
    for f in files:
        h.update(open(f, "rb").read())
    h.update(b"\n")          # echo "$(KFLAGS)" with no switches
    version = gaast_amd.lib().gaast_hip_version().decode()
    if os.environ.get("GAAST_HIP_LIB"):
        pytest.skip("an A/B build is selected")
    assert version.endswith(h.hexdigest()[:12]), (version, h.hexdigest()[:12])


def test_a_change_of_the_build_switches_rebuilds_every_object_that_sees_them():
    """KFLAGS reaches plan.cpp too (GAAST_JIT_NT, GAAST_INTERP_BUDGET_KB): with the revision hashing KFLAGS, a change of switches
    in an existing object directory must recompile plan.o and runtime.o, or the library reports the new revision over old code
    (make -n: nothing is built)."""
    import os
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaast_amd", "csrc")
    if not os.path.isdir(os.path.join(csrc, "_obj")):
        pytest.skip("no object directory (the library was not built in this tree)")
    plain = subprocess.run(["make", "-n", "-C", csrc], capture_output=True, text=True, check=True).stdout
    assert "plan.cpp" not in plain and "runtime.hip" not in plain, "the tree is not up to date: run build() first"
    dry = subprocess.run(["make", "-n", "-C", csrc, "KFLAGS=-DGAAST_INTERP_BUDGET_KB=96"], capture_output=True, text=True, check=True).stdout
    assert "device/plan.cpp" in dry and "device/runtime.hip" in dry and "host/expr.cpp" in dry, dry
