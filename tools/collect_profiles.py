#!/usr/bin/env python3
"""Files what tools/collect_profiles.sh left under gpurun_out/prof_<tag>/ into profiles/ (tracked): per-workload
rocprofv3 --kernel-trace --stats summaries, PMC per-dispatch medians, and profiles/traffic.json (the HBM bytes per launch
bench.py reports as roofline.traffic, keyed to the library's kernel revision).

    python3 tools/collect_profiles.py r02
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HIP_KERNEL = {"r12": "k_gp_mfma32p", "r8": "k_gp_mfma16x4<float>", "cl41": "gaast_jit", "cl41g1": "gaast_jit", "r12s": "k_gp_spinor12s",
              "r12d": "k_gp_mfma16x4<double>", "r8d": "k_gp_mfma16x4<double>", "sand9": "k_gp_mfma16x4<double,...,8,0,false,true> (chained)",
              "sand10": "k_gp_mfma16x4<double,...,9,0,false,true> (chained)", "gp12f32ee": "k_gp_mfma32p<false,11>",
              "gp7f32": "k_gp_mfma7<float>", "sand8": "k_gp_mfma7<double,...,false,true> (chained)", "sand9g1": "gaast_chain (hiprtc)",
              "sand9g1x": "gaast_chain (hiprtc), reference order", "sand8g1": "gaast_chain (hiprtc)", "sand10g1": "gaast_chain (hiprtc)",
              "gp6f32": "k_gp_mfma6<float,false,true>", "gp6f64": "k_gp_mfma6<double,false,true>", "cl41s": "gaast_jit",
              "vinv8": "gaast_jit (slabs in LDS)", "proj12": "gaast_jit (slabs in LDS)", "vinv12": "k_reduce_scale<double>",
              "cfg1_8": "gaast_chain (hiprtc), one list", "cfg1_12": "gaast_chain (hiprtc), one list", "unary12": "k_elementwise<double,4>"}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    # the revision of the library that ran on the GPU box (collect_profiles.sh records it), not of whatever is built here now
    rev = open(os.path.join(src, "library_version.txt")).read().strip()
    for d in sorted(glob.glob(os.path.join(src, "stats_*"))):
        name = os.path.basename(d)[len("stats_"):]
        # a repeated collection merges into the same directory: the newest summary is the one bench.json belongs to
        files = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        if not files:
            print("no kernel stats for", name)
            continue
        rows = list(csv.reader(open(files[-1], newline="")))
        try:
            bench = json.load(open(os.path.join(d, "bench.json")))
            note = (f"bench line of the same run: value {bench['value']:.4g} {bench['unit']}, kernel_ms {bench['roofline']['kernel_ms']:.4f}, "
                    f"roofline {bench['roofline']['bound']} frac {bench['roofline']['frac']:.3f}")
        except (OSError, ValueError, KeyError):
            note = "bench line unavailable"
        with open(os.path.join(dst, f"{tag}_bench_{name}_kernel_stats.csv"), "w", newline="") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency --workload {name}   ({tag}; {rev})\n")
            f.write(f"# {note}\n")
            w = csv.writer(f)
            for r in rows[:9]:
                w.writerow([c[:100] for c in r])
    table_path = os.path.join(dst, "traffic.json")
    try:
        table = json.load(open(table_path))
    except (OSError, ValueError):
        table = {}
    for path in sorted(glob.glob(os.path.join(src, "pmc_*_summary.csv"))):
        name = os.path.basename(path)[len("pmc_"):-len("_summary.csv")]
        rows = list(csv.DictReader(open(path, newline="")))
        vals = {r["counter"]: float(r["per_dispatch_median"]) for r in rows}
        with open(os.path.join(dst, f"{tag}_{name}_pmc_counters.csv"), "w") as f:
            f.write(f"# {tag} -- PMC counters of {HIP_KERNEL.get(name, '?')} (bench workload {name}, default batch), per dispatch (median); {rev}\n")
            f.write("# separate passes (tools/pmc_pass.sh):  rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-latency --workload " + name + "\n")
            f.write(open(path).read())
        try:
            bench = json.load(open(os.path.join(src, f"stats_{name}", "bench.json")))
        except (OSError, ValueError):
            continue
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            table[f"{name}:{bench['config']['batch_per_gpu']}"] = {
                "kernel": bench["roofline"]["kernel"], "hip_kernel": HIP_KERNEL.get(name, ""), "library": rev,
                "fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
                "bytes": 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024,
                "algorithmic_bytes": bench["roofline"]["algorithmic_bytes_per_launch"],
                "source": f"profiles/{tag}_{name}_pmc_counters.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 for gfx950)"}
    # entries measured with other kernel sources stay in the file but bench.py reports them as stale
    json.dump(table, open(table_path, "w"), indent=1, sort_keys=True)
    for name in ("bench_default.json", "bench_rehearsal_2ranks.json", "bench_rehearsal_3ranks.json", "bench_r8.json", "bench_cl41.json", "sweep_dims.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            open(os.path.join(dst, f"{tag}_{name}"), "w").write(open(p).read())
    print(sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
