#!/usr/bin/env python3
"""HBM bytes per launch from rocprofv3 PMC passes -> profiles/traffic.json (what bench.py reports as roofline.traffic).

    python3 tools/pmc_traffic.py --workload r12 --batch 65536 --launch-name 'product_dense_mfma[gp n=12]' \
        --kernel k_gp_mfma32 --fetch gpurun_out/pmc_r12_FETCH_SIZE --write gpurun_out/pmc_r12_WRITE_SIZE --tag r02

Each of --fetch / --write is the output directory of ONE separate pass
    rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d <dir> -- python3 bench.py --workload ... --steps 3 --warmup 1 ...
    rocprofv3 --pmc WRITE_SIZE  ...
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE do not fit one pass; both are in KiB; on gfx950 FETCH_SIZE
counts a wide coalesced read at half its bytes -> x2).  The entry records the library's kernel-source revision
(gaast_hip_version()); bench.py refuses to report a figure measured with other kernel sources (traffic_stale).
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(directory, counter, kernel_substr):
    """median per-dispatch value of `counter` over the dispatches of kernels whose name contains kernel_substr"""
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for a kernel matching {kernel_substr!r} under {directory}")
    vals.sort()
    return vals[len(vals) // 2], len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--launch-name", required=True, help="the step's name in SpecializedAst.launches()")
    ap.add_argument("--kernel", required=True, help="substring of the HIP kernel's name in the rocprofv3 CSV")
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--tag", default="")
    ap.add_argument("--revision", default=None, help="kernel-source revision (default: the built library's)")
    args = ap.parse_args()
    fetch_kib, nf = per_dispatch(args.fetch, "FETCH_SIZE", args.kernel)
    write_kib, nw = per_dispatch(args.write, "WRITE_SIZE", args.kernel)
    rev = args.revision
    if rev is None:
        sys.path.insert(0, ROOT)
        import gaast_amd
        rev = gaast_amd.lib().gaast_hip_version().decode()
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        table = {}
    table[f"{args.workload}:{args.batch}"] = {
        "kernel": args.launch_name, "hip_kernel": args.kernel, "library": rev,
        "fetch_size_kib": fetch_kib, "write_size_kib": write_kib,
        "bytes": 2 * fetch_kib * 1024 + write_kib * 1024,
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes ({nf} / {nw} dispatches, median), FETCH x2 "
                  f"(gfx950){', ' + args.tag if args.tag else ''}"}
    with open(path, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(table[f"{args.workload}:{args.batch}"]))


if __name__ == "__main__":
    main()
