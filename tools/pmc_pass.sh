#!/bin/bash
# One rocprofv3 PMC pass per counter group over a bench.py workload (separate passes, --kernel-trace only, the program
# itself after `--`); prints per-dispatch medians of the kernel whose name contains <kernel substring>.
#   tools/pmc_pass.sh <tag> <kernel substring> <bench args...>        output: gpurun_out/pmc_<tag>/summary.csv
tag=$1; kern=$2; shift 2
export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | tr ' ' '+')
  d=gpurun_out/pmc_${tag}/$name
  rm -rf $d; mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-latency "$@" > $d/log.txt 2>&1 || echo "pass $name failed: $(tail -2 $d/log.txt)"
done
python3 - "$tag" "$kern" <<'PY'
import csv, glob, sys, collections
tag, kern = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(list)
for path in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if kern in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
out = open(f"gpurun_out/pmc_{tag}/summary.csv", "w")
out.write("counter,per_dispatch_median,dispatches\n")
for k in sorted(vals):
    v = sorted(vals[k])
    out.write(f"{k},{v[len(v)//2]:.6g},{len(v)}\n")
out.close()
print(open(f"gpurun_out/pmc_{tag}/summary.csv").read())
PY
# the raw per-dispatch CSVs are tens of MiB per pass: gpurun copies at most 64 MiB back, the summary is what is filed
find gpurun_out/pmc_${tag} -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
