#!/bin/bash
# Round evidence in one GPU call: rocprofv3 --kernel-trace --stats of the bench command per workload, PMC passes (separate,
# --kernel-trace only), bench JSON lines.  Output under gpurun_out/prof_<tag>/; tools/collect_profiles.py files the
# summaries under profiles/.
#   tools/collect_profiles.sh <tag>
#   tools/collect_profiles.sh <tag> [stats|pmc|all]     (two calls when one would exceed the GPU call's time limit)
tag=${1:-r03}
phase=${2:-all}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# the kernel revision of the library that is MEASURED (bench.py reports a PMC traffic figure only for the revision it was taken with)
python3 -c "import gaast_amd; print(gaast_amd.lib().gaast_hip_version().decode())" > $out/library_version.txt
stats() { # name, bench args...
  name=$1; shift
  d=$out/stats_$name
  rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency "$@" > $d/bench.json 2> $d/log.txt || echo "stats $name failed"
}
if [ $phase != pmc ]; then
stats r12 --workload r12
stats r8 --workload r8
stats cl41 --workload cl41
stats cl41g1 --workload cl41g1
stats cl41s --workload cl41s
stats r12s --workload r12s
stats r8s --workload r8s
stats r12d --workload r12d
stats r8d --workload r8d
stats gp9f32 --workload gp9f32 --no-alt
stats gp10f32 --workload gp10f32 --no-alt
stats gp12f32ee --workload gp12f32ee --no-alt
stats gp12f64ee --workload gp12f64ee --no-alt
stats gp7f32 --workload gp7f32 --no-alt
stats gp8f32ee --workload gp8f32ee --no-alt
stats gp8f64ee --workload gp8f64ee --no-alt
stats sand9g1 --workload sand9g1 --no-alt
stats sand10g1 --workload sand10g1 --no-alt
stats sand8 --workload sand8 --no-alt
stats sand9 --workload sand9 --no-alt
stats sand10 --workload sand10 --no-alt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
# the N > 1 control flow with the LIBRARY's gather over the test transport (ranks share this box's GPU)
gcc -std=gnu11 -shared -fPIC -O1 -fvisibility=hidden -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_stub.c -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-Bsymbolic -Wl,-rpath,/opt/rocm/lib -o $out/librccl_stub.so
GAAST_BENCH_REHEARSAL=1 GAAST_BENCH_REHEARSAL_TRANSPORT=$PWD/$out/librccl_stub.so python3 bench.py --gpus 2 --steps 3 --warmup 1 --batch 4096 > $out/bench_rehearsal_2ranks.json 2> $out/bench_rehearsal.err
GAAST_BENCH_REHEARSAL=1 GAAST_BENCH_REHEARSAL_TRANSPORT=$PWD/$out/librccl_stub.so python3 bench.py --gpus 3 --steps 2 --warmup 1 --batch 1000 > $out/bench_rehearsal_3ranks.json 2>> $out/bench_rehearsal.err
rm -f $out/librccl_stub.so
python3 bench.py --workload r8 > $out/bench_r8.json 2>/dev/null
python3 bench.py --workload cl41 > $out/bench_cl41.json 2>/dev/null
fi
if [ $phase != stats ]; then
for spec in "r12:k_gp_mfma32:--workload r12" "r8:k_gp_mfma16:--workload r8" "cl41:gaast_jit:--workload cl41" "cl41g1:gaast_jit:--workload cl41g1" "r12s:k_gp_spinor12s:--workload r12s" "r12d:k_gp_mfma16:--workload r12d" "r8d:k_gp_mfma16:--workload r8d" "gp7f32:k_gp_mfma7:--workload gp7f32" "sand8:k_gp_mfma7:--workload sand8" "sand9g1:k_product_ell_chain:--workload sand9g1" "sand9:k_gp_mfma16:--workload sand9" "sand10:k_gp_mfma16:--workload sand10" "gp12f32ee:k_gp_mfma32:--workload gp12f32ee"; do
  name=${spec%%:*}; rest=${spec#*:}; kern=${rest%%:*}; args=${rest#*:}
  tools/pmc_pass.sh ${tag}_$name $kern $args > $out/pmc_$name.txt 2>&1
  cp gpurun_out/pmc_${tag}_$name/summary.csv $out/pmc_${name}_summary.csv
done
bash tools/sweep_dims.sh > $out/sweep_dims.txt 2>&1
fi
ls $out
