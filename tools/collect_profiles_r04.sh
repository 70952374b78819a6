#!/bin/bash
# Round-4 evidence: rocprofv3 --kernel-trace --stats of the bench command per workload, PMC passes (separate, --kernel-trace only),
# bench JSON lines.  Output under gpurun_out/prof_r04/; `python3 tools/collect_profiles.py r04` files the summaries under profiles/.
#   tools/collect_profiles_r04.sh stats1 | stats2 | pmc1 | pmc2 | pmc3      (one GPU call each: a call is limited to 20 minutes)
phase=${1:-stats1}
out=gpurun_out/prof_r04
mkdir -p $out
export TMPDIR=/tmp
python3 -c "import gaast_amd; print(gaast_amd.lib().gaast_hip_version().decode())" > $out/library_version.txt
stats() { # name, bench args...
  name=$1; shift
  d=$out/stats_$name
  rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency --no-alt --no-configs "$@" > $d/bench.json 2> $d/log.txt || echo "stats $name failed"
}
pmc() { # name, kernel substring, bench args...
  name=$1; kern=$2; shift 2
  bash tools/pmc_pass.sh r04_$name $kern --no-configs "$@" > $out/pmc_$name.txt 2>&1
  cp gpurun_out/pmc_r04_$name/summary.csv $out/pmc_${name}_summary.csv
}
case $phase in
stats1)
  stats r12 --workload r12
  stats r8 --workload r8
  stats cl41 --workload cl41
  stats cl41s --workload cl41s
  stats r12d --workload r12d
  stats gp6f32 --workload gp6f32
  stats gp6f64 --workload gp6f64
  stats gp7f32ee --workload gp7f32ee
  stats sand8g1 --workload sand8g1
  stats sand9g1 --workload sand9g1
  stats sand10g1 --workload sand10g1
  stats sand9g1x --workload sand9g1x
  python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
  ;;
stats2)
  stats vinv8 --workload vinv8
  stats vinv12 --workload vinv12
  stats proj8 --workload proj8
  stats proj12 --workload proj12
  stats cfg1_8 --workload cfg1_8
  stats cfg1_12 --workload cfg1_12
  stats unary12 --workload unary12
  stats unary12_per_arm --workload unary12 --flags 0x800
  stats vinv12_per_arm --workload vinv12 --flags 0x800
  stats sand8 --workload sand8
  stats sand9 --workload sand9
  stats sand10 --workload sand10
  python3 -m pytest tests -m gpu -q > $out/gpu_tests_final.txt 2>&1
  tail -3 $out/gpu_tests_final.txt
  ;;
pmc1)
  pmc r12 k_gp_mfma32 --workload r12
  pmc gp6f32 k_gp_mfma6 --workload gp6f32
  pmc gp6f64 k_gp_mfma6 --workload gp6f64
  pmc sand9g1 gaast_chain --workload sand9g1
  pmc sand9g1x gaast_chain --workload sand9g1x
  pmc cl41s gaast_jit --workload cl41s
  ;;
pmc2)
  pmc vinv8 gaast_jit --workload vinv8
  pmc vinv12 k_reduce_scale --workload vinv12
  pmc proj12 gaast_jit --workload proj12
  pmc cfg1_8 gaast_chain --workload cfg1_8
  pmc unary12 k_elementwise --workload unary12
  pmc cl41 gaast_jit --workload cl41
  pmc r8 k_gp_mfma16 --workload r8
  pmc r12d k_gp_mfma16x4 --workload r12d
  ;;
pmc3)
  pmc sand8 k_gp_mfma7 --workload sand8
  pmc sand9 k_gp_mfma16x4 --workload sand9
  pmc sand10 k_gp_mfma16x4 --workload sand10
  pmc sand10g1 gaast_chain --workload sand10g1
  ;;
esac
ls $out | head -80
