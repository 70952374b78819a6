# A/B of library builds on one box: tools/ab_libs.sh "<workloads>" "<lib dirs under gaast_amd/lib_ab, or 'cur'>" [bench args]
set -e
wl=$1; libs=$2; shift 2
mkdir -p gpurun_out/ab
for rep in 1 2; do
for w in $wl; do
  for lib in $libs; do
    if [ $lib = cur ]; then unset GAAST_HIP_LIB; else export GAAST_HIP_LIB=$PWD/gaast_amd/lib_ab/$lib/libgaast_hip.so; fi
    python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-latency --workload $w "$@" > gpurun_out/ab/${w}_$lib.json 2> gpurun_out/ab/${w}_$lib.err
    python3 -c "import json;d=json.load(open('gpurun_out/ab/${w}_$lib.json'));print('$w $lib %.4g kernel_ms %.4f frac %.3f' % (d['value'],d['roofline'].get('kernel_ms'),d['roofline']['frac']))"
  done
done
done
