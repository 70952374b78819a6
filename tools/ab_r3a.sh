# same-box A/B of two builds over the dense workloads: VARIANTS = directories under gaast_amd/lib_ab (or "main")
mkdir -p gpurun_out/r3ab
for rep in 1 2; do
for w in ${W:-r8 r8d gp9f32 gp9f64 r12d gp10f64}; do
for v in ${VARIANTS:-base main}; do
  if [ $v = main ]; then unset GAAST_HIP_LIB; else export GAAST_HIP_LIB=$PWD/gaast_amd/lib_ab/$v/libgaast_hip.so; fi
python bench.py --workload $w --steps 20 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $v', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac %.3f' % r['frac'])"
done; done; done 2>&1 | tee gpurun_out/r3ab/ab_${TAG:-x}.txt
