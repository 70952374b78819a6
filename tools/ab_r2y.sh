# k_gp_spinor12s with fewer vector instructions (tables expanded once per launch, pair-wise transforms, uniform-base I/O):
# spinor tests, then r12s / r66s / gp11f32s, new build against the previous one (gaast_amd/lib_base) on the same box
mkdir -p gpurun_out/r2y
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "spinor" > gpurun_out/r2y/t.txt 2>&1; tail -3 gpurun_out/r2y/t.txt
for rep in 1 2; do
for lib in lib lib_base; do
  for w in r12s r66s gp11f32s; do
GAAST_HIP_LIB=$PWD/gaast_amd/$lib/libgaast_hip.so python bench.py --workload $w --steps 20 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib $w', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'])"
  done
done
done 2>&1 | tee gpurun_out/r2y/ab.txt
