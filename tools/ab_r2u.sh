mkdir -p gpurun_out/r2u
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_dense_oracle.py -m gpu -q -k "dense or n6 or n7 or n8 or r8 or R8 or f64 or basis" > gpurun_out/r2u/t.txt 2>&1; tail -4 gpurun_out/r2u/t.txt
for rep in 1 2; do
for v in A S0; do
  if [ $v = A ]; then unset GAAST_HIP_LIB; else export GAAST_HIP_LIB=$PWD/gaast_amd/lib_ab/$v/libgaast_hip.so; fi
  for w in r8d gp6f64 gp7f64 gp9f64 gp10f64 gp6f32 gp7f32; do
python bench.py --workload $w --steps 30 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $w', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'])"
  done
done; done 2>&1 | tee gpurun_out/r2u/ab.txt
