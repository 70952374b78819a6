mkdir -p gpurun_out/r2s
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_dense_oracle.py -m gpu -q -k "dense or mfma or n8 or r8 or R8" > gpurun_out/r2s/t.txt 2>&1; tail -4 gpurun_out/r2s/t.txt
for rep in 1 2; do
for v in A N0; do
  if [ $v = A ]; then unset GAAST_HIP_LIB; else export GAAST_HIP_LIB=$PWD/gaast_amd/lib_ab/$v/libgaast_hip.so; fi
  for w in r8 gp9f32; do
python bench.py --workload $w --steps 30 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $w', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'])"
  done
done; done 2>&1 | tee gpurun_out/r2s/ab.txt
