mkdir -p gpurun_out/r2t
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_dense_oracle.py -m gpu -q -k "dense or mfma or n1 or r12 or R12 or basis" > gpurun_out/r2t/t.txt 2>&1; tail -4 gpurun_out/r2t/t.txt
for rep in 1 2 3; do
for v in A L0; do
  if [ $v = A ]; then unset GAAST_HIP_LIB; else export GAAST_HIP_LIB=$PWD/gaast_amd/lib_ab/$v/libgaast_hip.so; fi
  for w in r12 gp10f32; do
python bench.py --workload $w --steps 30 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $w', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'])"
  done
done; done 2>&1 | tee gpurun_out/r2t/ab.txt
