# k_gp_mfma32p (image-pair form): dense parity tests, then r12 / gp10f32 / gp11f32 / gp13f32
mkdir -p gpurun_out/r2x
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_dense_oracle.py tests/test_gpu_abi.py -m gpu -q -x > gpurun_out/r2x/t.txt 2>&1; tail -5 gpurun_out/r2x/t.txt
for rep in 1 2; do
  for w in r12 gp10f32 gp11f32 gp13f32; do
python bench.py --workload $w --steps 20 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'])"
  done
done 2>&1 | tee gpurun_out/r2x/ab.txt
