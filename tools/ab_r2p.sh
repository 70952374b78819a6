mkdir -p gpurun_out/r2p
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_abi.py tests/test_gpu_dense_oracle.py -m gpu -q -x > gpurun_out/r2p/t.txt 2>&1; tail -3 gpurun_out/r2p/t.txt
for w in gp6f64x gp7f64x r8x gp9f64x gp10f64x r12x gp8f32x gp12f32x; do
python bench.py --workload $w --steps 10 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['config']['batch_per_gpu'], '%.4g' % d['value'], r['kernel'][:36], '%.4f ms' % r['kernel_ms'])"
done 2>&1 | tee gpurun_out/r2p/ab.txt
