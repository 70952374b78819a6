mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_dense_oracle.py tests/test_gpu_abi.py -m gpu -q > gpurun_out/r2c/t.txt 2>&1; tail -4 gpurun_out/r2c/t.txt
for w in "cl41" "cl41 --flags 0x200" "cl41g1" "cl41g1 --flags 0x200" "cl41s" "cl41s --flags 0x200" "r8" "gp9f32" "gp5f64x" "gp5f64x --flags 0x200"; do
python bench.py --workload $w --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '%.4g' % d['value'], r['kernel'][:30], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'], 'ach=%.1f' % r['achieved'])"
done
