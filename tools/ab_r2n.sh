mkdir -p gpurun_out/r2n
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_abi.py -m gpu -q -x > gpurun_out/r2n/t.txt 2>&1; tail -3 gpurun_out/r2n/t.txt
for rep in 1 2; do
for w in "gp5f64x" "gp5f64x --flags 0x200" "gp6f32x" "gp6f32x --flags 0x200" "gp5f32x" "gp5f32x --flags 0x200" "cl41" "cl41 --flags 0x200"; do
python bench.py --workload $w --batch 4194304 --steps 20 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '%.4g' % d['value'], r['kernel'][:28], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'], 'ach=%.1f' % r['achieved'])"
done; done 2>&1 | tee gpurun_out/r2n/ab.txt
