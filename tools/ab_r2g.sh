mkdir -p gpurun_out/r2g
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_abi.py tests/test_gpu_abi_c_host.py tests/test_wire_format.py -m gpu -q -x > gpurun_out/r2g/t.txt 2>&1; tail -15 gpurun_out/r2g/t.txt
for w in "cl41" "cl41 --flags 0x200" "cl41g1" "cl41g1 --flags 0x200" "cl41s" "cl41s --flags 0x200" "gp5f64x" "gp5f64x --flags 0x200" "gp6f32x" "gp6f32x --flags 0x200"; do
python bench.py --workload $w --steps 30 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '%.4g' % d['value'], r['kernel'][:30], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'], 'ach=%.1f' % r['achieved'])"
done 2>&1 | tee gpurun_out/r2g/ab.txt
