mkdir -p gpurun_out/r2h
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r2h/t.txt 2>&1; tail -3 gpurun_out/r2h/t.txt
for rep in 1 2; do
for w in "cl41" "cl41 --flags 0x400" "cl41g1" "cl41g1 --flags 0x400" "cl41s" "cl41s --flags 0x400"; do
python bench.py --workload $w --steps 30 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'], 'ach=%.1f' % r['achieved'])"
done; done 2>&1 | tee gpurun_out/r2h/ab.txt
tools/pmc_pass.sh cl41 gaast_jit --workload cl41 2>&1 | tail -32
