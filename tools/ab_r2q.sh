mkdir -p gpurun_out/r2q
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "spinor" > gpurun_out/r2q/t.txt 2>&1; tail -5 gpurun_out/r2q/t.txt
for rep in 1 2; do for w in r12s gp11f32s; do
python bench.py --workload $w --steps 20 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '%.4g' % d['value'], r['kernel'][:40], '%.4f ms' % r['kernel_ms'])"
done; done 2>&1 | tee gpurun_out/r2q/ab.txt
