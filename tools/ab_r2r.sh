mkdir -p gpurun_out/r2r
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "spinor" > gpurun_out/r2r/t.txt 2>&1; tail -5 gpurun_out/r2r/t.txt
for w in r66s r12s; do
python bench.py --workload $w --steps 20 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['config']['batch_per_gpu'], '%.4g' % d['value'], r['kernel'][:40], '%.4f ms' % r['kernel_ms'], 'algGB/s=%.0f' % (r['bytes_per_item']*d['config']['batch_per_gpu']/(r['kernel_ms']*1e-3)*1e-9))"
done 2>&1 | tee gpurun_out/r2r/ab.txt
