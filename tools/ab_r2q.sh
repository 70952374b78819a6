mkdir -p gpurun_out/r2q
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "spinor" > gpurun_out/r2q/t.txt 2>&1; tail -5 gpurun_out/r2q/t.txt
for w in r12s gp11f32s gp10f32s gp9f32s r12ds gp11f64s gp10f64s gp9f64s gp8f32s gp8f64s; do
python bench.py --workload $w --steps 20 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['config']['batch_per_gpu'], '%.4g' % d['value'], r['kernel'][:40], '%.4f ms' % r['kernel_ms'], 'algGB/s=%.0f' % (r['bytes_per_item']*d['config']['batch_per_gpu']/(r['kernel_ms']*1e-3)*1e-9))"
done 2>&1 | tee gpurun_out/r2q/ab.txt
