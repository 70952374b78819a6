mkdir -p gpurun_out/r2e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_dense_oracle.py -m gpu -q -k "dense or mfma or n8 or n1 or r8 or r12 or R8 or R12 or basis" > gpurun_out/r2e/t.txt 2>&1; tail -4 gpurun_out/r2e/t.txt
for w in r12 r8 gp9f32 gp10f32 gp11f32 gp13f32; do
python bench.py --workload $w --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '%.4g' % d['value'], r['kernel'][:30], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'], 'ach=%.1f' % r['achieved'])"
done
tools/pmc_pass.sh r8 k_gp_mfma16 --workload r8 2>&1 | tail -32
