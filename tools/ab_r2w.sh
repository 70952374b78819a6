# k_gp_mfma16 in image-pair form: parity tests of everything dense, then r8 / gp9f32 (A = default, P1 = priority ramp)
mkdir -p gpurun_out/r2w
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_dense_oracle.py tests/test_gpu_abi.py -m gpu -q -x -k "dense or mfma or n8 or n9 or r8 or R8 or basis or gather or wrap" > gpurun_out/r2w/t.txt 2>&1; tail -5 gpurun_out/r2w/t.txt
for rep in 1 2; do
for v in A P1; do
  if [ $v = A ]; then unset GAAST_HIP_LIB; else export GAAST_HIP_LIB=$PWD/gaast_amd/lib_ab/$v/libgaast_hip.so; fi
  for w in r8 gp9f32; do
python bench.py --workload $w --steps 30 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $w', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'])"
  done
done; done 2>&1 | tee gpurun_out/r2w/ab.txt
