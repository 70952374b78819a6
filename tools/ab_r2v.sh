# timing-only decomposition of k_gp_mfma16 (variants built from a scratch copy with parts switched off; results wrong by design)
mkdir -p gpurun_out/r2v
for rep in 1 2; do
for v in A X1 X2 X3 X4 X5; do
  if [ $v = A ]; then unset GAAST_HIP_LIB; else export GAAST_HIP_LIB=$PWD/gaast_amd/lib_ab/$v/libgaast_hip.so; fi
  for w in r8; do
python bench.py --workload $w --steps 30 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $w', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'])"
  done
done; done 2>&1 | tee gpurun_out/r2v/ab.txt
