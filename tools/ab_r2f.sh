mkdir -p gpurun_out/r2f
run() { # label, lib, workload
python bench.py --workload $3 --steps 30 --no-cpu-baseline --no-alt --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$3', '%.4g' % d['value'], '%.4f ms' % r['kernel_ms'], 'frac=%.3f' % r['frac'])"
}
for rep in 1 2; do
for v in A B C D; do
  if [ $v = A ]; then unset GAAST_HIP_LIB; else export GAAST_HIP_LIB=$PWD/gaast_amd/lib_ab/$v/libgaast_hip.so; fi
  for w in r8 gp9f32 r12; do run $v x $w; done
done
done 2>&1 | tee gpurun_out/r2f/ab.txt
