#!/bin/bash
# Samples the GPU's clock and power (rocm-smi, read-only) while bench.py runs one workload for a few seconds:
#   tools/clocks_under_load.sh <workload> [steps]      -> gpurun_out/clocks/<workload>.txt
w=$1; steps=${2:-3000}
mkdir -p gpurun_out/clocks
out=gpurun_out/clocks/$w.txt
python bench.py --workload $w --steps $steps --warmup 5 --no-cpu-baseline --no-alt --no-latency --no-configs > gpurun_out/clocks/$w.json 2>/dev/null &
pid=$!
sleep 6     # import + setup + the transient of the first launches
: > $out
for i in 1 2 3 4 5; do
  if kill -0 $pid 2>/dev/null; then
    echo "== sample $i (bench running)" >> $out
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" >> $out
    sleep 1
  fi
done
wait $pid
python - "$w" <<'PY' >> $out
import json, sys
b = json.load(open(f"gpurun_out/clocks/{sys.argv[1]}.json")); r = b["roofline"]
print("bench:", "%.4g" % b["value"], b["unit"], "kernel_ms %.4f" % r["kernel_ms"], "frac %.3f" % r["frac"])
PY
cat $out
