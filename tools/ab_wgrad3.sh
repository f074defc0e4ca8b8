#!/bin/bash
# A/B of the 3x3 filter-gradient kernels inside the two-stream step: which channel counts take dpp_wgrad3_stream
mkdir -p gpurun_out/r03
export DPP_EXPERIMENT=1
run() {
  python bench.py --steps 300 --warmup 30 --allow-ablation --no-cpu-baseline "$@" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['value'], d['roofline']['frac'])
"
}
for rep in 1 2; do
  for cfg in none 64 32,64 16,32,64; do
    echo "== 128x128 stream C = $cfg: $(DPP_WGRAD3_STREAM_C=${cfg/none/0} run)"
  done
  for cfg in none 64 32,64 16,32,64; do
    echo "== 256x256 stream C = $cfg: $(DPP_WGRAD3_STREAM_C=${cfg/none/0} run --size 256 --steps 100)"
  done
done
