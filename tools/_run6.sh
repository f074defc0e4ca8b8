mkdir -p gpurun_out/r03
python -m pytest tests/test_gemm.py tests/test_full_size.py -m gpu -q -k "wgrad_stream or gradients or instantiation" 2>&1 | tail -4
python bench.py --no-cpu-baseline --profile-ops > gpurun_out/r03/bench1.json 2> gpurun_out/r03/bench1_ops.txt
cut -c1-200 gpurun_out/r03/bench1.json
grep -E "wgrad1x1|^gemm_mfma|^conv3x3_wgrad|sum of" gpurun_out/r03/bench1_ops.txt | awk '{print $3, $5, $6, $7}' | sort | uniq -c | sort -k3 -n | tail -40
export DPP_EXPERIMENT=1
for r in 32 64 128 256; do echo rpw $r; DPP_WGRAD_STREAM_RPW=$r python bench.py --no-cpu-baseline --steps 60 --warmup 10 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])'; done
echo off; DPP_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --steps 60 --warmup 10 | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])'
