# GPU run B of round 2: tests, FC1 kernel variants, config-5 bench modes, launch rate, single-stream kernel trace
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=6 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10"
echo "== default (FC1 stream f32)"; $B 2>/dev/null | tee $O/bench_stream.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
echo "== FC1 generic"; DPP_FC1_STREAM=0 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
echo "== FC1 stream kchunk32"; DPP_FC1_KCHUNK=32 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
echo "== FC1 stream slices16"; DPP_FC1_SLICES=16 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
echo "== FC1 stream slices64"; DPP_FC1_SLICES=64 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
echo "== bf16 128"; $B --dtype bf16 2>$O/bf16_128.err | tee $O/bench_bf16_128.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('bf16_forward_error_mm_vs_fp32'))"
echo "== 256 f32"; $B --size 256 --steps 20 --warmup 5 2>$O/f32_256.err | tee $O/bench_f32_256.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
echo "== 256 bf16"; $B --size 256 --steps 20 --warmup 5 --dtype bf16 2>$O/bf16_256.err | tee $O/bench_bf16_256.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('bf16_forward_error_mm_vs_fp32'))"
echo "== 256 f32 generic FC1"; DPP_FC1_STREAM=0 $B --size 256 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
timeout 400 python tools/launch_rate.py --steps 30 > $O/launch_rate.txt 2>&1; cat $O/launch_rate.txt
cd /tmp && export TMPDIR=/tmp
DPP_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/p2 -o run -- python $R/tools/step_profile.py 8 > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p2 -name '*_results.db' | head -1) 8 --by-grid > $O/kernel_stats_single_stream_by_grid.txt 2>&1
head -30 $O/kernel_stats_single_stream_by_grid.txt
