R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
timeout 300 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; cat $O/phase_profile.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 10"
echo "== deep ring K>=256 (default)"; $B 2>/dev/null | tee $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
echo "== deep ring off"; DPP_GEMM_DEEP_MIN_K=0 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
echo "== deep ring K>=128"; DPP_GEMM_DEEP_MIN_K=128 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
