# GPU run D of round 2: effect of the deferred prologue / residual prefetch / all-weights 3x3 staging
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 300 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; cat $O/phase_profile.txt
timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tee $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'], d['roofline'])"
DPP_NO_SIDE_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single-stream', d['ms_per_step'], d['value'])"
timeout 300 python tools/gemm_micro.py conv3 > $O/micro_conv3.txt 2>&1; cat $O/micro_conv3.txt
