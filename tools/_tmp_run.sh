cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm.py tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/tmp/err.txt | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["config"]["final_cost"])' | sed "s/^/$name /"; }
b pre1 X=1
b pre0 DPP_EPILOGUE_PREFETCH=0
b pre1b X=1
b pre0b DPP_EPILOGUE_PREFETCH=0
