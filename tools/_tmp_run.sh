cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels.py tests/test_gemm.py tests/test_augment.py -m gpu -x -q 2>&1 | tail -2
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 60 --warmup 10 $EXTRA 2>/tmp/err.txt | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["config"]["final_cost"])' | sed "s/^/$name /"; }
b new X=1
b new2 X=1
b new3 X=1
