cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm.py -m gpu -x -q -k ksplit 2>&1 | tail -2
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 60 --warmup 10 $EXTRA 2>/tmp/err.txt | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["config"]["final_cost"])' | sed "s/^/$name /"; }
b remap1 X=1
b remap0 DPP_KSPLIT_XCD_REMAP=0
b remap1b X=1
b remap0b DPP_KSPLIT_XCD_REMAP=0
