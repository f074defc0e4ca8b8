# Tail-shortening knobs against the bench step time:  gpurun -- 'bash tools/exp_tail.sh'
export DPP_EXPERIMENT=1      # the engine reads its experiment knobs only with this set (hipdp/engine.py: knob)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
python -m pytest tests/test_engine.py -m gpu -x -q -k "early" > $O/pytest_early.txt 2>&1; tail -3 $O/pytest_early.txt
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>$O/$name.err | tee $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["config"]["final_cost"], d["config"]["launches"])' | sed "s/^/$name /"; }
b base DPP_EARLY_ADAM=0 DPP_EARLY_REDUCE_MB=0
b adam DPP_EARLY_ADAM=1 DPP_EARLY_REDUCE_MB=0
b red16 DPP_EARLY_ADAM=0 DPP_EARLY_REDUCE_MB=16
b both8 DPP_EARLY_ADAM=1 DPP_EARLY_REDUCE_MB=8
b both16 DPP_EARLY_ADAM=1 DPP_EARLY_REDUCE_MB=16
b both32 DPP_EARLY_ADAM=1 DPP_EARLY_REDUCE_MB=32
b both64 DPP_EARLY_ADAM=1 DPP_EARLY_REDUCE_MB=64
b base2 DPP_EARLY_ADAM=0 DPP_EARLY_REDUCE_MB=0
