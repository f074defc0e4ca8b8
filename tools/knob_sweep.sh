#!/bin/bash
# Engine knob sweep on one workload:   gpurun -- 'SWEEP_ARGS="--size 256" bash tools/knob_sweep.sh "DPP_KSPLIT_MAX_M=32768" "DPP_STREAM16=3" ...'
# every argument is one configuration (space-separated VAR=value pairs); the unmodified build runs first and last.
export DPP_EXPERIMENT=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep; mkdir -p $O
b() { env $1 python bench.py --allow-ablation --no-cpu-baseline --steps ${SWEEP_STEPS:-100} --warmup 10 $SWEEP_ARGS 2>>$O/err.txt | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"])' | sed "s/^/$1 /" | tee -a $O/sweep.txt; }
: > $O/sweep.txt
b X=base
for cfg in "$@"; do b "$cfg"; done
b X=base2
