# The gradient branch spread over several side streams (DPP_SIDE_STREAMS) against the bench step time:  gpurun -- 'bash tools/exp_side_streams.sh'
export DPP_EXPERIMENT=1      # the engine reads its experiment knobs only with this set (hipdp/engine.py: knob)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
python -m pytest tests/test_data_parallel.py -m gpu -x -q > $O/pytest_dp.txt 2>&1; tail -3 $O/pytest_dp.txt
b() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>$O/$name.err | tee $O/$name.json | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["config"]["final_cost"])' | sed "s/^/$name /"; }
b s1 DPP_SIDE_STREAMS=1
b s2 DPP_SIDE_STREAMS=2
b s3 DPP_SIDE_STREAMS=3
b s4 DPP_SIDE_STREAMS=4
b s3q8 DPP_SIDE_STREAMS=3 GPU_MAX_HW_QUEUES=8
b s4q8 DPP_SIDE_STREAMS=4 GPU_MAX_HW_QUEUES=8
b s6q8 DPP_SIDE_STREAMS=6 GPU_MAX_HW_QUEUES=8
b s1b DPP_SIDE_STREAMS=1
DPP_SIDE_STREAMS=1 python tools/tail_probe.py 2>&1 | grep backward | tail -2 | sed "s/^/S1 /"
