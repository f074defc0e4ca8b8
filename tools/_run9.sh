export DPP_EXPERIMENT=1
b() { name=$1; shift; env "$@" python bench.py --allow-ablation --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"])' | sed "s/^/$name /"; }
for i in 1 2 3 4; do
b on X=1
b off DPP_WGRAD_STREAM=0
b s12 DPP_WGRAD_STREAM_STAGES=12 DPP_WGRAD_STREAM_RPW=256,256,128
done
