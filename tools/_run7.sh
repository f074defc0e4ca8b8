mkdir -p gpurun_out/r03
export DPP_EXPERIMENT=1
echo "== tail probe, stream wgrad on"; python tools/tail_probe.py 2>/dev/null | grep backward | tail -2
echo "== tail probe, stream wgrad off"; DPP_WGRAD_STREAM=0 python tools/tail_probe.py 2>/dev/null | grep backward | tail -2
b() { name=$1; shift; env "$@" python bench.py --allow-ablation --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"])' | sed "s/^/$name $* /"; }
b base X=1
b no_wgrad1 DPP_WHATIF_SKIP=wgrad1x1
b no_wgrad3 DPP_WHATIF_SKIP=conv3x3_wgrad
b no_wgrad_all DPP_WHATIF_SKIP=conv3x3_wgrad,wgrad1x1,fc_wgrad,stem_wgrad
b off_base DPP_WGRAD_STREAM=0
b off_no_wgrad1 DPP_WGRAD_STREAM=0 DPP_WHATIF_SKIP=wgrad1x1
b single_stream DPP_NO_SIDE_STREAM=1
b off_single_stream DPP_NO_SIDE_STREAM=1 DPP_WGRAD_STREAM=0
