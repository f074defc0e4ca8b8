# Ablation: the bench step with one kernel family removed (results are wrong; only the time means something): how much of the
# step each family is worth on the critical path, as opposed to its summed durations.   gpurun -- 'bash tools/whatif.sh'
# WHATIF_ARGS='--size 256' (any bench.py flags) selects another workload; WHATIF_ONLY='base no_wgrad_all' a subset of the rows.
export DPP_EXPERIMENT=1      # the engine reads its experiment knobs only with this set (hipdp/engine.py: knob)
cd $GRAFT_REPO_ROOT
O=gpurun_out/whatif; mkdir -p $O
b() { name=$1; shift; if [ -n "$WHATIF_ONLY" ] && ! echo " $WHATIF_ONLY " | grep -q " $name "; then return; fi; env "$@" python bench.py --allow-ablation --no-cpu-baseline --steps 60 --warmup 10 $WHATIF_ARGS 2>$O/$name.err | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"])' | sed "s/^/$name $* /" | tee -a $O/whatif.txt; }
: > $O/whatif.txt
b base X=1
b no_wgrad3 DPP_WHATIF_SKIP=conv3x3_wgrad
b no_wgrad1 DPP_WHATIF_SKIP=wgrad1x1
b no_wgrad_all DPP_WHATIF_SKIP=conv3x3_wgrad,wgrad1x1,fc_wgrad,stem_wgrad
b no_bn_finalize DPP_WHATIF_SKIP=bn_finalize
b no_bn_bwd_finalize DPP_WHATIF_SKIP=bn_bwd_finalize
b no_bn_bwd_apply DPP_WHATIF_SKIP=bn_bwd_apply
b no_dgrad3 DPP_WHATIF_SKIP=dgrad3x3
b no_conv3_fwd DPP_WHATIF_SKIP=conv3x3_
b no_dgrad1 DPP_WHATIF_SKIP=dgrad1x1
b no_conv1_fwd DPP_WHATIF_SKIP=conv1x1
b no_fc DPP_WHATIF_SKIP=fc_
b no_adam DPP_WHATIF_SKIP=adam
b no_reduce DPP_WHATIF_SKIP=reduce_multi
b no_stem DPP_WHATIF_SKIP=stem_
b no_augment DPP_WHATIF_SKIP=augment
b no_wtrans DPP_WHATIF_SKIP=conv3x3_wtrans,fill_zero
b base2 X=1
