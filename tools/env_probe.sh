# Runtime-environment knobs against the bench step time:  gpurun -- 'bash tools/env_probe.sh'
# (kernel-argument placement, queue count, scratch / signal handling: all ROCr / CLR switches, no code change)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/env_probe
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # name, env assignments...
  name=$1; shift
  ms=$(env "$@" python $R/bench.py --no-cpu-baseline --steps 60 --warmup 10 2>$O/$name.err | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')
  echo "$name $* ms_per_step=$ms" | tee -a $O/env_probe.txt
}
: > $O/env_probe.txt
run base0 X=1
run kernarg_dev1 HIP_FORCE_DEV_KERNARG=1
run kernarg_dev0 HIP_FORCE_DEV_KERNARG=0
run fgs0 ROC_USE_FGS_KERNARG=0
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq8 GPU_MAX_HW_QUEUES=8
run base1 X=1
