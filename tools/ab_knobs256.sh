R=$GRAFT_REPO_ROOT
get() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'])"; }
cd $R
export DPP_EXPERIMENT=1
for i in 1 2; do
python bench.py --no-cpu-baseline --no-trainer --headline-only --size 256 --dtype bf16 --steps 30 --warmup 5 2>/dev/null | get "bf16 256 default"
DPP_EARLY_ADAM=1 python bench.py --no-cpu-baseline --no-trainer --headline-only --size 256 --dtype bf16 --steps 30 --warmup 5 2>/dev/null | get "bf16 256 early_adam"
DPP_WGRAD_STREAM_STAGES=123 python bench.py --no-cpu-baseline --no-trainer --headline-only --size 256 --dtype bf16 --steps 30 --warmup 5 2>/dev/null | get "bf16 256 wgrad_stream stages 123"
DPP_BN_BWD_FUSE_NB=256 python bench.py --no-cpu-baseline --no-trainer --headline-only --size 256 --dtype bf16 --steps 30 --warmup 5 2>/dev/null | get "bf16 256 bn_bwd_fuse 256"
python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "f32 128 default"
DPP_EARLY_ADAM=1 python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "f32 128 early_adam"
done
