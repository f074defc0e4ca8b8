R=$GRAFT_REPO_ROOT
get() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'])"; }
export DPP_EXPERIMENT=1
for i in 1 2; do
cd $R
python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "r6 default"
DPP_FUSE_LOSS=0 python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "r6 fuse_loss=0"
DPP_ADAM_TICKED=0 python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "r6 adam_ticked=0"
DPP_ADAM_TICKED=0 DPP_FUSE_LOSS=0 python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "r6 both=0"
DPP_ADAM_TICKED=0 DPP_FUSE_LOSS=0 DPP_WGRAD3_T=0 python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "r6 both=0 wgrad3_t=0"
DPP_ADAM_TICKED=0 DPP_FUSE_LOSS=0 python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 --no-augment 2>/dev/null | get "r6 both=0 no-augment"
(cd $R/.ab_r05 && python bench.py --no-cpu-baseline --no-trainer --steps 200 --warmup 20 2>/dev/null | get "r5")
(cd $R/.ab_r05 && python bench.py --no-cpu-baseline --no-trainer --steps 200 --warmup 20 --no-augment 2>/dev/null | get "r5 no-augment")
done
