R=$GRAFT_REPO_ROOT
get() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'])"; }
cd $R
python -m pytest tests/test_kernels.py -m gpu -x -q -k "tile_walking or conv3x3" 2>&1 | tail -n 2
for i in 1 2; do
for p in 0 256 512 1024; do
DPP_C3_PERSIST=$p python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "f32 128 persist=$p"
done
for p in 0 256 512 1024 2048; do
DPP_C3_PERSIST=$p python bench.py --no-cpu-baseline --no-trainer --headline-only --size 256 --dtype bf16 --steps 30 --warmup 5 2>/dev/null | get "bf16 256 persist=$p"
done
done
