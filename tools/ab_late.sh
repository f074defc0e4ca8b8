# Same-box A/B of this tree against the last commit (a git worktree of HEAD built under .ab_head/; boxes of the pool differ by 2-3 % from
# one another, so only alternating runs on ONE box compare builds):
#   git worktree add -f .ab_head HEAD && make -C .ab_head/deep-prior-pp_amd/csrc -j6 hip && gpurun -- 'bash tools/ab_late.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
get() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], 'ms', j['value'], 'crops/s', 'fwd', (j.get('forward_only') or {}).get('ms_per_batch'))"; }
run() { (cd $1 && python bench.py --no-cpu-baseline --no-trainer --headline-only "${@:3}" 2>/dev/null | get "$2"); }
{
echo "same-box A/B, alternating runs: head = $(cd $R/.ab_head && git rev-parse --short HEAD 2>/dev/null) (.ab_head), new = this tree ($(cd $R && python -c 'import bench; print(bench.csrc_sha16())'))"
for i in 1 2 3; do
  run $R/.ab_head "head f32 128x128 bs128 (200 steps)" --steps 200 --warmup 20
  run $R          "new  f32 128x128 bs128 (200 steps)" --steps 200 --warmup 20
done
for i in 1 2; do
  run $R/.ab_head "head bf16 256x256 bs128 (30 steps)" --size 256 --dtype bf16 --steps 30 --warmup 5
  run $R          "new  bf16 256x256 bs128 (30 steps)" --size 256 --dtype bf16 --steps 30 --warmup 5
  run $R/.ab_head "head bf16 128x128 bs128 (100 steps)" --dtype bf16 --steps 100 --warmup 10
  run $R          "new  bf16 128x128 bs128 (100 steps)" --dtype bf16 --steps 100 --warmup 10
  run $R/.ab_head "head f32 256x256 bs128 (30 steps)" --size 256 --steps 30 --warmup 5
  run $R          "new  f32 256x256 bs128 (30 steps)" --size 256 --steps 30 --warmup 5
done
} > $O/ab_late.txt 2>&1
cat $O/ab_late.txt
