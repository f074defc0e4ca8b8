# Same-box A/B of this tree against the round-5 tree (a git worktree of commit ae867d8 built under .ab_r05/; boxes of the pool differ by
# 2-3 % from one another, so only alternating runs on ONE box compare builds):
#   git worktree add -f .ab_r05 ae867d8 && make -C .ab_r05/deep-prior-pp_amd/csrc hip && gpurun -- 'bash tools/ab_r06.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
get() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['ms_per_step'], 'ms', j['value'], 'crops/s', 'fwd', (j.get('forward_only') or {}).get('ms_per_batch'))"; }
{
echo "same-box A/B, alternating runs: r5 = commit ae867d8 (.ab_r05), r6 = this tree ($(cd $R && python -c 'import bench; print(bench.csrc_sha16())'))"
for i in 1 2 3; do
  (cd $R/.ab_r05 && python bench.py --no-cpu-baseline --no-trainer --steps 200 --warmup 20 2>/dev/null | get "r5 f32 128x128 bs128 (200 steps)")
  (cd $R && python bench.py --no-cpu-baseline --no-trainer --headline-only --steps 200 --warmup 20 2>/dev/null | get "r6 f32 128x128 bs128 (200 steps)")
done
for i in 1 2; do
  (cd $R/.ab_r05 && python bench.py --no-cpu-baseline --no-trainer --size 256 --dtype bf16 --steps 30 --warmup 5 2>/dev/null | get "r5 bf16 256x256 bs128 (30 steps)")
  (cd $R && python bench.py --no-cpu-baseline --no-trainer --headline-only --size 256 --dtype bf16 --steps 30 --warmup 5 2>/dev/null | get "r6 bf16 256x256 bs128 (30 steps)")
  (cd $R/.ab_r05 && python bench.py --no-cpu-baseline --no-trainer --dtype bf16 --steps 100 --warmup 10 2>/dev/null | get "r5 bf16 128x128 bs128 (100 steps)")
  (cd $R && python bench.py --no-cpu-baseline --no-trainer --headline-only --dtype bf16 --steps 100 --warmup 10 2>/dev/null | get "r6 bf16 128x128 bs128 (100 steps)")
  (cd $R/.ab_r05 && python bench.py --no-cpu-baseline --no-trainer --size 256 --steps 30 --warmup 5 2>/dev/null | get "r5 f32 256x256 bs128 (30 steps)")
  (cd $R && python bench.py --no-cpu-baseline --no-trainer --headline-only --size 256 --steps 30 --warmup 5 2>/dev/null | get "r6 f32 256x256 bs128 (30 steps)")
done
} > $O/ab_vs_r05.txt 2>&1
cat $O/ab_vs_r05.txt
