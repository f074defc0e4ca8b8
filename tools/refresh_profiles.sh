set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace -d /tmp/p1 -o run -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch eager > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name '*_results.db' | head -1) 28 > $O/kernel_stats_bench.txt 2>&1
DPP_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/p2 -o run -- python $R/tools/step_profile.py 8 > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p2 -name '*_results.db' | head -1) 8 --by-grid > $O/kernel_stats_single_stream_by_grid.txt 2>&1
DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p3 -o run -- python $R/tools/step_profile.py 3 > /dev/null 2>&1
DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p4 -o run -- python $R/tools/step_profile.py 3 > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/p3 -name '*_results.db' | head -1) $(find /tmp/p4 -name '*_results.db' | head -1) 3 --json $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
ls -la $O
