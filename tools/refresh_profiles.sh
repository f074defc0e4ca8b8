# Regenerate the rocprofv3 evidence of the current round on the GPU box:  gpurun -- 'bash tools/refresh_profiles.sh r02'
# Outputs land in gpurun_out/<round>/ and are then copied to profiles/<round>_*.
set -x
ROUND=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# kernel trace of the measured configuration (bench defaults: native plan, two streams) -- kernel-trace only, no counters
rocprofv3 --kernel-trace -d /tmp/p1 -o run -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only --no-trainer > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name '*_results.db' | head -1) 28 > $O/kernel_stats_bench.txt 2>&1
# un-overlapped kernel durations, one line per (kernel, grid)
DPP_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/p2 -o run -- python $R/tools/step_profile.py 8 > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p2 -name '*_results.db' | head -1) 8 --by-grid > $O/kernel_stats_single_stream_by_grid.txt 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md, rocprofv3 PMC slots), no other trace domains
DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p3 -o run -- python $R/tools/step_profile.py 3 > /dev/null 2>&1
DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p4 -o run -- python $R/tools/step_profile.py 3 > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/p3 -name '*_results.db' | head -1) $(find /tmp/p4 -name '*_results.db' | head -1) 3 --json $O/hbm_traffic.json > $O/hbm_traffic.txt 2>&1
# the same for BASELINE config 5's step (256x256, bs128): float32, and bf16 (bf16 MFMA operands + bf16-stored activations and gradients)
for dt in f32 bf16; do
  DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p3_$dt -o run -- python $R/tools/step_profile.py 2 256 $dt > /dev/null 2>&1
  DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p4_$dt -o run -- python $R/tools/step_profile.py 2 256 $dt > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/p3_$dt -name '*_results.db' | head -1) $(find /tmp/p4_$dt -name '*_results.db' | head -1) 2 --params 69046100 --json $O/hbm_traffic_256_$dt.json > $O/hbm_traffic_256_$dt.txt 2>&1
done
# dynamic instruction mix per wave (SQ counters, two passes of 8; counters only + kernel trace)
DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --kernel-trace -d /tmp/p5 -o run -- python $R/tools/step_profile.py 3 > /dev/null 2>&1
DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/p6 -o run -- python $R/tools/step_profile.py 3 > /dev/null 2>&1
python $R/tools/inst_summary.py $(find /tmp/p5 -name '*_results.db' | head -1) $(find /tmp/p6 -name '*_results.db' | head -1) > $O/instruction_mix.txt 2>&1
# matrix-core busy cycles (north_star: "rocprof MFMA utilisation"): one more SQ pass, counters only + kernel trace
DPP_NO_SIDE_STREAM=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --kernel-trace -d /tmp/p7 -o run -- python $R/tools/step_profile.py 3 > /dev/null 2>&1
python $R/tools/mfma_busy_summary.py $(find /tmp/p7 -name '*_results.db' | head -1) --json $O/mfma_busy.json > $O/mfma_busy.txt 2>&1
# HBM traffic of the deterministic forward: the same two counter passes over tools/forward_profile.py, calibrated with
# the factors of the train-step pass above (the forward has no adam_kernel to calibrate on)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p3_fwd -o run -- python $R/tools/forward_profile.py 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p4_fwd -o run -- python $R/tools/forward_profile.py 3 > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/p3_fwd -name '*_results.db' | head -1) $(find /tmp/p4_fwd -name '*_results.db' | head -1) 6 --cal-from $O/hbm_traffic.json --json $O/hbm_traffic_forward.json > $O/hbm_traffic_forward.txt 2>&1
# the deterministic forward (computeOutput's device function): kernel trace by grid + the fused block's phase stamps
rocprofv3 --kernel-trace -d /tmp/p8 -o run -- python $R/tools/forward_profile.py 8 > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p8 -name '*_results.db' | head -1) 11 --by-grid > $O/forward_kernels_by_grid.txt 2>&1
python $R/tools/forward_profile.py 30 > $O/forward_only.txt 2>/dev/null
python $R/tools/forward_profile.py 30 256 >> $O/forward_only.txt 2>/dev/null
python $R/tools/forward_profile.py 30 256 bf16 >> $O/forward_only.txt 2>/dev/null
python $R/tools/wgrad3_micro.py > $O/wgrad3_micro.txt 2>/dev/null
python $R/tools/wgrad3_micro.py --precision 1 >> $O/wgrad3_micro.txt 2>/dev/null
DPP_WGRAD3_T=0 python $R/tools/wgrad3_micro.py >> $O/wgrad3_micro.txt 2>/dev/null
# the LDS-tiled / tile-walking 3x3 convolution alone at the shapes of the bs128 steps
python $R/tools/conv3_micro.py > $O/conv3_micro.txt 2>/dev/null
# the 256 x 256 bf16 step: un-overlapped kernel durations by grid
DPP_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/p9 -o run -- python $R/tools/step_profile.py 3 256 bf16 > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/p9 -name '*_results.db' | head -1) 3 --by-grid > $O/kernels_256_bf16_by_grid.txt 2>&1
python $R/tools/branch_probe.py > $O/branch_probe.txt 2>/dev/null
python $R/tools/launch_rate.py > $O/launch_rate.txt 2>/dev/null
# the other bench modes of BASELINE.json
python $R/bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16_128.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --size 256 --steps 20 --warmup 5 > $O/bench_f32_256.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --size 256 --steps 20 --warmup 5 --dtype bf16 > $O/bench_bf16_256.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --batch 256 > $O/bench_bs256.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --workload cascade --size 256 --steps 20 --warmup 5 > $O/bench_cascade_f32_256.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --workload cascade --size 256 --steps 20 --warmup 5 --dtype bf16 > $O/bench_cascade_bf16_256.json 2>/dev/null
python $R/tools/trainer_throughput.py 2>/dev/null | tail -1 > $O/trainer_throughput.txt
python $R/tools/augment_bench.py > $O/augment_bench.json 2> $O/augment_bench.err
python $R/tools/augment_bench.py --batch 4096 --iters 50 >> $O/augment_bench.json 2>> $O/augment_bench.err
# the bench line LAST, with this run's counter files in place (bench.py quotes roofline.traffic / roofline.mfma_busy only from files under
# profiles/ whose recorded kernel-source fingerprint is the one it runs on)
cp $O/hbm_traffic.json $R/profiles/${ROUND}_hbm_traffic.json
cp $O/mfma_busy.json $R/profiles/${ROUND}_mfma_busy.json
cp $O/hbm_traffic_256_bf16.json $R/profiles/${ROUND}_hbm_traffic_256_bf16.json
cp $O/hbm_traffic_forward.json $R/profiles/${ROUND}_hbm_traffic_forward.json
python $R/bench.py > $O/bench.json 2> $O/bench.err
ls -la $O
