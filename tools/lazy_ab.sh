# Launch-by-launch comparison of the BatchNorm-backward plans (engine.LAZY_BN_BWD 0 | 1 | 3) on the bs128 step, single stream:
#   gpurun -- 'bash tools/lazy_ab.sh'   -> gpurun_out/lazy/seq_lazy<L>.txt (tools/prof_sequence.py)
cd /tmp && export TMPDIR=/tmp DPP_EXPERIMENT=1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lazy; mkdir -p $O
for L in 0 1 3; do
  DPP_LAZY_BN_BWD=$L DPP_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/pl_$L -o run -- python $R/tools/step_profile.py 6 > /dev/null 2>&1
  python $R/tools/prof_sequence.py $(find /tmp/pl_$L -name "*_results.db" | head -1) 2 > $O/seq_lazy$L.txt 2>&1
  head -1 $O/seq_lazy$L.txt
done
