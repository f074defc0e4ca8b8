# BatchNorm backward of the small maps as finalize + apply (DPP_BN_BWD_FUSE_NB=0) against dpp_bn_bwd_finalize_apply (128 / 256 blocks of sums),
# single stream, launch by launch:   gpurun -- 'bash tools/fa_ab.sh'   -> gpurun_out/fa/seq_fa<NB>.txt (tools/prof_sequence.py)
cd /tmp && export TMPDIR=/tmp DPP_EXPERIMENT=1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fa; mkdir -p $O
for L in 0 128 256; do
  DPP_BN_BWD_FUSE_NB=$L DPP_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace -d /tmp/pf_$L -o run -- python $R/tools/step_profile.py 6 > /dev/null 2>&1
  python $R/tools/prof_sequence.py $(find /tmp/pf_$L -name "*_results.db" | head -1) 2 > $O/seq_fa$L.txt 2>&1
  head -1 $O/seq_fa$L.txt
done
