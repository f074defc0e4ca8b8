# Round-5 same-box A/B runs of the bs128 train step (200 timed steps each, no CPU legs):  gpurun -- 'bash tools/ab_r05.sh'
# Each line: label, ms_per_step, crops/s.  Knobs are experiment-only (DPP_EXPERIMENT=1 is stamped into config.knobs by bench.py).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_r05
mkdir -p $O
run() {   # label, env..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env DPP_EXPERIMENT=1 "${envs[@]}" python $R/bench.py --no-cpu-baseline --no-trainer --steps 200 --warmup 20 "$@" > $O/$label.json 2> $O/$label.err
  python - "$label" "$O/$label.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][0])
    print('%-34s %8.4f ms  %9.1f crops/s  hip_event %s' % (sys.argv[1], d['ms_per_step'], d['value'], d['config'].get('hip_event_ms_per_step')))
except Exception as e:
    print('%-34s FAILED %s' % (sys.argv[1], e))
PY
}
if [ "${AB_SET:-1}" = "1" ]; then
run base --
run base_again --
run wgrad3_slices512 DPP_WGRAD3_STREAM_SLICES=512 --
run wgrad_blocks128 DPP_WGRAD_TARGET_BLOCKS=128 --
run both_fewer_slices DPP_WGRAD3_STREAM_SLICES=512 DPP_WGRAD_TARGET_BLOCKS=128 --
run c3_min_wgs512 DPP_C3_MIN_WGS=512 --
run c3_min_wgs256 DPP_C3_MIN_WGS=256 --
run bf16_256_base -- --size 256 --dtype bf16 --steps 60 --warmup 10
run bf16_256_lazy3 DPP_LAZY_BN_BWD=3 -- --size 256 --dtype bf16 --steps 60 --warmup 10
run bf16_256_lazy1 DPP_LAZY_BN_BWD=1 -- --size 256 --dtype bf16 --steps 60 --warmup 10
run bf16_256_slices DPP_WGRAD3_STREAM_SLICES=512 DPP_WGRAD_TARGET_BLOCKS=128 -- --size 256 --dtype bf16 --steps 60 --warmup 10
run bf16_256_c3_512 DPP_C3_MIN_WGS=512 -- --size 256 --dtype bf16 --steps 60 --warmup 10
elif [ "${AB_SET}" = "4" ]; then
# fourth set: BASELINE config 5 (256x256 bf16) with the row-stream filter-gradient kernels on more layers (existing knobs)
run s4_bf16_256_base -- --size 256 --dtype bf16 --steps 60 --warmup 10
run s4_bf16_256_w3_all DPP_WGRAD3_STREAM_C=16,32,64 -- --size 256 --dtype bf16 --steps 60 --warmup 10
run s4_bf16_256_w3_32 DPP_WGRAD3_STREAM_C=32,64 -- --size 256 --dtype bf16 --steps 60 --warmup 10
run s4_bf16_256_w1_12 DPP_WGRAD_STREAM_STAGES=12 -- --size 256 --dtype bf16 --steps 60 --warmup 10
run s4_bf16_256_w1_123 DPP_WGRAD_STREAM_STAGES=123 -- --size 256 --dtype bf16 --steps 60 --warmup 10
run s4_bf16_256_w3_all_w1_123 DPP_WGRAD3_STREAM_C=16,32,64 DPP_WGRAD_STREAM_STAGES=123 -- --size 256 --dtype bf16 --steps 60 --warmup 10
run s4_128_w3_all DPP_WGRAD3_STREAM_C=16,32,64 --
run s4_128_base --
else
# second set (AB_SET=2; the third and fifth sets of profiles/r05_ab.txt measured kernels / plans that were removed again: conv3x3_db_kernel,
# the late FC1 update): the new defaults (column-tile rule 512), MORE partial slices, the lazy BatchNorm-backward operand at 256x256 (float32: the
# two-tensor operand is not built for bf16-stored tensors)
run s2_base --
run s2_wgrad_blocks512 DPP_WGRAD_TARGET_BLOCKS=512 --
run s2_wgrad3_slices2048 DPP_WGRAD3_STREAM_SLICES=2048 --
run s2_f32_256_base -- --size 256 --steps 60 --warmup 10
run s2_f32_256_lazy3 DPP_LAZY_BN_BWD=3 -- --size 256 --steps 60 --warmup 10
run s2_bf16_256_base -- --size 256 --dtype bf16 --steps 60 --warmup 10
run s2_bf16_256_blocks512 DPP_WGRAD_TARGET_BLOCKS=512 -- --size 256 --dtype bf16 --steps 60 --warmup 10
fi
if [ "${AB_SET:-1}" != "1" ]; then exit 0; fi
# where the class API's time goes: the trainer loop under cProfile
python - <<'PY' > $O/trainer_cprofile.txt 2>&1
import cProfile, pstats, io, runpy, sys, os
sys.argv = ['trainer_throughput.py']
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tools', 'trainer_throughput.py'), run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25)
print(s.getvalue()[:6000])
PY
