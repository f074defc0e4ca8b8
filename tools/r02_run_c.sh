# GPU run C of round 2: tests, in-kernel phase profile, GPU-side micro-benchmarks, bf16 FC1 kernel parameters
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=6 > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.txt
timeout 300 python tools/phase_profile.py > $O/phase_profile.txt 2>&1; cat $O/phase_profile.txt
timeout 300 python tools/gemm_micro.py floor > $O/micro_floor.txt 2>&1; cat $O/micro_floor.txt
timeout 300 python tools/gemm_micro.py feats > $O/micro_feats.txt 2>&1; cat $O/micro_feats.txt
timeout 300 python tools/gemm_micro.py conv3 > $O/micro_conv3.txt 2>&1; cat $O/micro_conv3.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --size 256 --dtype bf16"
for kc in 32 64; do for sl in 16 32 64; do echo "== 256 bf16 kchunk $kc slices $sl"; DPP_FC1_KCHUNK=$kc DPP_FC1_SLICES=$sl $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done; done
