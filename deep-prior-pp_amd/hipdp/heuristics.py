"""
hipdp.heuristics -- which kernel, which tile, how many slices: the choices hipdp.engine makes while it lowers a net to launch plans, and
the experiment knobs behind them.

Every rule here was set by a measurement on the MI355X; the comment next to it names the profile (profiles/) or the section of DESIGN.md
that holds the numbers.  The product ignores the DPP_* environment variables below unless DPP_EXPERIMENT=1 is set as well (`knob`): the
tools/ scripts set it, bench.py stamps both into its output line (config.knobs).  hipdp.engine reads these names through the module
(`hz.NAME`), so a test can monkeypatch one of them for the engines it compiles.
"""
import os

def knob(name, default):
    """Experiment knobs (tile / kernel-variant choices the measurements of DESIGN.md section 5 were made with, and the ablation
    DPP_WHATIF_SKIP of tools/whatif.sh).  The product ignores them: they are read only when DPP_EXPERIMENT=1 is set as well, which
    the tools/ scripts do and bench.py stamps into its output line (config.knobs)."""
    if os.environ.get('DPP_EXPERIMENT', '0') != '1':
        return default
    return os.environ.get(name, default)


BN_RPB_TARGET_BLOCKS = int(knob('DPP_BN_BLOCKS', '1024'))
# BatchNorm backward of the small maps: when the per-block sums of a BatchNorm are at most this many blocks, the finalize is done by
# every workgroup of the apply pass itself (dpp_bn_bwd_finalize_apply: one launch instead of two in the data-gradient chain; the
# stand-alone finalize is a launch + one memory round trip, 4.8 us).  0 = always two launches (the default).
# Measured on the MI355X (profiles/r04_whatif.txt): the fused launch takes 5.9 us against 5.5 + 4.6 and the single-stream kernel time of
# a bs128 step falls by 25 us (<= 128 blocks: 11 BatchNorms) / 54 us (<= 256: 31) -- but the two-stream step does not move: 3.503-3.508
# ms (128) and 3.531 (256) against 3.490 with two launches.  The backward pass is not a chain any more: the data-gradient stream and
# the filter-gradient branch end together and share the machine (dropping ALL 61 bn_bwd_finalize launches buys 0.07 ms, the 61
# forward bn_finalize launches 0.37 ms), so a launch saved on the chain is not time saved.
BN_BWD_FUSE_MAX_BLOCKS = int(knob('DPP_BN_BWD_FUSE_NB', '0'))
BN_BWD_FUSE_TARGET_WGS = int(knob('DPP_BN_BWD_FUSE_WGS', '512'))



GEMM_TARGET_BLOCKS = int(knob('DPP_GEMM_TARGET_BLOCKS', '1024'))


def gemm_plan(M, N, K, allow_split=True):
    """(tile, splitk) heuristics.  These GEMMs are latency / HBM-bound (K and N of 16..256), so the tile is the LARGEST one
    that still gives about GEMM_TARGET_BLOCKS workgroups (4 per CU: enough loads in flight to hide HBM latency); K is split
    when even the smallest tile leaves the grid short."""
    if M <= 16:
        bm, bn, wm = 16, 64, 1
    elif M <= 32:
        bm, bn, wm = 32, 64, 1
    else:
        wm = 4
        cands = [(128, 64), (64, 64), (128, 32), (64, 32), (128, 16), (64, 16)]
        if knob('DPP_NO_128x64', '1') != '0':
            cands = cands[1:]          # measured: 64x64 beats 128x64 on every conv shape of the net (gemm_micro.py)
        cands = [(a, b) for (a, b) in cands if b <= max(16, 16 * (-(-N // 16))) or b == 16]
        bm, bn = cands[-1]
        for (a, b) in cands:
            if (-(-M // a)) * (-(-N // b)) >= GEMM_TARGET_BLOCKS:
                bm, bn = a, b
                break
    tiles = (-(-M // bm)) * (-(-N // bn))
    splitk = 1
    if allow_split and tiles < 512 and K >= 256:
        # few output tiles, long K (the FC layers): split K until ~2048 workgroups with >= 128 of K each
        splitk = int(min(max(1, 2048 // tiles), max(1, K // 128), 256))
    return (bm, bn, wm), splitk


# Deterministic-mode engines (computeOutput, validation, the cascade's refinement net): a whole bottleneck block as ONE launch
# (hipdp/evalfuse.py, csrc/resblock.hip) and the stored-statistics coefficients of all remaining BatchNorms in one launch.
# DPP_EVAL_FUSE=0: the layer-by-layer decomposition of rounds 1-4 (131 launches per forward pass of the 128x128 ResNet).
EVAL_FUSE = knob('DPP_EVAL_FUSE', '1') != '0'
# round 6: the cost and its gradient computed by the split-K reduction of the last HiddenLayer (dpp_reduce_partials_loss).  DPP_FUSE_LOSS=0: own launch.
FUSE_LOSS = knob('DPP_FUSE_LOSS', '1') != '0'
# round 6: the ADAM launch advances t itself (dpp_adam_ticked).  DPP_ADAM_TICKED=0: the adam_tick launch behind it.
ADAM_TICKED = knob('DPP_ADAM_TICKED', '1') != '0'
# round 6: the fused block in the bf16 mode (bf16-stored input / output).  DPP_EVAL_FUSE_BF16=0: the layer-by-layer bf16 forward (the path
# whose every product can be pinned against the oracle: tests/test_configs.py).
EVAL_FUSE_BF16 = knob('DPP_EVAL_FUSE_BF16', '1') != '0'
OVERLAP_ALLREDUCE = knob('DPP_OVERLAP_ALLREDUCE', '1') != '0'
EARLY_BUCKET_MIN = int(knob('DPP_EARLY_BUCKET_MIN', str(1 << 22)))      # elements; FC1 of the 128x128 ResNet has 16.8 M
# 1x1 convolutions can read the BatchNorm gradient as (G, x) instead of a materialised dX (see _resolve_view).
#   1: data AND filter gradient take the two-tensor operand.  Takes 20 bn_bwd_apply launches (107 us) off the main chain but
#      makes 40 GEMMs 3.5 us slower each, half of them on the gradient branch, which the end of the step waits for
#      (4.87 vs 4.71 ms per step).
#   2: only the data gradient does, and it leaves the dX it forms in memory for the filter gradient (dpp_act.out), which
#      then starts after it instead of beside it.
#   3 (round 4): as 2, but only where that data gradient runs on the wave-autonomous kernel (dpp_gemm variant 4: the bottleneck
#      ENTRIES, whose data gradient expands K = 16 / 32 / 64 channels to 4 K).  There the operand is a few registers of a kernel that
#      is bound by its OUTPUT: alone the launch costs +1.2 .. 2 us (tools/gemm_micro.py expand: 8.8 -> 10.3 us at stage 3/4) against
#      the 6-9 us bn_bwd_apply launch it replaces.  In the step it still loses (3.60 vs 3.50 ms): the 16 filter gradients that read
#      the kept dX start one launch later, the gradient branch -- which ends together with the chain -- costs 0.27 instead of 0.18 ms
#      (tools/branch_probe.py), and the chain itself measures the same (3.26 vs 3.27 ms without the branch).  Off.
LAZY_BN_BWD = int(knob('DPP_LAZY_BN_BWD', '0'))
BF16_DEFAULT = os.environ.get('DPP_BF16', '0') == '1'
# bf16 mode (BASELINE config 5) also STORES the activation tensors -- every [pixels][channels] tensor a convolution writes -- as
# bfloat16 (ABI v9, DPP_ST_*): rounded by the producer's epilogue (statistics from the f32 values), widened by every reader.  At
# 256x256 the step is bandwidth-bound and these tensors are read five to six times each.  DPP_BF16_STORE=0: f32 storage (rounds 2-3).
BF16_STORE = knob('DPP_BF16_STORE', '1') != '0'
# ... and the GRADIENTS of those tensors (the masked gradient G a data-gradient epilogue writes into a BatchNorm view, the dX that
# bn_bwd_apply writes): the backward pass moves twice the bytes of the forward pass.  DPP_BF16_GRADS=0: float32 gradients.
BF16_GRADS = knob('DPP_BF16_GRADS', '1') != '0'
# ... and runs the channel-expanding 1x1 convolutions / the data gradients of the reducing ones (dpp_gemm variant 4, K = 32 / 64) on
# bf16 MFMA operands (dpp_gemm_desc.precision).  DPP_BF16_GEMM=0: f32 MFMA there (rounds 2-3: only the 3x3 convolutions and FC1).
BF16_GEMM = knob('DPP_BF16_GEMM', '1') != '0'
# round 6: also on the LDS-tiled, K-split and 16-column-stream kernels (reducing 1x1 convolutions, their twins' data gradients, the 1x1
# filter gradients of stages 2-4).  DPP_BF16_GEMM_ALL=0: the round-4/5 set (wave-autonomous kernel only).
BF16_GEMM_ALL = knob('DPP_BF16_GEMM_ALL', '1') != '0'
BF16_GEMM_ROLES = tuple(knob('DPP_BF16_GEMM_ROLES', 'fwd,dgrad,wgrad').split(','))       # (bisecting: which products of the round-6 set)
BF16_GEMM_VARIANTS = tuple(knob('DPP_BF16_GEMM_VARIANTS', '0,2,3').split(','))            # (bisecting: on which dpp_gemm variants)
# round 6: the 3x3 filter gradients of the 16- / 32-channel layers on bf16 MFMA operands (dpp_conv3x3_wgrad_bf16).  DPP_BF16_WGRAD3=0: f32 MFMA.
BF16_WGRAD3 = knob('DPP_BF16_WGRAD3', '1') != '0'
# ADAM of the FC1 weight (90 % of the parameters, 470 MB of optimizer traffic) inside the backward pass: its gradient is final a few
# kernels into the pass, so the update runs on the gradient branch under the latency-bound data-gradient chain instead of in the
# serial tail of the step (step_plan only: cost_and_grads / allreduce paths keep backward and update apart).
# Measured on the MI355X (tools/exp_tail.sh, profiles/r02_tail_experiments.txt): 4.19-4.20 vs 4.18 ms per step -- the gradient branch
# finishes together with the data-gradient chain (tools/tail_probe.py: 2.51 vs 2.53 ms), so work moved onto it comes back as a longer
# wait at the join.  Off by default; the plan surgery stays tested (tests/test_engine.py).
EARLY_ADAM = knob('DPP_EARLY_ADAM', '0') != '0'
# Filter / bias gradient partials are summed by dpp_reduce_multi.  One launch at the end of the pass reads all of them (224 MB) in the
# serial tail of the step; with a threshold the jobs collected so far are reduced on the gradient branch as soon as they amount to
# this many bytes (their producers are on that branch or already issued on the main stream), and the tail launch keeps the rest.
# Measured: 4.21 (16 MB) / 4.19 (64 MB) vs 4.18 ms with the single launch, for the same reason as EARLY_ADAM.  0 = one launch (default).
EARLY_REDUCE_BYTES = int(knob('DPP_EARLY_REDUCE_MB', '0')) << 20
# the partials collected before the stem reduced on the branch beside the stem's filter gradient: 3.661-3.669 vs 3.673-3.680 ms over
# 300-step runs -- the branch, not the main stream, is what the join waits for; off
TAIL_REDUCE = knob('DPP_TAIL_REDUCE', '0') != '0'
# FC1 (the HiddenLayer behind the last conv map) on the weight-streaming kernels of dpp_fc_gemm instead of the generic dpp_gemm.
# f32: the three-stage kernel (fc_stream_kernel: 128 x 128 / 128 x 64 tiles, whole tiles only) runs the batch-128 FC1 forward /
# data gradient in 53 / 70 us against dpp_gemm's 102 / 104 us; the older double-buffered kernel, which takes ragged shapes, only
# ties with dpp_gemm in f32 (profiles/r02_fc1_kernels.txt), so f32 goes there only when the shape fits the three-stage kernel.
# bf16 (config 5) always uses dpp_fc_gemm: dpp_gemm has no bf16 counterpart.
#   DPP_FC1_STREAM=auto (default) | bf16: bf16 only | 1: always | 0: never (bf16 FC1 then falls back to f32 dpp_gemm)
FC1_STREAM = knob('DPP_FC1_STREAM', 'auto')
FC1_WGRAD_STREAM = knob('DPP_FC1_WGRAD_STREAM', '1') != '0'       # FC1's filter gradient on dpp_fc_wgrad_stream (f32)
FC1_WGRAD_DEFER = int(knob('DPP_FC1_WGRAD_DEFER', '0'))          # see CompiledNet._defer_fc1_wgrad
FC1_KCHUNK = int(knob('DPP_FC1_KCHUNK', '0'))
FC1_SLICES = int(knob('DPP_FC1_SLICES', '32'))
FC1_MIN_K = int(knob('DPP_FC1_MIN_K', '4096'))


def is_fc1_shape(Nb, K, Nout):
    """The weight-streaming shape: tens of MB of weights for at most a few hundred rows."""
    return K >= FC1_MIN_K and Nout >= 64 and Nout % 4 == 0 and K % 4 == 0
ROWSTREAM = knob('DPP_ROWSTREAM', '0') != '0'      # measured: no gain over the LDS-tiled kernel yet

def rowstream_plan(M, N, K, b_kc):
    """Tile of the barrier-free row-streaming GEMM variant for conv-shaped problems (M = pixels >> K, N), or None when it
    does not apply: the whole K x bn weight slice must fit the 64 KB LDS window, and the grid should fill the chip."""
    if not ROWSTREAM or M < 2048:
        return None
    K16 = (K + 15) // 16 * 16
    for bn in (64, 32, 16):
        if bn > 16 and bn >= 2 * N:
            continue
        lds = (bn * (K16 + 4) if b_kc else K16 * (bn + 4)) * 4
        if lds > 48 * 1024:
            continue
        bm = 128 if (M // 128) * (-(-N // bn)) >= 1024 else 64
        return (bm, bn, 4)
    return None


# dpp_gemm variant 2 (gemm_ksplit_kernel): 32 rows x all columns x the WHOLE K per workgroup, one memory round trip, the four waves
# split K.  For the long-K / narrow-N 1x1 convolutions of the late stages, K-contiguous A, whole tiles only.
#   DPP_KSPLIT = 0: off | 1: K = 256 (stage 3 / 4 bottleneck entries and the data gradients of their exits) | 2: K = 128 as well
KSPLIT = int(knob('DPP_KSPLIT', '2'))
# (round 6: 16384 -> 65536.  At 128 x 128 no K = 256 layer has more than 16 384 rows, so the old limit only sent the 256 x 256 net's stage-3
# layers (32 768 rows) back to the LDS-tiled kernel: bf16 256 x 256 step 7.754 -> 7.515 ms with the limit raised, same box)
KSPLIT_MAX_M = int(knob('DPP_KSPLIT_MAX_M', '65536'))


def ksplit_plan(M, N, K):
    if KSPLIT <= 0 or M % 32 or M > KSPLIT_MAX_M * (2 if K == 128 else 1):
        return None
    if K == 256 and N % 64 == 0:
        return (32, 64, 4)
    if K == 128 and N % 32 == 0 and KSPLIT >= 2:
        return (32, 32, 4)
    return None


# dpp_gemm variant 3 (gemm_stream16_kernel): the stage-1 bottleneck entries and the data gradients of their exits (K = 64 -> 16
# columns; 2 (default): also the K = 16 -> 64 columns data gradients; 3: and those forward convolutions) as a barrier-free row stream,
# see csrc/gemm.hip.  DPP_STREAM16 = 0 | 1 | 2 | 3.  No lower bound on the row count by default:
# the kernel sums k in another order than the LDS-tiled one, and a frame's joints must not depend on the batch it is evaluated in
# (tests/test_full_size.py compares batches of 8 and 128 at 1e-4 mm), so the choice of kernel must not depend on the batch either.
STREAM16 = int(knob('DPP_STREAM16', '2'))
STREAM16_MIN_M = int(knob('DPP_STREAM16_MIN_M', '128'))


def stream16_plan(M, N, K, forward):
    """K = 64 -> 16 columns: 128 rows per workgroup; K = 16 -> 64 columns: 64 rows, data gradients only (DPP_STREAM16 >= 2; the
    forward pass gains nothing there -- 22.8 vs 23.5 us: reading the residual and writing 256-byte rows 4 bytes per lane costs what the
    LDS transposition did -- and its other summation order put the bs256 training-mode forward at 1.02e-5 of the output scale from
    the float32 oracle, against the 1e-5 bar of tests/test_configs.py)."""
    if STREAM16 <= 0 or M < STREAM16_MIN_M:
        return None
    if N == 16 and K == 64 and M % 128 == 0:
        return (128, 16, 4)
    if N == 64 and K == 16 and M % 64 == 0 and (STREAM16 >= 3 or (STREAM16 >= 2 and not forward)):
        return (64, 64, 4)
    return None


# dpp_gemm variant 4 (gemm_expand_kernel): the channel-expanding 1x1 convolutions (K = 16 / 32 / 64 -> N = 64 / 128 / 256 columns: the
# bottleneck exits, with bias + residual + statistics) and the data gradients of the reducing ones (same shapes, BatchNorm-backward
# epilogue) as wave-autonomous 64-column strips -- no LDS, no barrier, 16-byte accesses straight from the MFMA D layout, see
# csrc/gemm.hip.  DPP_EXPAND = 0 | 1 (forward) | 2 (data gradients) | 3 (both).  Rows per wave (= rows per statistics block) by stage.
EXPAND = int(knob('DPP_EXPAND', '3'))
EXPAND_RPW = tuple(int(v) for v in knob('DPP_EXPAND_RPW', '128,64,32').split(','))      # stage 1 | stage 2 | stages 3-4
EXPAND_MAX_WGS = int(knob('DPP_EXPAND_MAX_WGS', '256'))                                # workgroups of one launch (0: no limit)


def expand_plan(M, N, K, forward):
    if not (EXPAND & (1 if forward else 2)) or K not in (16, 32, 64) or N % 64 or N < 2 * K:
        return None
    rpw = EXPAND_RPW[0 if M >= 65536 else (1 if M >= 16384 else 2)]
    while rpw > 32 and M % rpw:
        rpw //= 2
    # ... and more rows per wave where the launch would not be resident at once: the kernel's instances are ALLOCATED 196-374 registers (the code
    # objects' .vgpr_count; the profiler's column shows the architectural half), i.e. ONE workgroup per CU, two for the smallest -- the launches of
    # the 256 x 256 net (512-1 024 workgroups at the rows per wave tuned for 128 x 128) ran as two to four rounds, every round paying the entry
    # code's round trips again.  Same box, bf16 256 x 256: no limit 6.995 / 6.993 ms, 768 workgroups 6.944-6.978, 384 6.80 / 6.82, 256 6.796 / 6.814;
    # float32 256 x 256 9.228 -> 9.037 (tools/prof_summary.py --by-grid prints allocation, workgroups per CU and rounds of every launch).
    while EXPAND_MAX_WGS and (M // rpw) * (N // 64) // 4 > EXPAND_MAX_WGS and M % (2 * rpw) == 0:
        rpw *= 2
    if M % rpw:
        return None
    return (rpw, 64, 4)


C3_BM128_32 = knob('DPP_C3_BM128_32', '1') != '0'


C3_P64 = knob('DPP_C3_P64', '1') != '0'


def conv3x3_bm(pixels, Co, hw=None, prec=0):
    """Rows per workgroup of the 3x3 kernel (the choice dpp_conv3x3 makes for bm = 0, made explicit so that the host knows
    the row-block count of the fused epilogue partials).  `hw`: the map size, `prec`: bf16 MFMA operands, where the caller knows them."""
    if Co == 64 and C3_P64 and prec and hw in ((8, 8), (16, 16)) and pixels % 128 == 0:
        # the 64-channel layers of 8 x 8 maps (stages 3-4 at 128 x 128 input) on the tile-walking kernel: two whole images per workgroup, 16 of the
        # 64 columns each -- 256 workgroups, all nine weight slices resident (conv3x3_p_kernel, GEOM 1).  bf16 operands only: 2.998 -> 2.928 ms per
        # step; the float32 form needs 94 KB of LDS per workgroup and the step is slower with it (3.371 -> 3.400 ms), see conv3x3.hip
        return 128
    if Co == 32 and C3_BM128_32 and pixels % 128 == 0 and pixels // 128 >= 256:
        # the square 32-channel layers on the tile-walking kernel (conv3x3_p_kernel: 8 x 16 tiles): at 128 x 128 input 256 workgroups, one per
        # CU, whose barrier-free tap loop runs at the float32 MFMA rate -- 512 workgroups of 64 pixels spent 6.8 us in their nine barriers
        return 128
    return 128 if (pixels // 128) * (-(-Co // 64)) >= 512 else 64


# Projection shortcuts of the residual blocks on the second stream beside the bottleneck (see _emit_add).  DPP_SIDE_SHORTCUT=0: the
# later convolution of a sum absorbs the add and everything stays on the main stream (rounds 1-3).
SIDE_SHORTCUT = knob('DPP_SIDE_SHORTCUT', '1') != '0'


# 3x3 convolutions of the narrow square layers (and their data gradients) on the barrier-free dpp_conv3x3_stream for these channel
# counts.  OFF by default: measured on the MI355X (profiles/r04_conv3x3_stream.txt) the kernel is no faster than the LDS-tiled one on
# the shapes it was written for -- stage 1 (131 072 px, 16 -> 16) 14.7 us plain / 17.4 us with prologue + statistics against 12.1 /
# 13.4 us, insensitive to the tiles per wave (1 | 2 | 4), and the bs128 step is 3.55 ms with it against 3.49 ms; stage 2 (32 -> 32,
# 144 filter registers, one wave per SIMD) 16-22 us against 11-12 us.  Every wave re-loads the 9 KB filter through the texture path
# (twice the operand traffic) and redoes the BatchNorm prologue for each of the nine taps; the tiled kernel pays one barrier per tap
# but stages and activates each pixel once.
CONV3_STREAM_C = tuple(int(v) for v in knob('DPP_CONV3_STREAM_C', '').split(',') if v)


WGRAD_TARGET_BLOCKS = int(knob('DPP_WGRAD_TARGET_BLOCKS', '256'))


def wgrad_plan(Co, Ci, K):
    """Filter gradients reduce over K = pixels (1e4..1e5) into a small [Co][Ci] matrix.  Measured (tools/gemm_micro.py wgrad):
    the fewest, largest tiles that cover [Co][Ci] win, with the pixel reduction split until 512..1024 workgroups are in
    flight but no slice shorter than ~128 pixels."""
    if Co <= 16:
        tile = (16, 64, 1)
    elif Co <= 32:
        tile = (32, 64, 1)
    else:
        tile = (64, 64 if Ci > 32 else (32 if Ci > 16 else 16), 4)
    tiles = (-(-Co // tile[0])) * (-(-Ci // tile[1]))
    splitk = int(max(1, min(WGRAD_TARGET_BLOCKS // tiles, max(K // 128, min(128, K // 64)))))
    return tile, splitk


# 1x1 filter gradients on the row-streaming kernel of csrc/wgrad.hip.  Measured on the MI355X (profiles/r03_wgrad_stream.txt): alone
# (one stream) the kernel beats dpp_gemm's filter-gradient layout on every stage (stage 1: 14 vs 43 us per launch under the
# profiler), but the gradient branch runs BESIDE the data-gradient chain, and there what counts is how little a launch takes
# from the chain, not how fast it is: with 512 workgroups per launch the step got SLOWER (3.84 vs 3.77 ms), with 128 long-running
# workgroups (256 rows per wave) on the stage-1 layers only it is 3.72 ms.  Stages 2-4 stay on dpp_gemm (3.76 / 3.84 ms with the
# stream kernel there), as does the two-tensor dY operand of DPP_LAZY_BN_BWD.
WGRAD_STREAM = knob('DPP_WGRAD_STREAM', '1') != '0'
WGRAD_STREAM_RPW = tuple(int(v) for v in knob('DPP_WGRAD_STREAM_RPW', '256,128,128').split(','))      # stage 1 | stage 2 | stages 3-4
WGRAD_STREAM_STAGES = knob('DPP_WGRAD_STREAM_STAGES', '1')            # which of them take the kernel


# 3x3 filter gradients on dpp_wgrad3_stream for these channel counts (the rest stays on the LDS-tiled dpp_conv3x3_wgrad)
WGRAD3_STREAM_C = tuple(int(v) for v in knob('DPP_WGRAD3_STREAM_C', '64').split(',') if v)


WGRAD3_STREAM_SLICE_BUDGET = int(knob('DPP_WGRAD3_STREAM_SLICES', '1024'))       # slices of a 16-channel layer (9.2 KB each)


def wgrad3_stream_rows(M, C):
    """Pixel rows per wave of dpp_wgrad3_stream for a C -> C 3x3 layer over M pixels, or 0 (LDS-tiled kernel).  The partial slices of
    a layer are held to ~9.4 MB (1 024 slices of 16 x 9 x 16, 256 of 32 x 9 x 32, 64 of 64 x 9 x 64): more slices is more waves but
    the partials' write + re-read grows past the tensors themselves (tools/gemm_micro.py conv3, profiles/r03_wgrad3_stream.txt)."""
    if C not in WGRAD3_STREAM_C:
        return 0
    slices = max(1, WGRAD3_STREAM_SLICE_BUDGET // max(1, (C // 16) ** 2))
    return max(64, (M // slices) & ~3)


def wgrad_stream_rows(M):
    """Pixel rows per wave of dpp_wgrad_stream for a layer with M pixel rows, or 0: leave the layer on dpp_gemm."""
    stage = 0 if M >= 65536 else (1 if M >= 16384 else 2)
    if str(stage + 1) not in WGRAD_STREAM_STAGES:
        return 0
    r = WGRAD_STREAM_RPW[min(stage, len(WGRAD_STREAM_RPW) - 1)]
    return r if M >= 1024 else 32
