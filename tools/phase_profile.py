#!/usr/bin/env python3
"""Where the time goes INSIDE the small kernels of the critical chain: phase stamps (s_memrealtime, 100 MHz) written by the
profiling build of the library (make -C deep-prior-pp_amd/csrc prof -> deep-prior-pp_amd/lib_prof/libdpp_hip.so).
   python tools/phase_profile.py > profiles/r02_phase_profile.txt

For each case the kernel is launched alone (after warm-up launches), every workgroup stamps its phases, and the table gives, over
the workgroups: when they START relative to the first one (dispatch skew), the median / max duration of each phase, and the
kernel's span (first start -> last end).  gemm phases: 0 entry -> 1 loads of the first K chunk issued -> 2 first chunk in LDS
(load latency + commit + barrier) -> 3 K loop done -> 4 epilogue done.  conv3x3: 0 entry -> 1 halo staged (loads + LDS writes
issued) -> 2 halo + weights visible (barrier) -> 3 nine taps done -> 4 epilogue done."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hipdp import ops  # noqa: E402
from hipdp.lib import Act  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402

rt = TorchHipRuntime(lib_path=os.path.join(ROOT, 'deep-prior-pp_amd', 'lib_prof', 'libdpp_hip.so'))
TICK_US = 0.01


def profile(label, launch, nwg, nphase=5, slots=None):
    buf = rt.alloc((nwg + 8, 16), np.int64)
    for _ in range(5):
        launch(rt.stream)
    torch.cuda.synchronize()
    rt.lib.dpp_prof_set(buf.ptr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch(rt.stream)
    e1.record()
    torch.cuda.synchronize()
    rt.lib.dpp_prof_set(None)
    t = buf.get()[:nwg, :nphase].astype(np.float64) * TICK_US
    if slots is not None:
        t, nphase = t[:, list(slots)], len(slots)
    t0 = t[:, 0].min()
    start = t[:, 0] - t0
    span = t[:, nphase - 1].max() - t0
    ph = np.diff(t, axis=1)
    print('%-44s %4d WGs  span %6.2f us (events incl. launch: %6.2f)  start skew med %5.2f max %5.2f | phases med/max: %s' % (
        label, nwg, span, e0.elapsed_time(e1) * 1e3, np.median(start), start.max(),
        '  '.join('%5.2f/%5.2f' % (np.median(ph[:, k]), ph[:, k].max()) for k in range(nphase - 1))))
    sys.stdout.flush()


class BN(object):
    pass


def gemm_case(label, M, N, K, tile, feats, variant=0):
    A = rt.alloc((M, K), zero=False)
    rt.tensor(A).normal_()
    B = rt.alloc((N, K), zero=False)
    rt.tensor(B).normal_()
    Cb = rt.alloc((M, N), zero=False)
    kw = {}
    if 'act' in feats:
        mean, scale, beta = rt.alloc(K), rt.alloc(K), rt.alloc(K)
        a = Act(mean.ptr, scale.ptr, beta.ptr, 3, K)
        a._keep = (mean, scale, beta)
        kw['actA'] = a
    if 'bias' in feats:
        kw['bias'] = rt.alloc(N)
    if 'res' in feats:
        kw['residual'] = rt.alloc((M, N))
    if 'stats' in feats:
        kw['epi'] = ops.epilogue(stats=rt.alloc((-(-M // tile[0]), 2, N), zero=False))
    L = ops.gemm(rt, A, B, Cb, M, N, K, 1, 1, K, K, N, tile=tile, variant=variant, **kw)
    profile('%s %s' % (label, '+'.join(feats) or 'plain'), L, -(-M // tile[0]) * -(-N // tile[1]))


def conv_case(label, N, H, C, bm, feats):
    X = rt.alloc((N, H, H, C), zero=False)
    rt.tensor(X).normal_()
    Wk = rt.alloc((C, 9, C), zero=False)
    rt.tensor(Wk).normal_()
    Y = rt.alloc((N, H, H, C), zero=False)
    kw = {}
    if 'act' in feats:
        mean, scale, beta = rt.alloc(C), rt.alloc(C), rt.alloc(C)
        a = Act(mean.ptr, scale.ptr, beta.ptr, 3, C)
        a._keep = (mean, scale, beta)
        kw['actX'] = a
    if 'bias' in feats:
        kw['bias'] = rt.alloc(C)
    nblk = rt.lib.dpp_conv3x3_tiling(N, H, H, bm, None, None, None)
    if 'stats' in feats:
        kw['epi'] = ops.epilogue(stats=rt.alloc((nblk, 2, C), zero=False))
    L = ops.conv3x3(rt, X, N, H, H, C, Wk, C, Y, bm=bm, **kw)
    bn = 64 if C >= 64 else (32 if C >= 32 else 16)
    while bn > 16 and nblk * -(-C // bn) < 1024:
        bn >>= 1
    profile('%s bm=%d %s' % (label, bm, '+'.join(feats) or 'plain'), L, nblk * -(-C // bn))


def expand_case(label, M, N, K, rpw, forward=True):
    """gemm_expand_kernel (dpp_gemm variant 4).  Phases of a workgroup's first wave: 0 entry -> 1 filter slice, coefficients and the first 32 rows'
    loads issued -> 3 all rows done -> 4 statistics / BatchNorm-backward partials written."""
    A = rt.alloc((M, K), zero=False)
    rt.tensor(A).normal_()
    B = rt.alloc((N, K) if forward else (K, N), zero=False)
    rt.tensor(B).normal_()
    Cb = rt.alloc((M, N), zero=False)
    kw = {}
    nblk = M // rpw
    if forward:
        mean, scale, beta = rt.alloc(K), rt.alloc(K), rt.alloc(K)
        a = Act(mean.ptr, scale.ptr, beta.ptr, 3, K)
        a._keep = (mean, scale, beta)
        kw = dict(actA=a, bias=rt.alloc(N), residual=rt.alloc((M, N)), epi=ops.epilogue(stats=rt.alloc((nblk, 2, N), zero=False)))
    else:
        bn = BN()
        bn.mean, bn.inv_std, bn.scale, bn.beta_buf = rt.alloc(N), rt.alloc(N), rt.alloc(N), rt.alloc(N)
        kw = dict(epi=ops.epilogue(bn=bn, bn_x=rt.alloc((M, N)), bn_relu=True, bn_partial=rt.alloc((nblk, 2, N), zero=False)))
    L = ops.gemm(rt, A, B, Cb, M, N, K, 1, 1 if forward else 0, K, K if forward else N, N, tile=(rpw, 64, 4), variant=4, **kw)
    profile('%s rpw %d %s' % (label, rpw, 'forward' if forward else 'data gradient'), L, (M // rpw) * (N // 64) // 4, slots=(0, 1, 3, 4))


def fc_case(label, M, N, K, a_kc, b_kc, tile, splitk):
    A = rt.alloc((M, K) if a_kc else (K, M), zero=False)
    rt.tensor(A).normal_()
    B = rt.alloc((N, K) if b_kc else (K, N), zero=False)
    rt.tensor(B).normal_()
    Cb = rt.alloc((M, N), zero=False)
    part = rt.alloc((splitk, M, N), zero=False) if splitk > 1 else None
    L = ops.gemm(rt, A, B, None if splitk > 1 else Cb, M, N, K, a_kc, b_kc, K if a_kc else M, K if b_kc else N, N, splitk=splitk, partial=part, tile=tile)
    profile('%s tile %dx%d splitk %d' % (label, tile[0], tile[1], splitk), L, -(-M // tile[0]) * -(-N // tile[1]) * splitk)


def resblock_case(label, N, H, Nb, proj=False):
    """The fused deterministic-mode bottleneck block (csrc/resblock.hip).  Phases: 0 entry -> 1 halo loads issued, activated, written to
    LDS -> 2 barrier -> 3 phase A (c1 over the halo) -> 4 A1 written + barrier -> 5 phase B (3x3) -> 6 A2 written + barrier -> 7 phase C."""
    Cin, Cout = (2 * Nb if proj else 4 * Nb), 4 * Nb
    s = 2 if proj else 1
    Hi = H * s
    X = rt.alloc((N, Hi, Hi, Cin), zero=False)
    rt.tensor(X).normal_()
    Y = rt.alloc((N, H, H, Cout), zero=False)

    def vec(n, lo=0.5, hi=1.5):
        b = rt.alloc(n, zero=False)
        rt.tensor(b).uniform_(lo, hi)
        return b

    def w(*shape):
        b = rt.alloc(shape, zero=False)
        rt.tensor(b).normal_(0, 0.05)
        return b
    bns = [ops.bn_eval(vec(c, -0.3, 0.3), vec(c), vec(c), vec(c, -0.3, 0.3)) for c in (Cin, Nb, Nb)]
    kw = dict(Wsc=w(Cout, Cin), bsc=vec(Cout)) if proj else {}
    L = ops.resblock_eval(rt, X, N, Hi, Hi, Cin, s, Cout, Nb, bns[0], bns[1], bns[2], w(Nb, Cin), vec(Nb), w(Nb, 9, Nb), vec(Nb), w(Cout, Nb), vec(Cout),
                          Y, **kw)
    th, tw = (4 if Nb == 64 else 8), 8
    profile(label, L, N * -(-H // th) * -(-H // tw), nphase=8)


if __name__ == '__main__':
    if 'resblock' in sys.argv[1:]:
        resblock_case('block stage3/4 256 -> 64 -> 256, 8x8 maps', 128, 8, 64)
        resblock_case('block stage2   128 -> 32 -> 128, 16x16 maps', 128, 16, 32)
        resblock_case('block stage1    64 -> 16 ->  64, 32x32 maps', 128, 32, 16)
        resblock_case('projection stage3 128 -> 64 -> 256 /2', 128, 8, 64, proj=True)
        resblock_case('projection stage2  64 -> 32 -> 128 /2', 128, 16, 32, proj=True)
        resblock_case('projection stage1  32 -> 16 ->  64 /2', 128, 32, 16, proj=True)
        sys.exit(0)
    if 'expand' in sys.argv[1:]:
        for fwd in (True, False):
            expand_case('stage3/4  64 -> 256, 8 192 rows', 8192, 256, 64, 32, fwd)
            expand_case('stage2    32 -> 128, 32 768 rows', 32768, 128, 32, 64, fwd)
            expand_case('stage1    16 -> 64, 131 072 rows', 131072, 64, 16, 128, fwd)
        sys.exit(0)
    if 'fc' in sys.argv[1:]:
        fc_case('FC1 fwd   128 x 1024 x 16384', 128, 1024, 16384, 1, 0, (128, 64, 4), 32)
        fc_case('FC1 fwd   128 x 1024 x 16384', 128, 1024, 16384, 1, 0, (64, 64, 4), 32)
        fc_case('FC1 dgrad 128 x 16384 x 1024', 128, 16384, 1024, 1, 1, (128, 64, 4), 1)
        fc_case('FC1 dgrad 128 x 16384 x 1024', 128, 16384, 1024, 1, 1, (64, 64, 4), 1)
        fc_case('FC1 wgrad 16384 x 1024 x 128', 16384, 1024, 128, 0, 0, (128, 64, 4), 1)
        fc_case('FC1 wgrad 16384 x 1024 x 128', 16384, 1024, 128, 0, 0, (64, 64, 4), 1)
        sys.exit(0)
    full = ('act', 'bias', 'res', 'stats')
    for feats in ((), full):
        gemm_case('1x1 stage3/4 a 256->64  tile 64x16', 8192, 64, 256, (64, 16, 4), feats)
        gemm_case('1x1 stage3/4 c 64->256  tile 64x64', 8192, 256, 64, (64, 64, 4), feats)
        gemm_case('1x1 stage3/4 a 256->64  K-split 32x64', 8192, 64, 256, (32, 64, 4), feats, variant=2)
        gemm_case('1x1 stage2  a 128->32   tile 64x32', 32768, 32, 128, (64, 32, 4), feats)
        gemm_case('1x1 stage2  a 128->32   K-split 32x32', 32768, 32, 128, (32, 32, 4), feats, variant=2)
        gemm_case('1x1 stage2  c 32->128   tile 64x64', 32768, 128, 32, (64, 64, 4), feats)
        gemm_case('1x1 stage1  a 64->16    tile 128x16', 131072, 16, 64, (128, 16, 4), feats)
        gemm_case('1x1 stage1  c 16->64    tile 64x64', 131072, 64, 16, (64, 64, 4), feats)
    for feats in ((), ('act', 'bias', 'stats')):
        conv_case('3x3 stage3/4 64->64', 128, 8, 64, 64, feats)
        conv_case('3x3 stage2 32->32', 128, 16, 32, 64, feats)
        conv_case('3x3 stage1 16->16', 128, 32, 16, 128, feats)
