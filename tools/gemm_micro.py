#!/usr/bin/env python3
"""Micro-benchmark of single kernels of the hot path on the GPU (back-to-back launches, HIP events):
   python tools/gemm_micro.py      -> one line per (shape, variant, feature set) with us/launch and GB/s"""
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

rt = TorchHipRuntime()
REP = 200


def timeit(launch, rep=REP):
    """us per launch, GPU side: `rep` copies of the launch replayed as ONE explicit hipGraph chain (0.3 us of host work per
    launch), so that kernels shorter than the host's 3-5 us per eager launch are not measured at the host's pace.  The figure
    includes the dependent-kernel boundary (~1.5 us)."""
    plan = ops.NativePlan(rt, [(launch, False)] * rep, mode='graph1')
    plan.run(rt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        plan.run(rt)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * rep)


class BN(object):
    pass


def bench_gemm(M, N, K, b_kc, tile, variant, feats, label, stride2=None):
    """stride2 = (Ho, Wo): the A rows are the stride-2 samples of a (2 Ho, 2 Wo) map (the first 1x1 convolutions of a stage)."""
    A = rt.alloc((M * (4 if stride2 else 1), K), zero=False)
    rt.tensor(A).normal_()
    B = rt.alloc((N, K) if b_kc else (K, N), zero=False)
    rt.tensor(B).normal_()
    Cb = rt.alloc((M, N), zero=False)
    kw = {}
    if 'relu' in feats:
        kw['actA'] = Act(None, None, None, 1, K)
    if 'act' in feats:
        mean, scale, beta = rt.alloc(K), rt.alloc(K), rt.alloc(K)
        a = Act(mean.ptr, scale.ptr, beta.ptr, 3, K)
        a._keep = (mean, scale, beta)
        kw['actA'] = a
    if 'lazy' in feats or 'lazykeep' in feats:                    # the mode-4 operand: gradient through a BatchNorm formed from (G, x)
        bn = BN()
        bn.mean, bn.scale = rt.alloc(K), rt.alloc(K)
        q, pp, x2 = rt.alloc(K), rt.alloc(K), rt.alloc((M, K))
        kw['actA'] = ops.act_bn_bwd(bn, q, pp, x2, K, out=rt.alloc((M, K)) if 'lazykeep' in feats else None)
        kw['actA']._keep2 = (bn, q, pp, x2)
    if 'bias' in feats:
        kw['bias'] = rt.alloc(N)
    if 'res' in feats:
        kw['residual'] = rt.alloc((M, N))
    if 'stats' in feats:
        nblk = -(-M // tile[0])
        kw['epi'] = ops.epilogue(stats=rt.alloc((nblk, 2, N), zero=False))
    if 'bnbwd' in feats:
        nblk = -(-M // tile[0])
        bn = BN()
        bn.mean, bn.inv_std, bn.scale, bn.beta_buf = rt.alloc(N), rt.alloc(N), rt.alloc(N), rt.alloc(N)
        kw['epi'] = ops.epilogue(bn=bn, bn_x=rt.alloc((M, N)), bn_partial=rt.alloc((nblk, 2, N), zero=False))
    if stride2:
        from hipdp.lib import RowMap
        kw['mapA'] = RowMap.strided(2, stride2[0], stride2[1], 2 * stride2[0], 2 * stride2[1])
    L = ops.gemm(rt, A, B, Cb, M, N, K, 1, int(b_kc), K, K if b_kc else N, N, tile=tile, variant=variant, **kw)
    try:
        us = timeit(L)
    except Exception as e:      # noqa: BLE001
        print('%-34s tile=%-12s %s' % (label, tile, e))
        return
    byts = 4.0 * (M * K + K * N + M * N * (2 if 'res' in feats else 1) + (M * N if 'bnbwd' in feats else 0))
    print('%-34s M=%6d N=%4d K=%4d tile=%-12s v%d %-22s %7.2f us  %6.0f GB/s' %
          (label, M, N, K, tile, variant, '+'.join(feats) or '-', us, byts / us * 1e-3))


def bench_fc(label, M, N, K, a_kc, b_kc, tile, splitk):
    A = rt.alloc((M, K), zero=False)
    rt.tensor(A).normal_()
    B = rt.alloc((K, N), zero=False)
    rt.tensor(B).normal_()
    Cb = rt.alloc((M, N), zero=False)
    lda = K if a_kc else M
    ldb = K if b_kc else N
    try:
        if splitk > 1:
            part = rt.alloc((splitk, M, N), zero=False)
            L1 = ops.gemm(rt, A, B, None, M, N, K, a_kc, b_kc, lda, ldb, N, splitk=splitk, partial=part, tile=tile)
            L2 = ops.reduce_partials(rt, part, splitk, M * N, Cb)
            us1, us2 = timeit(L1, 50), timeit(L2, 50)
        else:
            us1, us2 = timeit(ops.gemm(rt, A, B, Cb, M, N, K, a_kc, b_kc, lda, ldb, N, tile=tile), 50), 0.0
    except Exception as e:
        print('%-12s tile=%s splitk=%d: %s' % (label, tile, splitk, e))
        return
    byts = 4.0 * (M * K + K * N + M * N)
    print('%-12s M=%6d N=%6d K=%6d akc=%d bkc=%d tile=%-13s splitk=%3d  gemm %7.2f + reduce %6.2f us  (%5.0f GB/s algorithmic)' %
          (label, M, N, K, a_kc, b_kc, tile, splitk, us1, us2, byts / (us1 + us2) * 1e-3))


def main_fc():
    for tile in ((64, 16, 4), (64, 32, 4), (64, 64, 4), (128, 64, 4), (128, 32, 4)):
        for sk in (8, 16, 32, 64):
            bench_fc('fc1 fwd', 128, 1024, 16384, 1, 0, tile, sk)
    for tile in ((64, 16, 4), (64, 32, 4), (64, 64, 4), (128, 64, 4), (128, 32, 4)):
        for sk in (1, 2, 4):
            bench_fc('fc1 dgrad', 128, 16384, 1024, 1, 1, tile, sk)
    for tile in ((64, 64, 4), (128, 64, 4), (128, 32, 4), (64, 32, 4)):
        bench_fc('fc1 wgrad', 16384, 1024, 128, 0, 0, tile, 1)
    for tile in ((64, 16, 4), (64, 32, 4), (64, 64, 4), (128, 32, 4)):
        for sk in (1, 4, 8):
            bench_fc('fc2 fwd', 128, 1024, 1024, 1, 0, tile, sk)
            bench_fc('fc2 dgrad', 128, 1024, 1024, 1, 1, tile, sk)
        bench_fc('fc2 wgrad', 1024, 1024, 128, 0, 0, tile, 1)


def main_fcw():
    """FC1's filter gradient: dpp_fc_gemm's layout against dpp_fc_wgrad_stream."""
    for Nb, K, N in ((128, 16384, 1024), (128, 65536, 1024), (256, 16384, 1024)):
        X = rt.alloc((Nb, K), zero=False)
        rt.tensor(X).normal_()
        dY = rt.alloc((Nb, N), zero=False)
        rt.tensor(dY).normal_()
        dW = rt.alloc((K, N), zero=False)
        mean, scale, beta = rt.alloc(256), rt.alloc(256), rt.alloc(256)
        act = Act(mean.ptr, scale.ptr, beta.ptr, 3, 256)
        t0 = timeit(ops.fc_gemm(rt, X, dY, dW, K, N, Nb, 0, 0, K, N, N, actA=act), 20)
        t1 = timeit(ops.fc_wgrad_stream(rt, X, dY, dW, Nb, K, N, actX=act), 20)
        fl = 2.0 * Nb * K * N
        print('FC wgrad Nb=%d K=%d N=%d: fc_gemm %7.2f us (%5.1f TFLOP/s) | fc_wgrad_stream %7.2f us (%5.1f TFLOP/s, %4.0f GB/s of output)' %
              (Nb, K, N, t0, fl / t0 * 1e-6, t1, fl / t1 * 1e-6, 4.0 * K * N / t1 * 1e-3))


def main_wgrad():
    # 1x1 filter gradients: dW[Co][Ci] = dY^T . act(X), reduction over pixels
    for label, Co, Ci, px in (('stage1 a', 16, 64, 131072), ('stage1 c', 64, 16, 131072), ('stage2 a', 32, 128, 32768),
                              ('stage2 c', 128, 32, 32768), ('stage3 a', 64, 256, 8192), ('stage3 c', 256, 64, 8192)):
        tiles = [(16, 64, 1), (32, 64, 1), (64, 16, 4), (64, 64, 4), (64, 32, 4)]
        for tile in tiles:
            if tile[0] > max(16, Co) or tile[1] > max(16, Ci):
                continue
            for sk in (128, 256, 512, 1024, 2048):
                if px // sk < 64:
                    continue
                bench_fc(label, Co, Ci, px, 0, 0, tile, sk)


def main_feats():
    for label, M, N, K, tile in (('stage3/4 conv c 64->256', 8192, 256, 64, (64, 64, 4)), ('stage1 conv c 16->64', 131072, 64, 16, (64, 64, 4)),
                                 ('stage3/4 conv a 256->64', 8192, 64, 256, (64, 16, 4)), ('stage1 conv a 64->16', 131072, 16, 64, (128, 16, 4))):
        for feats in ((), ('relu',), ('act',), ('bias',), ('res',), ('stats',), ('act', 'bias', 'res'), ('act', 'bias', 'res', 'stats'), ('bnbwd',)):
            bench_gemm(M, N, K, True, tile, 0, feats, label)


def main_stride2():
    """The six stride-2 1x1 convolutions of the forward pass (first block of stages 1-3: bottleneck entry and projection shortcut)."""
    for label, M, N, K, hw, feats in (('s1 entry 32->16', 131072, 16, 32, 32, ('act', 'bias', 'stats')), ('s1 proj 32->64', 131072, 64, 32, 32, ('act', 'bias', 'res', 'stats')),
                                      ('s2 entry 64->32', 32768, 32, 64, 16, ('act', 'bias', 'stats')), ('s2 proj 64->128', 32768, 128, 64, 16, ('act', 'bias', 'res', 'stats')),
                                      ('s3 entry 128->64', 8192, 64, 128, 8, ('act', 'bias', 'stats')), ('s3 proj 128->256', 8192, 256, 128, 8, ('act', 'bias', 'res', 'stats'))):
        for tile in ((64, 64, 4), (128, 32, 4), (64, 32, 4), (128, 16, 4), (64, 16, 4), (32, 64, 1)):
            if tile[1] > max(16, N):
                continue
            bench_gemm(M, N, K, True, tile, 0, feats, label + ' s2', stride2=(hw, hw))
        bench_gemm(M, N, K, True, (64, min(64, N), 4) if N >= 64 else (64, 16, 4), 0, feats, label + ' s1 (contiguous rows)')


def main_conv3():
    for label, N, H, C in (('stage1 3x3 16->16', 128, 32, 16), ('stage2 3x3 32->32', 128, 16, 32), ('stage3/4 3x3 64->64', 128, 8, 64),
                           ('256: stage1 16->16', 128, 64, 16), ('256: stage2 32->32', 128, 32, 32), ('256: stage3/4 64->64', 128, 16, 64)):
        X = rt.alloc((N, H, H, C), zero=False)
        rt.tensor(X).normal_()
        Wk = rt.alloc((C, 9, C), zero=False)
        rt.tensor(Wk).normal_()
        Y = rt.alloc((N, H, H, C), zero=False)
        for bm in (64, 128):
            try:
                us = timeit(ops.conv3x3(rt, X, N, H, H, C, Wk, C, Y, bm=bm))
            except Exception as e:      # noqa: BLE001
                print(label, bm, e)
                continue
            px = N * H * H
            print('%-22s bm=%3d plain  %7.2f us  %6.1f TFLOP/s  %6.0f GB/s' % (label, bm, us, 2.0 * px * 9 * C * C / us * 1e-6, 8.0 * px * C / us * 1e-3))
            dY = rt.alloc((N, H, H, C), zero=False)
            rt.tensor(dY).normal_()
            nblk = rt.lib.dpp_conv3x3_wgrad_blocks(N, H, H, C, C, bm)
            part = rt.alloc((nblk, C * 9 * C), zero=False)
            us = timeit(ops.conv3x3_wgrad(rt, X, N, H, H, C, dY, C, part, bm=bm))
            print('%-22s bm=%3d wgrad  %7.2f us  %6.1f TFLOP/s  %6.0f GB/s (%d partial blocks)' %
                  (label, bm, us, 2.0 * px * 9 * C * C / us * 1e-6, 8.0 * px * C / us * 1e-3, nblk))
        mean, scale, beta = rt.alloc(C), rt.alloc(C), rt.alloc(C)
        act = Act(mean.ptr, scale.ptr, beta.ptr, 3, C)
        rows = rt.lib.dpp_conv3x3_stream_rows(N, H, H, C)
        if rows:
            px = N * H * H
            stats = rt.alloc((px // rows, 2, C), zero=False)
            for what, kw in (('plain', {}), ('act+bias+stats', dict(actX=act, bias=mean, epi=ops.epilogue(stats=stats)))):
                us = timeit(ops.conv3x3_stream(rt, X, N, H, H, C, Wk, Y, **kw))
                print('%-22s stream %-15s %7.2f us  %6.1f TFLOP/s  %6.0f GB/s' % (label, what, us, 2.0 * px * 9 * C * C / us * 1e-6, 8.0 * px * C / us * 1e-3))
            bm = 128 if px // 128 >= 512 else 64
            nb = rt.lib.dpp_conv3x3_tiling(N, H, H, bm, None, None, None)
            st2 = rt.alloc((nb, 2, C), zero=False)
            us = timeit(ops.conv3x3(rt, X, N, H, H, C, Wk, C, Y, actX=act, bias=mean, bm=bm, epi=ops.epilogue(stats=st2)))
            print('%-22s tiled  %-15s %7.2f us' % (label, 'act+bias+stats', us))
        for rpw in (64, 128, 256, 512):
            nsl = rt.lib.dpp_wgrad3_stream_slices(C, C, N, H, H, rpw)
            if nsl <= 0 or nsl * C * 9 * C * 4 > 64 << 20:
                continue
            part = rt.alloc((nsl, C * 9 * C), zero=False)
            us = timeit(ops.wgrad3_stream(rt, dY, C, X, C, N, H, H, rpw, part, actX=act))
            red = timeit(ops.reduce_partials(rt, part, nsl, C * 9 * C, Wk))
            print('%-22s rpw=%3d wgrad stream %7.2f us  %6.1f TFLOP/s  %6.0f GB/s (%d slices, %.1f MB; reduce %.2f us)' %
                  (label, rpw, us, 2.0 * px * 9 * C * C / us * 1e-6, 8.0 * px * C / us * 1e-3, nsl, nsl * C * 9 * C * 4e-6, red))


def main_floor():
    """What the small kernels of the critical chain cost on the GPU (kernel + boundary), measured inside a graph chain."""
    ctr = rt.alloc(1, np.int64)
    print('empty kernel (counter_add)                         %6.2f us' % timeit(ops.counter_add(rt, ctr, 1)))
    for C, nb, M in ((64, 128, 8192), (256, 128, 8192), (32, 512, 32768), (128, 512, 32768), (16, 2048, 131072), (64, 2048, 131072)):
        part = rt.alloc((2, C, nb), zero=False)
        rt.tensor(part).normal_()
        v = [rt.alloc(C) for _ in range(6)]
        print('bn_finalize      C=%3d nb=%4d                      %6.2f us' % (
            C, nb, timeit(ops.bn_finalize(rt, part, nb, M, M // nb, C, v[0], 1e-4, v[1], v[2], v[3], v[4], v[5], 0.1))))
        print('bn_bwd_finalize  C=%3d nb=%4d                      %6.2f us' % (
            C, nb, timeit(ops.bn_bwd_finalize(rt, part, nb, M, C, v[0], v[1], v[2], v[3]))))
        G, X, dX = rt.alloc((M, C), zero=False), rt.alloc((M, C), zero=False), rt.alloc((M, C), zero=False)
        us = timeit(ops.bn_bwd_apply(rt, G, X, M, C, v[0], v[1], v[2], v[3], v[4], dX))
        print('bn_bwd_apply     M=%6d C=%3d                     %6.2f us  %6.0f GB/s' % (M, C, us, 12.0 * M * C / us * 1e-3))


def main_expand():
    """Variant 4 (gemm_expand_kernel) against the kernels the engine used before, forward (bias + residual + statistics, BN+ReLU prologue) and
    data gradient (BatchNorm-backward epilogue) of the three expanding shapes."""
    for label, M, N, K, tile0, rpws in (('stage3/4 64->256', 8192, 256, 64, (64, 32, 4), (32, 64)), ('stage2 32->128', 32768, 128, 32, (64, 64, 4), (32, 64, 128)),
                                       ('stage1 16->64', 131072, 64, 16, (64, 64, 4), (64, 128, 256))):
        for feats, b_kc in ((('act', 'bias', 'res', 'stats'), True), (('bnbwd',), False), (('bnbwd', 'res'), False), (('bnbwd', 'res', 'lazy'), False),
                            (('bnbwd', 'res', 'lazykeep'), False)):
            bench_gemm(M, N, K, b_kc, tile0, 0, feats, label + (' fwd' if b_kc else ' dgrad'))
            if K == 16 and not b_kc and 'lazy' not in feats and 'lazykeep' not in feats:
                bench_gemm(M, N, K, b_kc, (64, 64, 4), 3, feats, label + ' dgrad s16')
            for rpw in rpws:
                bench_gemm(M, N, K, b_kc, (rpw, 64, 4), 4, feats, label + (' fwd' if b_kc else ' dgrad'))


def main():
    if 'expand' in sys.argv[1:]:
        return main_expand()
    if 'floor' in sys.argv[1:]:
        return main_floor()
    if 'stride2' in sys.argv[1:]:
        return main_stride2()
    if 'conv3' in sys.argv[1:]:
        return main_conv3()
    if 'feats' in sys.argv[1:]:
        return main_feats()
    if 'fc' in sys.argv[1:]:
        return main_fc()
    if 'fcw' in sys.argv[1:]:
        return main_fcw()
    if 'wgrad' in sys.argv[1:]:
        return main_wgrad()
    z = rt.alloc(4)
    print('launch floor (fill_zero 4 floats): %.2f us' % timeit(ops.fill_zero(rt, z, 4)))
    shapes = [('stage3/4 conv a 256->64', 8192, 64, 256), ('stage3/4 conv c 64->256', 8192, 256, 64),
              ('stage2 conv a 128->32', 32768, 32, 128), ('stage2 conv c 32->128', 32768, 128, 32),
              ('stage1 conv a 64->16', 131072, 16, 64), ('stage1 conv c 16->64', 131072, 64, 16)]
    for label, M, N, K in shapes:
        for variant in (0, 1):
            for tile in ((64, min(64, max(16, N)), 4), (128, min(64, max(16, N)), 4), (64, 16, 4)):
                for feats in ((), ('act', 'bias', 'res', 'stats')):
                    try:
                        bench_gemm(M, N, K, True, tile, variant, feats, label)
                    except Exception as e:      # unsupported tile for the variant
                        print('%-34s tile=%s v%d: %s' % (label, tile, variant, e))


if __name__ == '__main__':
    main()
