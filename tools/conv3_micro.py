#!/usr/bin/env python3
"""The LDS-tiled 3x3 convolution alone (forward form: BatchNorm+ReLU prologue, bias, statistics partials; data-gradient form: the
BatchNorm-backward mask + sums of the epilogue) at the shapes of the bs128 steps: us per launch (replayed as one hipGraph chain between two
HIP events: includes the kernel boundary) and GB/s of the algorithmic bytes.  Environment: DPP_C3_PERSIST (workgroups of the tile-walking form,
0 = one workgroup per tile), DPP_C3_BM (rows per workgroup).
   python tools/conv3_micro.py [--iters 50] [--phases]     (--phases: the profiling build's phase stamps, conv3x3_kernel only)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hipdp import heuristics as hz  # noqa: E402
from hipdp import ops  # noqa: E402
from hipdp.lib import Act  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402


def bits(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


class _BN(object):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--phases', action='store_true')
    ap.add_argument('--bm', type=int, default=0)
    args = ap.parse_args()
    rt = TorchHipRuntime(lib_path=os.path.join(ROOT, 'deep-prior-pp_amd', 'lib_prof', 'libdpp_hip.so')) if args.phases else TorchHipRuntime()
    rng = np.random.RandomState(1)
    print('DPP_C3_PERSIST=%s bm=%s' % (os.environ.get('DPP_C3_PERSIST', '(default)'), args.bm or '(heuristic)'))
    cases = ((128, 32, 32, 16, False, 0), (128, 16, 16, 32, False, 0), (128, 8, 8, 64, False, 0),
             (128, 64, 64, 16, True, 1), (128, 32, 32, 32, True, 1), (128, 16, 16, 64, True, 1), (128, 64, 64, 16, False, 0))
    for (N, H, W, C, b16, prec) in cases:
        up = (lambda v: rt.upload(bits(v))) if b16 else (lambda v: rt.upload(np.ascontiguousarray(v, np.float32)))
        X = up(rng.normal(size=(N, H, W, C)))
        Xb = up(rng.normal(size=(N, H, W, C)))
        Wk = rt.upload((rng.normal(size=(C, 9, C)) * 0.1).astype(np.float32))
        Y = rt.alloc((N, H, W, C), np.uint16 if b16 else np.float32, zero=False) if b16 else rt.alloc((N, H, W, C), zero=False)
        vec = lambda lo, hi: rt.upload(rng.uniform(lo, hi, C).astype(np.float32))  # noqa: E731
        mean, scale, beta, bias = vec(-.3, .3), vec(.5, 1.5), vec(-.3, .3), vec(-.1, .1)
        act = ops.act(Act.BN_RELU, mean, scale, beta, C)
        bm = args.bm or hz.conv3x3_bm(N * H * W, C, hw=(H, W), prec=prec)
        nblk = rt.lib.dpp_conv3x3_tiling(N, H, W, bm, None, None, None)
        stats = rt.alloc((nblk, 2, C), zero=False)
        bn = _BN()
        bn.mean, bn.inv_std, bn.scale, bn.beta_buf = mean, scale, scale, beta
        part = rt.alloc((nblk, 2, C), zero=False)
        forms = (('forward ', dict(actX=act, bias=bias, epi=ops.epilogue(stats=stats))),
                 ('datagrad', dict(epi=ops.epilogue(bn=bn, bn_x=Xb, bn_relu=True, bn_partial=part))))
        for (label, kw) in forms:
            launch = ops.conv3x3(rt, X, N, H, W, C, Wk, C, Y, bm=bm, precision=prec, **kw)
            plan = ops.NativePlan(rt, [(launch, False)] * args.iters, mode='graph1')
            plan.run(rt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            for _ in range(3):
                plan.run(rt)
            e1.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (3 * args.iters)
            px = float(N) * H * W
            es = 2 if b16 else 4
            byts = px * C * es * (2 if label == 'forward ' else 3)
            line = 'N %3d  %3dx%-3d  C %2d  %s %s bm %3d tiles %4d : %7.2f us  %7.1f GB/s' % (
                N, H, W, C, 'bf16' if b16 else 'f32 ', label, bm, nblk, us, byts / us / 1e3)
            if args.phases:
                buf = rt.alloc((nblk * 4 + 8, 16), np.int64)
                rt.lib.dpp_prof_set(buf.ptr)
                launch(rt.stream)
                torch.cuda.synchronize()
                rt.lib.dpp_prof_set(None)
                t = buf.get()[:, :5].astype(np.float64) * 0.01
                t = t[t[:, 4] > 0]
                if len(t):
                    t0 = t[:, 0].min()
                    ph = np.diff(t, axis=1)
                    st = np.sort(t[:, 0] - t0)
                    live = [(int(((t[:, 0] - t0) <= x) .sum()) - int(((t[:, 4] - t0) <= x).sum())) for x in (3.0, 10.0, 20.0)]
                    line += ' | %d WGs span %6.2f  start med %5.2f max %5.2f  started by 3 us: %d  resident at 3/10/20 us: %s  phases med/max: %s' % (
                        len(t), t[:, 4].max() - t0, np.median(t[:, 0] - t0), (t[:, 0] - t0).max(), int((st <= 3.0).sum()), live,
                        '  '.join('%5.2f/%5.2f' % (np.median(ph[:, k]), ph[:, k].max()) for k in range(4)))
            print(line)
            sys.stdout.flush()


if __name__ == '__main__':
    main()
