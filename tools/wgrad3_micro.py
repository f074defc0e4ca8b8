#!/usr/bin/env python3
"""The 3x3 filter-gradient launch of the 16- / 32-channel layers alone, at the shapes of the bs128 steps (128x128 and 256x256 input, float32
and bf16-stored operands): us per launch (replayed as one hipGraph chain between two HIP events: includes the kernel boundary), TFLOP/s,
GB/s of the algorithmic bytes.  DPP_WGRAD3_T=0 selects the round-1 kernel (conv3x3_wgrad_kernel) for an A/B in a second process.
   python tools/wgrad3_micro.py [--iters 50]"""
import argparse
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


def bits(a):
    """float32 -> bfloat16 bit patterns (uint16), round to nearest even: a bf16-stored tensor"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--precision', type=int, default=0)
    args = ap.parse_args()
    rt = TorchHipRuntime()
    rng = np.random.RandomState(1)
    print('DPP_WGRAD3_T=%s' % os.environ.get('DPP_WGRAD3_T', '(default: on)'))
    for (N, H, W, C, b16) in ((128, 32, 32, 16, False), (128, 16, 16, 32, False), (128, 64, 64, 16, True), (128, 32, 32, 32, True),
                              (128, 64, 64, 16, False), (256, 32, 32, 16, False)):
        x = rng.normal(size=(N, H, W, C)).astype(np.float32)
        dy = rng.normal(size=(N, H, W, C)).astype(np.float32)
        X, dY = rt.upload(x), rt.upload(dy)
        if b16:
            X, dY = rt.upload(bits(x)), rt.upload(bits(dy))
        mean, scale, beta = (rt.upload(v.astype(np.float32)) for v in (rng.normal(size=C) * 0.3, rng.uniform(0.5, 1.5, C), rng.normal(size=C) * 0.3))
        act = ops.act(Act.BN_RELU, mean, scale, beta, C)
        nblk = rt.lib.dpp_conv3x3_wgrad_blocks(N, H, W, C, C, 128)
        part = rt.alloc((nblk, C, 9, C), zero=False)
        kw = dict(precision=args.precision) if args.precision else {}
        launch = ops.conv3x3_wgrad(rt, X, N, H, W, C, dY, C, part, actX=act, bm=128, **kw)
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
        flops = 2.0 * px * 9 * C * C
        byts = px * C * 2 * (2 if b16 else 4) + nblk * 9.0 * C * C * 4
        print('N %3d  %3dx%-3d  C %2d  %s  slices %3d : %7.2f us  %6.1f TFLOP/s  %7.1f GB/s' % (
            N, H, W, C, 'bf16-stored' if b16 else 'f32        ', nblk, us, flops / us / 1e6, byts / us / 1e3))


if __name__ == '__main__':
    main()
