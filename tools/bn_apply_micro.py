#!/usr/bin/env python3
"""bn_bwd_apply alone at the BatchNorm shapes of the bs128 step (float32, 128x128 input), rows per workgroup swept: us per launch (one hipGraph
chain between two HIP events) and GB/s of the algorithmic bytes (G, X in; dX out).
   python tools/bn_apply_micro.py [--iters 50]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hipdp import ops  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    args = ap.parse_args()
    rt = TorchHipRuntime()
    for (M, C) in ((131072, 64), (131072, 16), (32768, 128), (32768, 32), (8192, 256), (8192, 64)):
        G, X, dX = (rt.alloc((M, C), zero=False) for _ in range(3))
        for b in (G, X):
            rt.tensor(b).normal_()
        vec = [rt.alloc(C, zero=False) for _ in range(5)]
        for v in vec:
            rt.tensor(v).uniform_(0.5, 1.5)
        line = 'M %6d  C %3d :' % (M, C)
        for rpb in (16, 32, 64, 128, 256):
            nb = -(-M // rpb)
            cs = rt.alloc((nb, C), zero=False)
            launch = ops.bn_bwd_apply(rt, G, X, M, C, vec[0], vec[1], vec[2], vec[3], vec[4], dX, rpb=rpb, colsum=cs)
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
            line += '  rpb %3d (%4d wgs) %6.2f us %5.0f GB/s |' % (rpb, nb, us, 12.0 * M * C / us / 1e3)
        print(line)
        sys.stdout.flush()


if __name__ == '__main__':
    main()
