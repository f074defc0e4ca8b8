#!/usr/bin/env python3
"""What the filter-gradient branch costs the bs128 train step:  python tools/branch_probe.py [--batch 128]

Times the full step plan, then the same plan with groups of launches of the gradient branch REMOVED (timing experiment only:
the gradients of those layers are stale).  The difference between `all` and `no side launches` is what a free gradient branch
would buy; the per-group lines show which kernels of the branch cost the chain the most."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--steps', type=int, default=40)
    a = ap.parse_args()
    import numpy as np
    import torch
    from hipdp import engine, ops
    from hipdp.runtime import TorchHipRuntime
    from net.resnet import ResNet, ResNetParams
    rt = TorchHipRuntime()
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=128, hIn=128, batchSize=a.batch, numJoints=1, nDims=30))
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    eng.set_lr(1e-3)
    full = eng.step_plan()

    def timed(plan):
        for _ in range(5):
            plan.run(rt)
        torch.cuda.synchronize()
        best = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(a.steps):
                plan.run(rt)
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / a.steps * 1e3)
        return min(best)

    def without(pred):
        p = ops.Plan('probe')
        dropped = 0
        for op, side in full.ops:
            if isinstance(op, ops.Launch) and pred(op, side):
                dropped += 1
                continue
            p.ops.append((op, side))
        p.uses_side = full.uses_side
        return p, dropped

    side_names = {}
    for op, side in full.ops:
        if side and isinstance(op, ops.Launch):
            key = op.name.rstrip('0123456789').rstrip('_')
            side_names[key] = side_names.get(key, 0) + 1
    print('launches on the gradient branch:', ', '.join('%s x%d' % kv for kv in sorted(side_names.items())))
    base = timed(full)
    print('%-44s %4d launches  %.3f ms/step' % ('all', len(full), base))
    cases = [('no side launches', lambda op, side: side)]
    for key in sorted(side_names):
        cases.append(('without %s' % key, lambda op, side, key=key: side and op.name.rstrip('0123456789').rstrip('_') == key))
    for label, pred in cases:
        p, dropped = without(pred)
        t = timed(p)
        print('%-44s %4d launches  %.3f ms/step  (%+.3f, %d dropped)' % (label, len(p), t, t - base, dropped))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
