#!/usr/bin/env python3
"""Time the ScaleNet (type 1, the CoM-refinement net) train step on the GPU: python tools/scalenet_probe.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hipdp import engine  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402
from net.scalenet import ScaleNet, ScaleNetParams  # noqa: E402

rt = TorchHipRuntime()
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64            # main_nyu_com_refine.py:149
net = ScaleNet(np.random.RandomState(23455), cfgParams=ScaleNetParams(type=1, batchSize=B, numJoints=1, nDims=3))
eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
eng.set_lr(5e-4)
for _ in range(10):
    eng.run_step_plans()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    eng.run_step_plans()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 100
print('ScaleNet type 1 bs%d train step: %.3f ms  (%.0f crops/s), launches %s' % (B, dt * 1e3, B / dt, eng.num_launches()))

if '--ops' in sys.argv:
    # GPU-side time of every launch of the step on its own (50 copies chained in one explicit hipGraph lane, incl. the 1.6 us boundary)
    from hipdp import ops

    def timeit(launch, rep=50):
        plan = ops.NativePlan(rt, [(launch, False)] * rep, mode='graph1')
        plan.run(rt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        plan.run(rt)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / rep
    tot = 0.0
    for ph, plan in (('fwd', eng.fwd), ('loss', eng.lossplan), ('bwd', eng.bwd), ('upd', eng.upd)):
        for l in plan.launches():
            if l.kernels == 1:
                t = timeit(l)
                tot += t
                print('%-5s %-28s %8.2f us' % (ph, l.name, t))
    print('sum of the launches, one by one: %.1f us' % tot)
