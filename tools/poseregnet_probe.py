#!/usr/bin/env python3
"""Time the PoseRegNet (type 0) bs128 train step on the GPU: python tools/poseregnet_probe.py"""
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
from net.poseregnet import PoseRegNet, PoseRegNetParams  # noqa: E402

rt = TorchHipRuntime()
B = 128
net = PoseRegNet(np.random.RandomState(23455), cfgParams=PoseRegNetParams(type=0, batchSize=B, numJoints=1, nDims=30))
eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
eng.set_lr(1e-3)
for _ in range(5):
    eng.run_step_plans()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    eng.run_step_plans()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print('PoseRegNet type 0 bs128 train step: %.3f ms  (%.0f crops/s), launches fwd %d bwd %d upd %d' %
      (dt * 1e3, B / dt, len(eng.fwd), len(eng.bwd), len(eng.upd)))

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
    for ph, plan in (('fwd', eng.fwd), ('loss', eng.lossplan), ('bwd', eng.bwd), ('upd', eng.upd)):
        for l in plan.launches():
            if l.kernels == 1:
                print('%-5s %-28s %8.2f us' % (ph, l.name, timeit(l)))
