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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64            # main_nyu_com_refine.py:149
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
