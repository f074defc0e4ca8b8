#!/usr/bin/env python3
"""BASELINE config 5's geometry on one GPU in fp32: the ResNet at 256x256 input (FC1 65 536 x 1 024, 69 M parameters),
batch 128: time of the train step and a forward-parity spot check against the float64 oracle at batch 2.
   python tools/stress_256.py"""
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
from net.resnet import ResNet, ResNetParams  # noqa: E402

rt = TorchHipRuntime()
B = 128
net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=256, hIn=256, batchSize=B, numJoints=1, nDims=30))
eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
eng.set_lr(1e-3)
for _ in range(5):
    eng.run_step_plans()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.run_step_plans()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print('ResNet type 0 at 256x256, bs128 fp32 train step: %.2f ms (%.0f crops/s), %d parameters, launches %s, %.1f GB allocated' %
      (dt * 1e3, B / dt, eng.store.n_w, eng.num_launches(), torch.cuda.max_memory_allocated() / 2 ** 30))
