#!/usr/bin/env python3
"""Bare bs128 ResNet train step (no augmentation, no bench plumbing), timed over 50 steps:   python tools/step_probe.py"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/deep-prior-pp_amd')
import numpy as np, torch
from net.resnet import ResNet, ResNetParams
from hipdp import engine
from hipdp.runtime import TorchHipRuntime
rt = TorchHipRuntime()
B = 128
net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, batchSize=B, numJoints=1, nDims=30))
eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
eng.set_lr(1e-3)
for _ in range(5): eng.run_step_plans()
torch.cuda.synchronize()
# host issue time vs gpu time
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): eng.run_step_plans()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('issue %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
# single stream (no side) eager
rt.has_side_stream = False
for _ in range(3): eng.run_step_plans()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): eng.run_step_plans()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('single-stream: issue %.3f total %.3f' % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
g = rt.capture(eng.run_step_plans)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print('single-stream graph %.3f' % ((time.perf_counter() - t0) / 10 * 1e3))
rt.has_side_stream = True
g2 = rt.capture(eng.run_step_plans)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g2.replay()
torch.cuda.synchronize(); print('two-stream graph %.3f' % ((time.perf_counter() - t0) / 10 * 1e3))
