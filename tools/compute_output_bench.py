#!/usr/bin/env python3
"""The rate of `net.computeOutput(x)` itself -- host array in, host array out (/root/reference/src/net/netbase.py:217-316) -- next to the device
rate of its compiled function (bench.py forward_only):   python tools/compute_output_bench.py [frames] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hipdp import runtime as R  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rt = TorchHipRuntime()
R.set_default_runtime(rt)
net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=128, hIn=128, batchSize=B, numJoints=1, nDims=30))
net.setDeterministic()
x = np.random.RandomState(3).uniform(-1, 1, (frames, 1, 128, 128)).astype(np.float32)
net.computeOutput(x[:2 * B])
torch.cuda.synchronize()
best = None
for _ in range(3):
    t0 = time.perf_counter()
    out = net.computeOutput(x)
    dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
assert out.shape == (frames, 30) and np.isfinite(out).all()
print('computeOutput: %d frames at batch %d in %.4f s = %.0f crops/s (%.3f ms per batch), host arrays in and out' % (frames, B, best, frames / best, best / (frames / B) * 1e3))
