#!/usr/bin/env python3
"""Run a few deterministic forward passes (computeOutput's device function, CompiledNet(train=False)) of the bs128 NYU ResNet for
rocprofv3 --kernel-trace:
   rocprofv3 --kernel-trace -d gpurun_out/prof -o fwd -- python tools/forward_profile.py [batches] [size] [f32|bf16]
Without rocprofv3 it prints the HIP-event time per batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hipdp import engine  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
bf16 = len(sys.argv) > 3 and sys.argv[3] == 'bf16'
B = int(os.environ.get('DPP_FWD_BATCH', '128'))
rt = TorchHipRuntime()
net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=size, hIn=size, batchSize=B, numJoints=1, nDims=30))
eng = engine.CompiledNet(net, train=False, runtime=rt, bf16=bf16)
rng = np.random.RandomState(3)
eng.set_input(rng.uniform(-1, 1, (B, 1, size, size)).astype(np.float32))
for _ in range(3):
    eng.fwd.run(rt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(torch.cuda.current_stream())
for _ in range(batches):
    eng.fwd.run(rt)
e1.record(torch.cuda.current_stream())
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / batches
print('forward only: %.4f ms per batch of %d = %.0f crops/s, %d launches' % (ms, B, B / ms * 1e3, len(eng.fwd)))
