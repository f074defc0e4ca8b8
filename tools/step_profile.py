#!/usr/bin/env python3
"""Run a few single-stream train steps of the bs128 NYU ResNet for rocprofv3 --kernel-trace (the two-stream schedule hides
which kernels are on the critical path, so profiles of kernel durations are taken with the side stream disabled):
   DPP_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace -d gpurun_out/prof -o step -- python tools/step_profile.py [steps] [size] [f32|bf16]
size 256 with bf16 is BASELINE config 5's train step (bf16 MFMA operands, bf16-stored activations and gradients)."""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
bf16 = len(sys.argv) > 3 and sys.argv[3] == 'bf16'
rt = TorchHipRuntime()
net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=size, hIn=size, batchSize=128, numJoints=1, nDims=30))
eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), bf16=bf16)
eng.set_lr(1e-3)
for _ in range(steps):
    eng.run_step_plans()
torch.cuda.synchronize()
