#!/usr/bin/env python3
"""Smoke-check the data-parallel code path on ONE GPU: a 1-rank RCCL process group runs the bucketed gradient all-reduce
(early FC1 bucket from the side stream, wait + rest before ADAM) exactly as an N-rank job would, and the weights after a
few steps must equal those of the engine without data parallelism.   python tools/dp_single_rank_check.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from hipdp import engine, parallel  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402
from net.resnet import ResNet, ResNetParams  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)       # lazily connected, like hipdp.parallel.init_from_env
rt = TorchHipRuntime()
B = 128
rng = np.random.RandomState(3)
x = rng.uniform(-1, 1, (B, 1, 128, 128)).astype(np.float32)
y = rng.normal(0, 0.3, (B, 30)).astype(np.float32)
W = []
for use_dp in (False, True):
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, batchSize=B, numJoints=1, nDims=30))
    dp = parallel.DataParallel(rt, sync_bn=False) if use_dp else None
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), dp=dp)
    if use_dp:
        assert eng._early_slice is not None, "early bucket not planned"
    for _ in range(3):
        eng.train_step(x, y, 1e-3)
    for _ in range(20):                      # the clocks need ~100 ms of continuous work to ramp up after the host syncs above
        eng.run_step_plans()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eng.run_step_plans()
    torch.cuda.synchronize()
    print('dp=%s: %.3f ms/step, upd plan: %s' % (use_dp, (time.perf_counter() - t0) / 50 * 1e3, [o.name for o in eng.upd.steps()]))
    W.append(eng.store.w.get().copy())
print('max |w_dp - w_single| = %.3e' % np.abs(W[0] - W[1]).max())
assert np.array_equal(W[0], W[1])      # same number of steps on both engines
dist.destroy_process_group()
print('OK')
