#!/usr/bin/env python3
"""Host cost per kernel launch and step time of the bs128 ResNet train step, by launch mode:
   python tools/launch_rate.py [--steps 30] > profiles/r02_launch_rate.txt

For each mode (python: one ctypes call per launch from the interpreter; native: the recorded plan issued from C++,
dpp_plan_run; graph / graph1: explicit hipGraph, two lanes / one chain) two numbers:
  * host us per launch with the stream saturated -- the wall time of the issuing loop alone (no synchronisation until the
    end, the queue never drains), divided by the launches issued: if this is above the GPU's time per launch, the step is
    host-bound and every dependent kernel waits for its successor to ARRIVE;
  * ms per step on the GPU (synchronised total / steps).
Also the 1-stream variants (DPP_NO_SIDE_STREAM=1) to separate the cost of the second stream from the cost of issuing."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))


def worker(mode, steps, batch):
    os.environ['DPP_LAUNCH_MODE'] = mode
    import numpy as np
    import torch
    from hipdp import engine
    from hipdp.runtime import TorchHipRuntime
    from net.resnet import ResNet, ResNetParams
    rt = TorchHipRuntime()
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=128, hIn=128, batchSize=batch, numJoints=1, nDims=30))
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    eng.set_lr(1e-3)
    n = sum(eng.num_launches().values()) + len(eng.lossplan)
    for _ in range(5):
        eng.run_step_plans()
    torch.cuda.synchronize()
    # (a) one step issued into an EMPTY queue (synchronised before): the host's own cost per launch, nothing throttles it
    t_one = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_step_plans()
        t_one.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t_one = sorted(t_one)[len(t_one) // 2]
    # (b) many steps back to back: once the queue is full the issuing loop runs at the GPU's pace
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run_step_plans()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('%-8s side_stream=%d  launches/step %d  host %.2f us/launch into an empty queue (%.3f ms/step), %.2f us/launch saturated  '
          'GPU %.3f ms/step  -> %s' % (mode, int(rt.has_side_stream), n, t_one / n * 1e6, t_one * 1e3, t_issue / steps / n * 1e6,
                                        t_all / steps * 1e3, 'host-bound' if t_one > 0.9 * t_all / steps else 'GPU-bound'))
    sys.stdout.flush()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--worker', default=None)
    a = ap.parse_args()
    if a.worker:
        worker(a.worker, a.steps, a.batch)
    else:
        for side in ('0', '1'):
            for mode in ('python', 'native', 'graph', 'graph1'):
                if mode == 'graph1' and side == '1':
                    continue
                env = dict(os.environ, DPP_NO_SIDE_STREAM=side)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', mode, '--steps', str(a.steps), '--batch', str(a.batch)],
                                   env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                out = r.stdout.decode().strip()
                print(out if out else '%-8s side_stream=%d FAILED rc=%d: %s' % (mode, 1 - int(side), r.returncode, r.stderr.decode()[-300:]))
                sys.stdout.flush()
