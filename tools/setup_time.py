#!/usr/bin/env python3
"""One-off set-up steps of the training scripts on the device: pose sampling (20 k and 1e6 poses), PCA fit, evaluation of 8 252 frames:
   python tools/setup_time.py"""
import sys, os, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np, torch
from tools import synth
from util.pcaprior import DevicePCA, sample_random_poses_device
from util.handpose_evaluation import DeviceHandposeEvaluation
di, imgs, coms, cubes, Ms, gts, pm, pc = synth.crop_db(4096, 128, 14)
rng = np.random.RandomState(1)
for n in (20000, 1000000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    P = sample_random_poses_device(di, rng, gts, coms, cubes, n, ['com', 'rot', 'none'], keep_on_device=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    pca = DevicePCA(n_components=30).fit(P)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('n=%d: sampling %.1f ms (host draws + kernel), PCA fit %.1f ms' % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
gt = np.random.RandomState(2).normal(0, 30, (8252, 14, 3)).astype('float32'); pr = gt + np.random.RandomState(3).normal(0, 5, gt.shape).astype('float32')
t0 = time.perf_counter(); ev = DeviceHandposeEvaluation(gt, pr); m = ev.getMeanError(); mx = ev.getMaxError(); t1 = time.perf_counter()
print('evaluation of 8252 frames: %.1f ms (mean %.2f max %.2f)' % ((t1 - t0) * 1e3, m, mx))
