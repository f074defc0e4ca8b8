#!/usr/bin/env python3
"""The initial-crop kernels alone: full 640x480 NYU-sized depth frames -> normalised 128x128 crops.
   python tools/crop_bench.py [--batch 256]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hipdp import ops  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402
from oracle import augment as A  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--iters', type=int, default=50)
args = ap.parse_args()
rt = TorchHipRuntime()
B, H, W = args.batch, 480, 640
cam = A.Camera.nyu()
frames1, coms1 = A.synthetic_frames(np.random.RandomState(1), 8, cam, H, W, (300., 300., 300.))
frames = np.tile(frames1, (B // 8 + 1, 1, 1))[:B]
coms = np.tile(coms1, (B // 8 + 1, 1))[:B]
fr, co = rt.upload(frames), rt.upload(coms)
cu = rt.upload(np.tile(np.float32([300., 300., 300.]), (B, 1)))
rec = rt.alloc(B * rt.lib.dpp_crop_record_bytes(), np.uint8)
out, M = rt.alloc((B, 128, 128), zero=False), rt.alloc((B, 9), zero=False)
launches = [ops.crop_prepare(rt, fr, B, H, W, co, cu, abs(cam.fx), abs(cam.fy), 128, rec, M), ops.crop_warp(rt, fr, rec, B, H, W, 128, out)]
for _ in range(5):
    for o in launches:
        o(rt.stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(args.iters):
    for o in launches:
        o(rt.stream)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / args.iters
t0 = time.perf_counter()
n = 8
for i in range(n):
    d, _, _ = A.detector_preprocess(frames[i])
    c, _, _ = A.crop_area_3d(d, coms[i], (300., 300., 300.), abs(cam.fx), abs(cam.fy))
    A.normalize_crop(c, coms[i][2], 300.)
cpu = n / (time.perf_counter() - t0)
byts = B * (H * W * 4 + 128 * 128 * 4)          # one pass over the frame (valid range) + the crop written
print(json.dumps(dict(metric='depth frames cropped / sec (cropArea3D + normalisation)', value=round(B / (us * 1e-6), 1), unit='frames/sec',
                      config=dict(workload='%d NYU-sized 640x480 frames -> 128x128 crops, cube 300 mm' % B), us_per_batch=round(us, 2),
                      roofline=dict(bound='hbm', achieved=round(byts / (us * 1e-6) / 1e9, 1), peak=8000.0, unit='GB/s',
                                    frac=round(byts / (us * 1e-6) / 8e12, 4)),
                      cpu_baseline=dict(value=round(cpu, 1), unit='frames/sec', cores=1, kind='port',
                                        sample='%d frames through the NumPy restatement (oracle/augment.py crop_area_3d)' % n))))
