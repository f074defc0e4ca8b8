#!/usr/bin/env python3
"""BASELINE config 3: the fused online augmentation (rot / scale / trans / CoM-jitter) measured alone on ICVL geometry,
batch 256: crops/s and GB/s of the two kernels (augment_prepare + augment_warp), next to the oracle's NumPy restatement of
NetTrainer.augmentCrop on one host core (the reference runs it in 8 worker processes, nettrainer.py:59).
   python tools/augment_bench.py [--batch 256] [--iters 200] [--modes com,rot,sc,none]"""
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
from hipdp.augmenter import MODE_CODE  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402
from oracle import augment as A  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--modes', default='com,rot,sc,none')
    ap.add_argument('--cpu-crops', type=int, default=64)
    args = ap.parse_args()
    rt = TorchHipRuntime()
    B, J, E = args.batch, 16, 30
    rng = np.random.RandomState(23455)
    cam = A.Camera.icvl()
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, B, cam, cube=(250., 250., 250.), joints=J)
    pca_mean = rng.normal(0, 0.05, J * 3).astype(np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(J * 3, E)))
    f32 = lambda a: rt.upload(np.ascontiguousarray(a, np.float32))       # noqa: E731
    im, co, cu, mm, gt = f32(imgs), f32(coms), f32(cubes), f32(Ms.reshape(B, 9)), f32(gts)
    pm, pc = f32(pca_mean), f32(q.T)
    rec = rt.alloc(B * rt.lib.dpp_augment_record_bytes(), np.uint8)
    modes = args.modes.split(',')
    table = rt.upload(np.array([MODE_CODE[m] for m in modes], np.int32))
    ctr = rt.alloc(1, np.int64)
    x_out, y_out = rt.alloc((B, 128, 128)), rt.alloc((B, E))
    camt = (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y)
    launches = [ops.augment_prepare(rt, im, co, cu, mm, gt, B, J, 128, camt, rec, y_out, mode_table=table, n_modes=len(modes), seed=1234,
                                    counter=0, pca_mean=pm, pca_comp=pc, E=E, counter_dev=ctr),
                ops.augment_warp(rt, im, rec, B, 128, x_out), ops.counter_add(rt, ctr, 1)]
    for _ in range(10):
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
    # one host core, the oracle's NumPy restatement of the same per-crop arithmetic
    n = min(args.cpu_crops, B)
    mi, offs, rots, scs = A.draw_params(np.random.RandomState(7), n, len(modes))
    t0 = time.perf_counter()
    for i in range(n):
        A.augment_crop(imgs[i].copy(), gts[i].copy(), cam.joint3DToImg(coms[i]), cubes[i], Ms[i], modes[mi[i]], offs[i], rots[i], scs[i],
                       cam, abs(cam.fx), abs(cam.fy))
    cpu = n / (time.perf_counter() - t0)
    res = dict(metric='augmented depth-crops/sec (fused rot/scale/trans/CoM-jitter kernel alone)', value=round(B / (us * 1e-6), 1),
               unit='depth-crops/sec', config=dict(workload='ICVL geometry, 16 joints, batch %d, modes %s, 128x128 crops, PCA projection to 30-D' % (B, modes)),
               us_per_batch=round(us, 2), launches_per_batch=3, algorithmic_bytes_per_crop=131072,
               roofline=dict(bound='hbm', achieved=round(B * 131072 / (us * 1e-6) / 1e9, 1), peak=8000.0, unit='GB/s',
                             frac=round(B * 131072 / (us * 1e-6) / 8e12, 4)),
               cpu_baseline=dict(value=round(cpu, 1), unit='depth-crops/sec', cores=1, kind='port',
                                 sample='%d crops through the NumPy restatement of augmentCrop (oracle/augment.py)' % n))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
