#!/usr/bin/env python3
"""BASELINE config 3: the fused online augmentation (rot / scale / trans / CoM-jitter) measured alone on ICVL geometry,
batch 256: crops/s and GB/s of the ONE fused launch (dpp_augment: prepare + warp + draw-counter advance), next to the oracle's
NumPy restatement of NetTrainer.augmentCrop on one host core (the reference runs it in 8 worker processes, nettrainer.py:59).
Timed on the GPU side: the launch is replayed `iters` times as one hipGraph chain (the figure includes the 1.6 us kernel boundary).
   python tools/augment_bench.py [--batch 256] [--iters 200] [--modes com,rot,sc,none] [--splits 0]"""
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
    ap.add_argument('--splits', type=int, default=0, help='workgroups per crop (0 = the library chooses)')
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
    modes = args.modes.split(',')
    table = rt.upload(np.array([MODE_CODE[m] for m in modes], np.int32))
    x_out, y_out = rt.alloc((B, 128, 128)), rt.alloc((B, E))
    camt = (cam.fx, cam.fy, cam.ux, cam.uy, cam.flip_y)
    st = ops.AugmentState(rt, B, seed=1234)
    (launch,) = st.ops(im, co, cu, mm, gt, J, 128, camt, x_out, y_out, mode_table=table, n_modes=len(modes), pca_mean=pm, pca_comp=pc, E=E,
                       splits=args.splits)
    plan = ops.NativePlan(rt, [(launch, False)] * args.iters, mode='graph1')
    plan.run(rt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        plan.run(rt)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * args.iters)
    assert int(st.counter.get()[0]) == 4 * args.iters            # every replayed launch drew fresh parameters
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
               us_per_batch=round(us, 2), launches_per_batch=1, splits=args.splits, algorithmic_bytes_per_crop=131072,
               roofline=dict(bound='hbm', achieved=round(B * 131072 / (us * 1e-6) / 1e9, 1), peak=8000.0, unit='GB/s',
                             frac=round(B * 131072 / (us * 1e-6) / 8e12, 4)),
               cpu_baseline=dict(value=round(cpu, 1), unit='depth-crops/sec', cores=1, kind='port',
                                 sample='%d crops through the NumPy restatement of augmentCrop (oracle/augment.py)' % n))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
