#!/usr/bin/env python3
"""Phase stamps of the fused augmentation launch (profiling build: make -C deep-prior-pp_amd/csrc prof): per workgroup
0 entry -> 1 per-sample geometry done on lane 0 -> 2 barrier (crop maximum done on waves 1-3) -> 3 labels / projection -> 4 pixels done.
   python tools/augment_phase.py [--batch 256] [--modes com,rot,sc,none]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hipdp import ops  # noqa: E402
from hipdp.augmenter import MODE_CODE  # noqa: E402
from hipdp.runtime import TorchHipRuntime  # noqa: E402
from tools import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--modes', default='com,rot,sc,none')
args = ap.parse_args()
rt = TorchHipRuntime(lib_path=os.path.join(ROOT, 'deep-prior-pp_amd', 'lib_prof', 'libdpp_hip.so'))
B, J, E = args.batch, 16, 30
di, imgs, coms, cubes, Ms, gts, pm, pc = synth.crop_db(B, 128, J, dataset='icvl', cube=(250., 250., 250.))
f32 = lambda a: rt.upload(np.ascontiguousarray(a, np.float32))       # noqa: E731
from hipdp.augmenter import camera_tuple  # noqa: E402
for modes in [m.split('+') for m in args.modes.replace(',', '+').split(';')] if ';' in args.modes else [args.modes.split(',')] + [[m] for m in args.modes.split(',')]:
    table = rt.upload(np.array([MODE_CODE[m] for m in modes], np.int32))
    x_out, y_out = rt.alloc((B, 128, 128)), rt.alloc((B, E))
    st = ops.AugmentState(rt, B, seed=1234)
    (launch,) = st.ops(f32(imgs), f32(coms), f32(cubes), f32(Ms.reshape(B, 9)), f32(gts), J, 128, camera_tuple(di), x_out, y_out, mode_table=table,
                       n_modes=len(modes), pca_mean=f32(pm), pca_comp=f32(pc), E=E)
    nwg = 4096
    buf = rt.alloc((nwg + 8, 16), np.int64)
    for _ in range(5):
        launch(rt.stream)
    torch.cuda.synchronize()
    rt.lib.dpp_prof_set(buf.ptr)
    launch(rt.stream)
    torch.cuda.synchronize()
    rt.lib.dpp_prof_set(None)
    t = buf.get()[:, :5].astype(np.float64) * 0.01
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    ph = np.diff(t, axis=1)
    print('%-22s %4d WGs  span %6.2f us  start skew med %5.2f max %5.2f | geometry %5.2f/%5.2f  wait-max %5.2f/%5.2f  labels %5.2f/%5.2f  pixels %5.2f/%5.2f  (med/max)' % (
        ','.join(modes), len(t), t[:, 4].max() - t0, np.median(t[:, 0] - t0), (t[:, 0] - t0).max(),
        np.median(ph[:, 0]), ph[:, 0].max(), np.median(ph[:, 1]), ph[:, 1].max(), np.median(ph[:, 2]), ph[:, 2].max(), np.median(ph[:, 3]), ph[:, 3].max()))
