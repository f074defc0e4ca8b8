#!/usr/bin/env python3
"""Frames per second of NYUImporter.loadSequence (PNG decode on the host + device crops in chunks of 256) on a synthetic sequence in
the original file format, with and without the decoding thread pool:   python tools/importer_bench.py [frames]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'deep-prior-pp_amd'))
import numpy as np  # noqa: E402
import scipy.io  # noqa: E402
from PIL import Image  # noqa: E402
from data.importers import DepthImporter, NYUImporter  # noqa: E402
from tools import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 768
base = tempfile.mkdtemp(prefix='nyu_bench_')
di = synth.importer_of('nyu')
frames, coms, gt3d = synth.depth_frames(np.random.RandomState(1), 64, di, 480, 640, (300., 300., 300.), 14)
imp = NYUImporter(base, useCache=False)
xyz, uvd = np.zeros((1, n, 36, 3), np.float32), np.zeros((1, n, 36, 3), np.float32)
os.makedirs(os.path.join(base, 'train'))
for i in range(n):
    k = i % 64
    xyz[0, i, imp.restrictedJointsEval] = gt3d[k] + di.jointImgTo3D(coms[k])
    uvd[0, i, imp.restrictedJointsEval] = di.joints3DToImg(xyz[0, i, imp.restrictedJointsEval])
    v = frames[k].astype(np.int32)
    Image.fromarray(np.stack([np.zeros_like(v), v >> 8, v & 255], axis=2).astype(np.uint8)).save(os.path.join(base, 'train', 'depth_1_%07d.png' % (i + 1)))
scipy.io.savemat(os.path.join(base, 'train', 'joint_data.mat'), {'joint_xyz': xyz, 'joint_uvd': uvd})
imp.loadSequence('train', Nmax=64)                      # warm-up (library load, first kernels)
real = DepthImporter._read_ahead
for workers in (1, 8, 16):
    DepthImporter._read_ahead = lambda self, entries, window=32, workers=workers: real(self, entries, window, workers)
    t0 = time.perf_counter()
    seq = imp.loadSequence('train')
    dt = time.perf_counter() - t0
    print('%2d decoding threads: %d frames in %.2f s = %.0f frames/s' % (workers, len(seq.data), dt, len(seq.data) / dt))
