"""examples/main_nyu_posereg_embedding.py -- the Python-3 driver for the real NYU layout -- end to end on a tiny dataset written in the
original file format (RGB-packed depth PNGs + joint_data.mat under train / test_1 / test_2): importer, device crops, device PCA prior,
trainer with online augmentation, prior layer, evaluation."""
import importlib.util
import os

import numpy as np
import pytest
import scipy.io
from PIL import Image

from data.importers import NYUImporter
from hipdp import runtime as R
from oracle import augment as A
from tests.backends import BACKENDS, get_runtime
from tests.test_importers import _frames_and_joints

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_nyu(base, name, n, seed):
    cam = A.Camera.nyu()
    frames, gt3D14, gtuvd14 = _frames_and_joints(cam, n, 14, 480, 640, (300., 300., 300.), 13, seed)
    imp = NYUImporter(base, useCache=False)
    xyz, uvd = np.zeros((1, n, 36, 3), np.float32), np.zeros((1, n, 36, 3), np.float32)
    xyz[0][:, imp.restrictedJointsEval], uvd[0][:, imp.restrictedJointsEval] = gt3D14, gtuvd14
    d = os.path.join(base, name)
    os.makedirs(d)
    scipy.io.savemat(os.path.join(d, 'joint_data.mat'), {'joint_xyz': xyz, 'joint_uvd': uvd})
    for i in range(n):
        v = frames[i].astype(np.int32)
        Image.fromarray(np.stack([np.zeros_like(v), v >> 8, v & 255], axis=2).astype(np.uint8)).save(os.path.join(d, 'depth_1_%07d.png' % (i + 1)))


@pytest.mark.parametrize('backend', BACKENDS)
def test_nyu_driver_end_to_end(backend, tmp_path):
    R.set_default_runtime(get_runtime(backend))
    base = str(tmp_path / 'NYU')
    for name, n, seed in (('train', 5, 1), ('test_1', 2, 2), ('test_2', 2, 3)):
        _write_nyu(base, name, n, seed)
    spec = importlib.util.spec_from_file_location('nyu_driver', os.path.join(ROOT, 'examples', 'main_nyu_posereg_embedding.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    net = 'resnet' if backend == 'hip' else 'poseregnet'        # (the 128x128 ResNet is too slow for the SIMT emulator)
    costs, results = mod.main(['--data', base, '--net', net, '--epochs', '1', '--batch', '2', '--embedding', '6', '--prior-poses', '300',
                               '--out', str(tmp_path / 'eval'), '--cache', str(tmp_path / 'cache')])
    assert len(costs) == 3 and np.all(np.isfinite(costs))        # 5 crops padded to 3 minibatches of 2
    assert set(results) == {'test_1', 'test_2'} and all(np.isfinite(v).all() and v[0] > 0 for v in results.values())
    assert os.path.isfile(str(tmp_path / 'eval' / 'network_prior.pkl'))
