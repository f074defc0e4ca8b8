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
from hipdp import heuristics  # noqa: E402
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


def _write_msra(base, subject, n, seed, cube):
    import struct
    cam = A.Camera.msra()
    frames, gt3D, _ = _frames_and_joints(cam, n, 21, 240, 320, cube, 5, seed)
    d = os.path.join(base, subject, '1')
    os.makedirs(d)
    with open(os.path.join(d, 'joint.txt'), 'w') as f:
        f.write('%d\n' % n)
        for i in range(n):
            j = gt3D[i].copy()
            j[:, 2] *= -1.                                       # the files hold -z
            f.write(' '.join('%.4f' % v for v in j.reshape(-1)) + '\n')
            ys, xs = np.nonzero(frames[i])
            top, bottom, left, right = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
            with open(os.path.join(d, '%06d_depth.bin' % i), 'wb') as fb:
                fb.write(struct.pack('6i', 320, 240, left, top, right, bottom))
                frames[i][top:bottom, left:right].astype(np.float32).tofile(fb)


def _load_driver(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'examples', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('backend', BACKENDS)
def test_msra_crossval_driver_end_to_end(backend, tmp_path):
    """examples/main_msra15_posereg_embedding_crossval.py (BASELINE configs[3]) on three tiny subjects in the original file format
    (<subject>/<gesture>/joint.txt + NNNNNN_depth.bin): every fold trains on the other two, appends its prior, tests on the third."""
    R.set_default_runtime(get_runtime(backend))
    base = str(tmp_path / 'MSRA15')
    for k, (subj, cube) in enumerate((('P0', (200., 200., 200.)), ('P3', (180., 180., 180.)), ('P8', (150., 150., 150.)))):
        _write_msra(base, subj, 2, 10 + k, cube)
    net = 'resnet' if backend == 'hip' else 'poseregnet'
    costs, results = _load_driver('main_msra15_posereg_embedding_crossval').main(
        ['--data', base, '--subjects', 'P0,P3,P8', '--net', net, '--epochs', '1', '--batch', '2', '--embedding', '6', '--prior-poses', '200',
         '--out', str(tmp_path / 'eval'), '--cache', str(tmp_path / 'cache')])
    assert set(results) == {'P0', 'P3', 'P8', 'all'} and all(np.isfinite(v).all() and v[0] > 0 for v in results.values())
    assert all(len(c) == 2 and np.all(np.isfinite(c)) for c in costs.values())          # 4 training crops = 2 minibatches of 2 per fold
    assert all(os.path.isfile(str(tmp_path / 'eval' / ('network_prior_%d.pkl' % i))) for i in range(3))


@pytest.mark.parametrize('backend', BACKENDS)
def test_nyu_com_refine_driver_then_cascade(backend, tmp_path):
    """examples/main_nyu_com_refine.py trains the ScaleNet on a tiny NYU-format dataset and writes net_ScaleNet.pkl; the pose
    regression driver then takes that checkpoint (--refine) and crops every frame through the device cascade."""
    R.set_default_runtime(get_runtime(backend))
    base = str(tmp_path / 'NYU')
    for name, n, seed in (('train', 4, 1), ('test_1', 2, 2), ('test_2', 2, 3)):
        _write_nyu(base, name, n, seed)
    costs, results = _load_driver('main_nyu_com_refine').main(['--data', base, '--epochs', '1', '--batch', '4', '--out', str(tmp_path / 'refine'),
                                                               '--cache', str(tmp_path / 'cache_refine')])
    assert len(costs) == 2 and np.all(np.isfinite(costs))           # 4 frames x (annotated centre, centre of mass) = 2 minibatches of 4
    assert set(results) == {'test_1', 'test_2'} and all(np.isfinite(v).all() for v in results.values())
    ckpt = str(tmp_path / 'refine' / 'net_ScaleNet.pkl')
    assert os.path.isfile(ckpt)
    net = 'resnet' if backend == 'hip' else 'poseregnet'
    costs2, results2 = _load_driver('main_nyu_posereg_embedding').main(
        ['--data', base, '--net', net, '--epochs', '1', '--batch', '2', '--embedding', '6', '--prior-poses', '300', '--refine', ckpt,
         '--out', str(tmp_path / 'eval'), '--cache', str(tmp_path / 'cache_posereg')])
    assert len(costs2) == 2 and np.all(np.isfinite(costs2))
    assert all(np.isfinite(v).all() and v[0] > 0 for v in results2.values())


def _write_icvl(base, name, n, seed):
    cam = A.Camera.icvl()
    frames, _, gtuvd = _frames_and_joints(cam, n, 16, 240, 320, (250., 250., 250.), 0, seed)
    lines = []
    for i in range(n):
        os.makedirs(os.path.join(base, 'Depth', '201403121135'), exist_ok=True)
        rel = '201403121135/%s_%04d.png' % (name, i)
        Image.fromarray(frames[i].astype(np.uint16)).save(os.path.join(base, 'Depth', rel))
        lines.append(rel + ' ' + ' '.join('%.4f' % v for v in gtuvd[i].reshape(-1)) + ' \n')
    with open(os.path.join(base, name + '.txt'), 'w') as f:
        f.writelines(lines)


@pytest.mark.parametrize('backend', BACKENDS)
def test_icvl_driver_end_to_end(backend, tmp_path):
    """examples/main_icvl_posereg_embedding.py (BASELINE configs[2]) on a tiny ICVL-format dataset (labels file + 16-bit PNGs), with all
    four augmentation modes."""
    R.set_default_runtime(get_runtime(backend))
    base = str(tmp_path / 'ICVL')
    _write_icvl(base, 'train', 4, 21)
    _write_icvl(base, 'test_seq_1', 2, 22)
    net = 'resnet' if backend == 'hip' else 'poseregnet'
    costs, results = _load_driver('main_icvl_posereg_embedding').main(
        ['--data', base, '--net', net, '--epochs', '1', '--batch', '2', '--embedding', '6', '--prior-poses', '300', '--aug-modes', 'com,rot,sc,none',
         '--out', str(tmp_path / 'eval'), '--cache', str(tmp_path / 'cache')])
    assert len(costs) == 2 and np.all(np.isfinite(costs))
    assert np.isfinite(results['test_seq_1']).all() and results['test_seq_1'][0] > 0
    assert os.path.isfile(str(tmp_path / 'eval' / 'network_prior.pkl'))


@pytest.mark.parametrize('backend', BACKENDS)
def test_synthetic_driver_end_to_end(backend, tmp_path):
    """examples/main_synthetic_posereg_embedding.py -- the reference's training script with rendered frames as the data source -- runs
    through: device crops, pose sampling + PCA kept on the device, trainer, prior layer, evaluation."""
    R.set_default_runtime(get_runtime(backend))
    net = 'resnet' if backend == 'hip' else 'poseregnet'
    costs, (mean_err, max_err, mean_pose_err) = _load_driver('main_synthetic_posereg_embedding').main(
        ['--net', net, '--frames', '8', '--epochs', '1', '--batch', '4', '--embedding', '6', '--prior-poses', '200', '--out', str(tmp_path / 'eval')])
    assert len(costs) == 2 and np.all(np.isfinite(costs))
    assert np.isfinite([mean_err, max_err, mean_pose_err]).all() and mean_err > 0
    assert os.path.isfile(str(tmp_path / 'eval' / 'network_prior.pkl'))


@pytest.mark.gpu
def test_resnet_learns_the_synthetic_pose_manifold(tmp_path):
    """Not a parity test but the thing parity is for: six epochs of the bs128 ResNet through PoseRegNetTrainer (online augmentation,
    PCA prior, ADAM, BatchNorm in training mode) on rendered depth frames of a 16-joint hand bring the mean joint error well below
    what predicting the mean pose gives, and the training cost falls by more than half."""
    R.set_default_runtime(get_runtime('hip'))
    costs, (mean_err, max_err, mean_pose_err) = _load_driver('main_synthetic_posereg_embedding').main(
        ['--net', 'resnet', '--frames', '4096', '--epochs', '6', '--out', str(tmp_path / 'eval')])
    assert np.all(np.isfinite(costs)) and len(costs) == 6 * 32
    assert np.mean(costs[-8:]) < 0.5 * np.mean(costs[:8]), (np.mean(costs[:8]), np.mean(costs[-8:]))
    assert mean_err < 0.75 * mean_pose_err, (mean_err, mean_pose_err)


@pytest.mark.gpu
def test_resnet_learns_with_bf16_operands_too(tmp_path, monkeypatch):
    """The same run with bf16 MFMA operands in the 3x3 convolutions and FC1 (BASELINE config 5's arithmetic) learns as well."""
    from hipdp import engine
    monkeypatch.setattr(heuristics, 'BF16_DEFAULT', True)
    R.set_default_runtime(get_runtime('hip'))
    costs, (mean_err, max_err, mean_pose_err) = _load_driver('main_synthetic_posereg_embedding').main(
        ['--net', 'resnet', '--frames', '4096', '--epochs', '6', '--out', str(tmp_path / 'eval')])
    assert np.all(np.isfinite(costs)) and np.mean(costs[-8:]) < 0.5 * np.mean(costs[:8])
    assert mean_err < 0.75 * mean_pose_err, (mean_err, mean_pose_err)
