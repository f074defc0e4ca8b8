"""Dataset readers: ICVL / NYU / MSRA15 `loadSequence` on tiny datasets written in the original file
formats (16-bit PNG + label lines; RGB-packed PNG + joint_data.mat; binary depth patches + joint.txt), cropped by the device
kernels and compared with the oracle's restatement of what the reference does per frame (importers.py:232-420, 596-700,
943-1064)."""
import os
import struct

import numpy as np
import pytest
import scipy.io
from PIL import Image

from data.dataset import Dataset
from data.importers import ICVLImporter, MSRA15Importer, NYUImporter
from hipdp import runtime as R
from oracle import augment as A
from tests.backends import BACKENDS, get_runtime


def _frames_and_joints(cam, n, J, H, W, cube, crop_idx, seed):
    rng = np.random.RandomState(seed)
    frames, coms = A.synthetic_frames(rng, n, cam, H, W, cube)
    frames = np.round(frames)                                    # the file formats hold integer millimetres (ICVL / NYU)
    com3d = np.stack([cam.jointImgTo3D(c) for c in coms])
    gt3D = com3d[:, None, :] + rng.normal(0, 30., (n, J, 3)).astype(np.float32)
    gt3D[:, crop_idx] = com3d
    gtuvd = np.stack([np.stack([cam.joint3DToImg(j) for j in g]) for g in gt3D]).astype(np.float32)
    return frames.astype(np.float32), gt3D.astype(np.float32), gtuvd


def _check(seq, imp, cam, frames, gtuvd, gt3D, cube, names=None):
    assert len(seq.data) == len(frames) and tuple(seq.config['cube']) == tuple(cube)
    fx, fy = abs(cam.fx), abs(cam.fy)
    for i, fr in enumerate(seq.data):
        d, _, _ = A.detector_preprocess(frames[i])
        ref, M, _ = A.crop_area_3d(d, fr.gtorig[imp.crop_joint_idx], cube, fx, fy)      # the centre as parsed from the label file
        assert np.array_equal(fr.dpt, ref) and fr.dpt.dtype == np.float32
        np.testing.assert_allclose(fr.T, M, rtol=1e-6, atol=1e-6)
        com3D = imp.jointImgTo3D(fr.gtorig[imp.crop_joint_idx])
        np.testing.assert_allclose(fr.com, com3D, rtol=0, atol=1e-4)
        np.testing.assert_allclose(fr.gtorig, gtuvd[i], rtol=0, atol=2e-3)
        np.testing.assert_allclose(fr.gt3Dorig, gt3D[i], rtol=0, atol=2e-2)
        np.testing.assert_allclose(fr.gt3Dcrop, fr.gt3Dorig - fr.com, rtol=0, atol=1e-5)
        uv1 = np.concatenate([fr.gtorig[:, :2], np.ones((fr.gtorig.shape[0], 1), np.float32)], axis=1)
        np.testing.assert_allclose(fr.gtcrop[:, :2], (uv1 @ np.asarray(fr.T).T)[:, :2], rtol=0, atol=1e-3)
    stack, labels = Dataset([seq]).imgStackDepthOnly(seq.name)
    assert stack.shape == (len(frames), 1, 128, 128) and stack.min() >= -1.0 and stack.max() <= 1.0 + 1e-6
    np.testing.assert_allclose(labels, np.stack([f.gt3Dcrop for f in seq.data]) / (cube[2] / 2.), rtol=0, atol=1e-6)


@pytest.mark.parametrize('backend', BACKENDS)
def test_icvl_reader(backend, tmp_path):
    R.set_default_runtime(get_runtime(backend))
    cam, cube = A.Camera.icvl(), (250, 250, 250)
    frames, gt3D, gtuvd = _frames_and_joints(cam, 5, 16, 240, 320, (250., 250., 250.), 0, 1)
    base = str(tmp_path / 'ICVL')
    folders = ['201403121135', '201403121135', '45', '45', '22-5']
    lines = []
    for i in range(5):
        os.makedirs(os.path.join(base, 'Depth', folders[i]), exist_ok=True)
        rel = '{}/image_{:04d}.png'.format(folders[i], i)
        Image.fromarray(frames[i].astype(np.uint16)).save(os.path.join(base, 'Depth', rel))
        lines.append(rel + ' ' + ' '.join('%.4f' % v for v in gtuvd[i].reshape(-1)) + ' \n')
    lines.insert(2, 'missing/none.png ' + ' '.join(['1.0'] * 48) + ' \n')          # a listed file that does not exist is skipped
    with open(os.path.join(base, 'train.txt'), 'w') as f:
        f.writelines(lines)
    imp = ICVLImporter(base, useCache=True, cacheDir=str(tmp_path / 'cache'))
    seq = imp.loadSequence('train')
    _check(seq, imp, cam, frames, gtuvd, gt3D, cube)
    assert [os.path.basename(f.fileName) for f in seq.data] == ['image_%04d.png' % i for i in range(5)]
    # second call is served from the pickle cache (same content), Nmax and shuffle apply afterwards
    seq2 = imp.loadSequence('train', Nmax=3, shuffle=True, rng=np.random.RandomState(0))
    assert len(seq2.data) == 3 and os.path.isfile(str(tmp_path / 'cache' / 'ICVLImporter_train_None_gt_250_cache.pkl'))
    # sub-sequences: '0' = the long-named (un-rotated) folders
    imp2 = ICVLImporter(base, useCache=False)
    assert [f.subSeqName for f in imp2.loadSequence('train', subSeq=['0']).data] == ['0', '0']
    assert [f.subSeqName for f in imp2.loadSequence('train', subSeq=['45', '22-5']).data] == ['45', '45', '22-5']
    with pytest.raises(TypeError):
        imp2.loadSequence('train', subSeq='45')
    # frames are cropped while the file list is being read (chunks of 2 here): only crops are kept, never the whole sequence of
    # raw frames -- and the result does not depend on the chunking, nor does Nmax
    import data.importers as I
    real = I.DepthImporter._crop_stream
    calls = []

    def small_chunks(self, config, docom, side, chunk=256):
        st = real(self, config, docom, side, chunk=2)
        inner = st._flush

        def spy():
            calls.append(len(st.pending))
            inner()
        st._flush = spy
        return st
    I.DepthImporter._crop_stream = small_chunks
    try:
        seq3 = imp2.loadSequence('train')
        seq4 = imp2.loadSequence('train', Nmax=3)
    finally:
        I.DepthImporter._crop_stream = real
    assert max(calls) <= 2 and len(calls) >= 3
    assert len(seq3.data) == 5 and all(np.array_equal(a.dpt, b.dpt) and np.array_equal(a.T, b.T) for a, b in zip(seq3.data, seq.data))
    assert len(seq4.data) == 3


@pytest.mark.parametrize('backend', BACKENDS)
def test_icvl_reader_with_refinement_net(backend, tmp_path):
    """loadSequence(docom=True) with `di.refineNet` set (the 'comref' data of main_nyu_posereg_embedding.py, importers.py:382-396):
    every frame goes crop -> CoM -> ScaleNet -> crop.  The chunked device cascade equals the per-frame HandDetector.cropArea3D."""
    from net.scalenet import ScaleNet, ScaleNetParams
    from util.handdetector import HandDetector
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    cam, cube = A.Camera.icvl(), (250, 250, 250)
    frames, gt3D, gtuvd = _frames_and_joints(cam, 3, 16, 240, 320, (250., 250., 250.), 0, 2)
    base = str(tmp_path / 'ICVL')
    os.makedirs(os.path.join(base, 'Depth', 'a'), exist_ok=True)
    lines = []
    for i in range(3):
        rel = 'a/image_{:04d}.png'.format(i)
        Image.fromarray(frames[i].astype(np.uint16)).save(os.path.join(base, 'Depth', rel))
        lines.append(rel + ' ' + ' '.join('%.4f' % v for v in gtuvd[i].reshape(-1)) + ' \n')
    with open(os.path.join(base, 'train.txt'), 'w') as f:
        f.writelines(lines)
    net = ScaleNet(np.random.RandomState(23455), cfgParams=ScaleNetParams(type=1, batchSize=2, numJoints=1, nDims=3))
    for i, l in enumerate(net.layers):
        if hasattr(l, 'b'):
            l.b.set_value(np.random.RandomState(i).normal(0, 0.05, l.b.get_value().shape).astype(np.float32))
    net.setDeterministic()
    imp = ICVLImporter(base, useCache=False)
    imp.refineNet = net
    seq = imp.loadSequence('train', docom=True)
    assert len(seq.data) == 3
    for i, fr in enumerate(seq.data):
        hd = HandDetector(frames[i].copy(), abs(imp.fx), abs(imp.fy), importer=imp, refineNet=net)
        c, M, com = hd.cropArea3D(com=fr.gtorig[imp.crop_joint_idx], size=cube, docom=True)
        np.testing.assert_allclose(fr.com, imp.jointImgTo3D(com), rtol=0, atol=1e-3)
        assert (fr.dpt != c).mean() < 1e-3 and fr.dpt.shape == (128, 128)
        np.testing.assert_allclose(fr.gt3Dcrop, fr.gt3Dorig - fr.com, rtol=0, atol=1e-5)
        # the refinement moved the centre away from plain docom
        _, _, com_plain = HandDetector(frames[i].copy(), abs(imp.fx), abs(imp.fy), importer=imp).cropArea3D(
            com=fr.gtorig[imp.crop_joint_idx], size=cube, docom=True)
        assert np.abs(np.asarray(com) - np.asarray(com_plain)).max() > 1e-3


@pytest.mark.parametrize('backend', BACKENDS)
def test_nyu_reader(backend, tmp_path):
    R.set_default_runtime(get_runtime(backend))
    cam, cube = A.Camera.nyu(), (300, 300, 300)
    n = 3
    frames, gt3D14, gtuvd14 = _frames_and_joints(cam, n, 14, 480, 640, (300., 300., 300.), 13, 2)
    imp = NYUImporter(str(tmp_path / 'NYU'), useCache=False)
    xyz, uvd = np.zeros((1, n, 36, 3), np.float32), np.zeros((1, n, 36, 3), np.float32)
    xyz[0][:, imp.restrictedJointsEval], uvd[0][:, imp.restrictedJointsEval] = gt3D14, gtuvd14
    d = os.path.join(str(tmp_path / 'NYU'), 'train')
    os.makedirs(d)
    scipy.io.savemat(os.path.join(d, 'joint_data.mat'), {'joint_xyz': xyz, 'joint_uvd': uvd})
    for i in range(n):
        v = frames[i].astype(np.int32)
        rgb = np.stack([np.zeros_like(v), v >> 8, v & 255], axis=2).astype(np.uint8)
        Image.fromarray(rgb).save(os.path.join(d, 'depth_1_%07d.png' % (i + 1)))
    seq = imp.loadSequence('train', docom=False)
    assert imp.numJoints == 14
    _check(seq, imp, cam, frames, gtuvd14, gt3D14, cube)


@pytest.mark.parametrize('backend', BACKENDS)
def test_msra_reader(backend, tmp_path):
    R.set_default_runtime(get_runtime(backend))
    cam, cube = A.Camera.msra(), (200, 200, 200)
    n = 4
    frames, gt3D, gtuvd = _frames_and_joints(cam, n, 21, 240, 320, (200., 200., 200.), 5, 3)
    base = str(tmp_path / 'MSRA')
    for g, idx in (('1', [0, 1]), ('IP', [2, 3])):
        d = os.path.join(base, 'P0', g)
        os.makedirs(d)
        with open(os.path.join(d, 'joint.txt'), 'w') as f:
            f.write('%d\n' % len(idx))
            for k, i in enumerate(idx):
                j = gt3D[i].copy()
                j[:, 2] *= -1.                                   # the files hold -z
                f.write(' '.join('%.4f' % v for v in j.reshape(-1)) + '\n')
                ys, xs = np.nonzero(frames[i])
                top, bottom, left, right = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
                with open(os.path.join(d, '%06d_depth.bin' % k), 'wb') as fb:
                    fb.write(struct.pack('6i', 320, 240, left, top, right, bottom))
                    frames[i][top:bottom, left:right].astype(np.float32).tofile(fb)
    imp = MSRA15Importer(base, useCache=False)
    seq = imp.loadSequence('P0')
    _check(seq, imp, cam, frames, gtuvd, gt3D, cube)
    assert [f.subSeqName for f in imp.loadSequence('P0', subSeq=['IP']).data] == ['IP', 'IP']


def test_load_baseline_readers_against_reference_outputs(tmp_path):
    """tests/golden/baseline.npz (make_golden_r6.py): ICVLImporter / NYUImporter.loadBaseline and loadBaseline2D
    (/root/reference/src/data/importers.py:422-484, 1079-1175) run by the reference on small result files; the files are rebuilt here from
    the numbers the fixture keeps (text rows, the .mat arrays, the depth frames)."""
    import scipy.io
    from PIL import Image
    from data.importers import ICVLImporter, NYUImporter
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'baseline.npz'))
    icvl = ICVLImporter('x')
    for first in (0, 1):
        fn = str(tmp_path / ('lrf_%d.txt' % first))
        with open(fn, 'w') as fh:
            for k, r in enumerate(g['icvl_rows']):
                fh.write((('image_%04d.png ' % k) if first else '') + ' '.join('%.4f' % v for v in r.reshape(-1)) + '\n')
                if k == 2:
                    fh.write('\n')
        got = icvl.loadBaseline(fn, firstName=bool(first))
        assert isinstance(got, list) and got[0].dtype == np.float32
        np.testing.assert_allclose(np.stack(got), g['icvl_3d_%d' % first], rtol=2e-7, atol=0)
        np.testing.assert_array_equal(np.stack(icvl.loadBaseline2D(fn, firstName=bool(first))), g['icvl_2d_%d' % first])
    nyu = NYUImporter('x')
    fn = str(tmp_path / 'pred.txt')
    with open(fn, 'w') as fh:
        for r in g['nyu_rows']:
            fh.write(' '.join('%.4f' % v for v in r.reshape(-1)) + '\n')
    np.testing.assert_allclose(np.stack(nyu.loadBaseline(fn)), g['nyu_text_3d'], rtol=2e-7, atol=0)
    assert nyu.numJoints == 14
    uvc, depth, gt = g['nyu_uvc'], g['nyu_depth'], g['nyu_gt']
    names = np.empty((1, 14), dtype=object)
    for j in range(14):
        names[0, j] = 'J%d' % j
    scipy.io.savemat(str(tmp_path / 'test_predictions.mat'), {'conv_joint_names': names, 'pred_joint_uvconf': uvc[None]})
    for f in range(depth.shape[0]):
        if f == 2:
            continue                                                # a missing frame is skipped
        rgb = np.zeros(depth.shape[1:] + (3,), np.uint8)
        rgb[..., 1], rgb[..., 2] = depth[f] >> 8, depth[f] & 255
        Image.fromarray(rgb).save(str(tmp_path / ('depth_1_%07d.png' % (f + 1))))
    got = nyu.loadBaseline(str(tmp_path / 'test_predictions.mat'), gt)
    assert len(got) == 3
    np.testing.assert_allclose(np.stack(got), g['nyu_mat_3d'], rtol=2e-7, atol=0)
    np.testing.assert_array_equal(np.stack(nyu.loadBaseline2D(str(tmp_path / 'test_predictions.mat'))), g['nyu_mat_2d'])
