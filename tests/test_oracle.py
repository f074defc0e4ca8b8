"""Pins for the oracle: the reference-generated golden fixtures (tests/golden/make_golden.py) and the
independent torch-autograd cross-check.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import augment as A
from oracle import layers as L
from oracle import nets, torch_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def shapes():
    return json.load(open(os.path.join(GOLD, 'shapes.json')))


@pytest.mark.parametrize('name', ['resnet_t0_128', 'resnet_t1_128', 'resnet_t1_256', 'resnet_t0_64_b4'])
def test_resnet_shapes_match_reference_layerparams(shapes, name):
    g = shapes[name]
    net = nets.build_resnet(**g['args'])
    assert len(net['layers']) == len(g['layers'])
    for l, r in zip(net['layers'], g['layers']):
        assert list(l['in_dim']) == r['inputDim']
        assert list(l['out_dim']) == r['outputDim'], (l, r)
        if l['kind'] in ('conv', 'convpool'):
            assert [l['nf'], l['in_dim'][1], l['k'][0], l['k'][1]] == r['filter_shape']


@pytest.mark.parametrize('name', ['poseregnet_t0', 'poseregnet_t11', 'poseregnet_t0_b16'])
def test_poseregnet_layers_match_reference_params(shapes, name):
    g = shapes[name]
    net = nets.build_poseregnet(**g['args'])
    kinds = {'ConvPoolLayerParams': 'convpool', 'HiddenLayerParams': 'fc', 'DropoutLayerParams': 'dropout'}
    assert [l['kind'] for l in net['layers']] == [kinds[r['cls']] for r in g['layers']]
    for l, r in zip(net['layers'], g['layers']):
        assert list(l['out_dim']) == r['outputDim']
        if l['kind'] == 'convpool':
            assert list(l['pool']) == r['poolsize'] and r['border_mode'] == l['border']
            assert r['activation'] == 'ReLU'
        if l['kind'] == 'fc':
            assert (r['activation'] == 'ReLU') == (l['act'] == 'relu')
    assert list(net['out_dim']) == g['outputDim']


@pytest.mark.parametrize('name', ['scalenet_t1', 'scalenet_t1_96_b8'])
def test_scalenet_layers_match_reference_params(shapes, name):
    """The reference's real ScaleNetParams (net/scalenet.py:33-127) against the oracle's and the product's layer lists."""
    from net.scalenet import ScaleNetParams
    g = shapes[name]
    net = nets.build_scalenet(**g['args'])
    kinds = {'ConvPoolLayerParams': 'convpool', 'HiddenLayerParams': 'fc', 'DropoutLayerParams': 'dropout'}
    assert [l['kind'] for l in net['layers']] == [kinds[r['cls']] for r in g['layers']]
    cfg = ScaleNetParams(**g['args'])
    assert [list(d) for d in cfg.inputDim] == g['inputDim'] == [list(d) for d in net['in_dim']]
    for l, r, p in zip(net['layers'], g['layers'], cfg.layers):
        assert list(l['out_dim']) == r['outputDim'] == list(p.outputDim) and list(l['in_dim']) == r['inputDim'] == list(p.inputDim)
        if l['kind'] == 'convpool':
            assert list(l['pool']) == r['poolsize'] == list(p.poolsize) and r['border_mode'] == l['border']
            assert r['filter_shape'] == list(p.filter_shape)
    assert list(net['out_dim']) == g['outputDim'] == list(cfg.outputDim)


def test_param_count_matches_survey():
    net = nets.build_resnet(type=1, numJoints=14, nDims=3)
    P = nets.init_params(net, np.random.RandomState(0))
    n = sum(int(np.prod(a.shape)) for v in P.values() for a in v[:2])
    assert n == 18714452          # 18.714 M, SURVEY.md section 8(d)
    assert len(net['layers']) == 190


def test_geometry_against_reference():
    g = np.load(os.path.join(GOLD, 'geometry.npz'))
    for i in range(32):
        out = A.rotate_point_2d(g['rot_pts'][i].astype('float64'), g['rot_ctr'][i], g['rot_ang'][i])
        np.testing.assert_allclose(out, g['rot_out64'][i], rtol=0, atol=1e-9)
        out32 = A.rotate_point_2d(g['rot_pts'][i], g['rot_ctr'][i], g['rot_ang'][i])
        np.testing.assert_allclose(out32, g['rot_out'][i], rtol=2e-6, atol=2e-5)
    cams = dict(icvl=A.Camera.icvl(), msra=A.Camera.msra(), nyu=A.Camera.nyu())
    for nm, cam in cams.items():
        s64 = g['proj_in_float64']
        x3 = np.stack([cam.jointImgTo3D(p) for p in s64])
        assert np.array_equal(x3, g['to3d_%s_float64' % nm])          # float64 inputs: bit-exact
        back = np.stack([cam.joint3DToImg(p) for p in x3.astype('float64')])
        assert np.array_equal(back, g['toimg_%s_float64' % nm])
        # float32 inputs: the fixture ran under NumPy-2 scalar rules (all-float32), the oracle keeps the
        # NumPy-1.x float64 intermediates of the reference's own environment -> 1 ulp tolerance
        s32 = g['proj_in_float32']
        x3 = np.stack([cam.jointImgTo3D(p) for p in s32])
        np.testing.assert_allclose(x3, g['to3d_%s_float32' % nm], rtol=3e-7, atol=1e-5)
    for nm, (fx, fy) in (('icvl', (241.42, 241.42)), ('nyu', (588.03, 587.07))):
        for i in range(64):
            b = A.com_to_bounds(g['ctb_com'][i], (g['ctb_cube'][i],) * 3, fx, fy)
            np.testing.assert_allclose(np.array(b, 'float64'), g['ctb_' + nm][i], rtol=0, atol=0)


def test_chunks_partition():
    for c in json.load(open(os.path.join(GOLD, 'chunks.json'))):
        n, k = c['n'], c['k']
        mine = [[i, min(i + k, n)] for i in range(0, n, k)]
        assert mine == c['chunks']
        from util.helpers import chunks
        got = list(chunks(list(range(n)), k))
        assert [[g[0], g[-1] + 1] for g in got] == c['chunks'] and sum(got, []) == list(range(n))
    from util.helpers import chunks
    assert list(chunks([], 3)) == [] and list(chunks('abcde', 2)) == ['ab', 'cd', 'e']


def test_shuffle_many_inplace_applies_the_references_permutation():
    """helpers.json: the reference's own shuffle_many_inplace (/root/reference/src/util/helpers.py:87-108) on arange(n)."""
    from util.helpers import shuffle_many_inplace
    for c in json.load(open(os.path.join(GOLD, 'helpers.json'))):
        rng = np.random.RandomState(c['seed'])
        a = np.arange(c['n'])
        b = np.arange(c['n'] * 6, dtype=np.float32).reshape(c['n'], 2, 3)
        shuffle_many_inplace([a, b], random_state=rng)
        assert a.tolist() == c['perm'] and np.array_equal(b[:, 0, 0], 6. * a)
        assert int(rng.randint(1 << 30)) == c['next_draw']          # the stream was consumed draw for draw
    with pytest.raises(ValueError):
        shuffle_many_inplace([np.arange(3)], random_state=5)
    with pytest.raises(AssertionError):
        shuffle_many_inplace([np.arange(3), np.arange(4)])
    np.random.seed(3)
    a = np.arange(9)
    shuffle_many_inplace([a])                                        # global stream when no state is given
    assert sorted(a.tolist()) == list(range(9))


def test_dataset_stacks_against_the_reference():
    """dataset.npz: the reference's Dataset.imgStackDepthOnly (/root/reference/src/data/dataset.py:72-111), both normalisations."""
    import sys
    sys.path.insert(0, GOLD)
    try:
        from make_golden_r4 import synthetic_sequences
    finally:
        sys.path.remove(GOLD)
    from data.dataset import Dataset, ICVLDataset, MSRA15Dataset, NYUDataset
    from data import importers
    g = np.load(os.path.join(GOLD, 'dataset.npz'))
    seqs = synthetic_sequences()
    for s in seqs:
        for nz in (False, True):
            img, lab = Dataset(seqs, localCache=False).imgStackDepthOnly(s.name, normZeroOne=nz)
            assert img.dtype == np.float32 and lab.dtype == np.float32
            assert np.array_equal(img, g['%s_img_%d' % (s.name, nz)]) and np.array_equal(lab, g['%s_lab_%d' % (s.name, nz)])
    ds = Dataset(seqs)
    assert ds.imgSeq('test_1') is seqs[1] and ds.imgSeq('nope') == [] and ds.imgStackDepthOnly('nope') == []
    first = ds.imgStackDepthOnly('train')
    assert ds.imgStackDepthOnly('train')[0] is first[0]              # localCache
    ds.imgSeqs = seqs[:1]
    assert ds.imgStackDepthOnly('train')[0] is not first[0] and ds.imgSeqs == seqs[:1]
    for cls, imp, default in ((ICVLDataset, importers.ICVLImporter, '../../data/ICVL/'), (MSRA15Dataset, importers.MSRA15Importer, '../../data/MSRA15/'),
                              (NYUDataset, importers.NYUImporter, '../../data/NYU/')):
        assert isinstance(cls().lmi, imp) and cls().lmi.basepath == default
        d = cls(seqs, basepath='/data/x/', localCache=False)
        assert d.lmi.basepath == '/data/x/' and d.imgSeqs is seqs and cls.__name__ == type(d).__name__


def test_conv_is_true_convolution():
    rng = np.random.RandomState(1)
    x = rng.normal(size=(2, 3, 9, 8))
    W = rng.normal(size=(4, 3, 3, 3))
    y = L.conv2d_fwd(x, W, None, (1, 1), 'valid')
    import scipy.signal
    ref = np.zeros_like(y)
    for n in range(2):
        for f in range(4):
            for c in range(3):
                ref[n, f] += scipy.signal.convolve2d(x[n, c], W[f, c], mode='valid')
    np.testing.assert_allclose(y, ref, atol=1e-12)
    # 'half' + stride 2 == every second output of the stride-1 'half' result
    y1 = L.conv2d_fwd(x, W, None, (1, 1), 'half')
    y2 = L.conv2d_fwd(x, W, None, (2, 2), 'half')
    np.testing.assert_allclose(y2, y1[:, :, ::2, ::2], atol=1e-12)
    assert y2.shape[2:] == (5, 4)


def test_numpy_backward_matches_torch_autograd():
    rng = np.random.RandomState(23455)
    net = nets.build_resnet(type=1, wIn=32, hIn=32, batchSize=4, numJoints=14, nDims=3)
    P = nets.perturb_bn(nets.init_params(net, rng, np.float64), net, rng)
    x = nets.synthetic_crops(rng, 4, 32, 32, np.float64)
    y = rng.normal(0, .3, (4, 42))
    cost, G, _, out = nets.cost_and_grads(net, P, x, y)
    c2, G2, out2 = torch_ref.cost_and_grads(net, P, x, y)
    assert abs(cost - c2) < 1e-12 * abs(c2)
    np.testing.assert_allclose(out, out2, atol=1e-12)
    gmax = max(np.abs(G2[i][s]).max() for i in G2 for s in range(2))
    for i in G:
        for s in range(2):
            np.testing.assert_allclose(G[i][s], G2[i][s], rtol=1e-8, atol=1e-12 * gmax)


def test_poseregnet_backward_matches_torch_autograd():
    rng = np.random.RandomState(5)
    net = nets.build_poseregnet(type=11, batchSize=3, numJoints=14, nDims=3, wIn=64, hIn=64)
    # dropout in eval form on both sides (0.7 * x is linear): exercises conv-pool / fc backward
    for l in net['layers']:
        if l['kind'] == 'dropout':
            l['kind'] = 'relu'       # keep graph shape; relu of a relu output is the identity
    P = nets.init_params(net, rng, np.float64)
    P = nets.perturb_bn(P, net, rng)
    x = nets.synthetic_crops(rng, 3, 64, 64, np.float64)
    y = rng.normal(0, .3, (3, 42))
    cost, G, _, out = nets.cost_and_grads(net, P, x, y)
    c2, G2, out2 = torch_ref.cost_and_grads(net, P, x, y)
    assert abs(cost - c2) < 1e-12 * abs(c2)
    for i in G:
        for s in range(2):
            np.testing.assert_allclose(G[i][s], G2[i][s], rtol=1e-8, atol=1e-13)


def test_scalenet_backward_matches_torch_autograd():
    """ScaleNet (three towers, concatenation; scalenet.py:49-180): analytic backward of the multi-input graph vs autograd."""
    rng = np.random.RandomState(6)
    net = nets.build_scalenet(batchSize=3, numJoints=1, nDims=3, wIn=96, hIn=96)
    for l in net['layers']:
        if l['kind'] == 'dropout':
            l['kind'] = 'relu'
    P = nets.init_params(net, rng, np.float64)
    xs = nets.scalenet_inputs(nets.synthetic_crops(rng, 3, 96, 96, np.float64))
    assert [a.shape[2] for a in xs] == [96, 48, 24]
    y = rng.normal(0, .3, (3, 3))
    cost, G, _, out = nets.cost_and_grads(net, P, xs, y)
    c2, G2, out2 = torch_ref.cost_and_grads(net, P, xs, y)
    assert abs(cost - c2) < 1e-12 * abs(c2)
    for i in G:
        for s in range(2):
            np.testing.assert_allclose(G[i][s], G2[i][s], rtol=1e-8, atol=1e-13)


def test_adam_constants_are_float32_rounded():
    p = [np.array([1.0, -2.0], np.float64)]
    g = [np.array([0.5, 0.25], np.float64)]
    m = [np.zeros(2)]
    v = [np.zeros(2)]
    t = L.adam_step(p, g, m, v, 1.0, 1e-3)
    b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))
    np.testing.assert_allclose(m[0], (1 - b1) * g[0], rtol=1e-15)
    np.testing.assert_allclose(v[0], (1 - b2) * g[0] ** 2, rtol=1e-15)
    assert t == 2.0
    # first step of ADAM moves every weight by ~lr
    np.testing.assert_allclose(p[0], [1.0 - 1e-3, -2.0 - 1e-3], atol=1e-9)
    assert L.lr_of_ep(1e-3, 1) == np.float32(1e-4) and L.lr_of_ep(1e-3, 2) == np.float32(1e-3 / 3.)
    assert L.lr_of_ep(1e-3, 3) == np.float32(1e-3 * np.exp(-0.12))


def test_compute_output_padding():
    rng = np.random.RandomState(2)
    net = nets.build_poseregnet(type=0, batchSize=4, numJoints=1, nDims=30, wIn=64, hIn=64)
    P = nets.init_params(net, rng, np.float64)
    x = nets.synthetic_crops(rng, 6, 64, 64, np.float64)
    out = nets.compute_output(net, P, x)
    assert out.shape == (6, 30)
    o2, _ = nets.forward(net, P, np.concatenate([x[4:], x[5:], x[5:]], 0), False)
    np.testing.assert_allclose(out[4:], o2[:2], atol=1e-12)


def test_augment_modes_run_and_are_consistent():
    rng = np.random.RandomState(7)
    cam = A.Camera.icvl()
    fx, fy = cam.fx, cam.fy
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, 4, cam)
    for i, mode in enumerate(['com', 'rot', 'sc', 'none']):
        com2d = cam.joint3DToImg(coms[i])
        off = rng.randn(3) * 5.
        imgD, lab, cube, com, M, rot = A.augment_crop(imgs[i].copy(), gts[i].copy(), com2d, cubes[i], Ms[i], mode,
                                                      off, 33.3, 1.03, cam, fx, fy)
        assert imgD.shape == (128, 128) and imgD.dtype == np.float32
        assert imgD.min() >= -1.0 and imgD.max() <= 1.0
        assert lab.shape == (16, 3)
        if mode == 'none':
            np.testing.assert_allclose(imgD, imgs[i], atol=2e-6)
            np.testing.assert_allclose(lab, gts[i] / 125., rtol=1e-6)
        if mode == 'sc':
            np.testing.assert_allclose(cube, cubes[i] * 1.03, rtol=1e-6)
    # rotating by 0 / offset 0 / scale 1 is the identity (the reference's early returns)
    com2d = cam.joint3DToImg(coms[0])
    for mode in ('com', 'rot', 'sc'):
        imgD, lab, *_ = A.augment_crop(imgs[0].copy(), gts[0].copy(), com2d, cubes[0], Ms[0], mode,
                                       np.zeros(3), 0., 1., cam, fx, fy)
        np.testing.assert_allclose(imgD, imgs[0], atol=2e-6)


def test_warp_restatements_identity_and_shift():
    rng = np.random.RandomState(3)
    src = rng.uniform(1, 2, (128, 128)).astype('float32')
    np.testing.assert_array_equal(A.warp_perspective_nn(src, np.eye(3)), src)
    np.testing.assert_array_equal(A.warp_affine_nn(src, np.eye(3)[:2]), src)
    T = np.eye(3)
    T[0, 2], T[1, 2] = 3, -2           # forward map: dst(x, y) = src(x - 3, y + 2)
    w = A.warp_perspective_nn(src, T)
    np.testing.assert_array_equal(w[0:126, 3:], src[2:128, 0:125])
    assert (w[:, :3] == 0).all() and (w[126:, :] == 0).all()
    w2 = A.warp_affine_nn(src, T[:2])
    np.testing.assert_array_equal(w, w2)
    # 90-degree rotation about (64, 64) is an exact pixel permutation
    R = A.rotation_matrix_2d((64, 64), 90.)
    w3 = A.warp_affine_nn(src, R)
    # cv2 convention: positive angle = counter-clockwise; dst(x, y) = src(128 - y, x) inside the image
    yy, xx = np.mgrid[1:128, 0:128]
    np.testing.assert_array_equal(w3[yy, xx], src[xx, 128 - yy])


def _golden(name):
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name))


def test_crop_helpers_against_reference_outputs():
    """tests/golden/crop.npz was produced by the reference's own HandDetector (util/handdetector.py:53-130, 204-226,
    260-296, 540-558; NumPy / SciPy code that runs here): depth-range preprocessing, comToBounds, getCrop, calculateCoM,
    refineCoMIterative -- the oracle restatement and the product's host helpers reproduce it."""
    from oracle import augment as A
    from util.handdetector import HandDetector
    g = _golden('crop.npz')
    frames, coms = g['frames'], g['coms']
    for i in range(frames.shape[0]):
        d, mn, mx = A.detector_preprocess(frames[i])
        assert np.array_equal(d, g['pre'][i]) and (mn, mx) == tuple(g['range'][i])
        b = A.com_to_bounds(coms[i], (250., 250., 250.), 241.42, 241.42)
        # pixel bounds exactly; zstart / zend were evaluated in float32 by today's NumPy when the fixture was made (float32
        # scalar + python float), in float64 by the NumPy of 2017 and by the restatement
        assert tuple(b[:4]) == tuple(int(v) for v in g['bounds'][i][:4])
        np.testing.assert_allclose(b[4:], g['bounds'][i][4:], rtol=1e-6)
        assert np.array_equal(A.get_crop(d, *b), g['crop_%d' % i])
        np.testing.assert_allclose(A.calculate_com(d, mn, mx), g['com_full'][i], rtol=1e-6)
        np.testing.assert_allclose(A.calculate_com(g['crop_%d' % i], mn, mx), g['com_crop'][i], rtol=1e-6)
        hd = HandDetector(frames[i].copy(), 241.42, 241.42)
        assert np.array_equal(hd.dpt, g['pre'][i])
        np.testing.assert_allclose(hd.comToBounds(coms[i], (250., 250., 250.)), g['bounds'][i], rtol=1e-6)
        assert np.array_equal(hd.getCrop(hd.dpt, *b), g['crop_%d' % i])
        np.testing.assert_allclose(hd.calculateCoM(hd.dpt), g['com_full'][i], rtol=1e-6)
        np.testing.assert_allclose(hd.refineCoMIterative(coms[i].astype('float64'), 3, (250., 250., 250.)), g['com_it'][i], rtol=1e-5)


def test_sample_random_poses_against_reference_outputs():
    """tests/golden/poses.npz: the reference's HandDetector.sampleRandomPoses (util/handdetector.py:805-909) run here on seeded
    inputs: same draws, same modes, same arithmetic.  Agreement is to float32 round-off, not bit for bit: the reference's
    scalar projections (importers.py:80-119) mix float32 scalars with Python floats, which NumPy evaluated in float64 in 2017
    and evaluates in float32 today (the fixture), while the product evaluates them in float64 throughout."""
    from data.importers import ICVLImporter, NYUImporter
    from util.handdetector import HandDetector
    g = _golden('poses.npz')
    for nm, di in (('icvl', ICVLImporter('x')), ('nyu', NYUImporter('x'))):
        args = (g['%s_gt' % nm], g['%s_com' % nm], g['%s_cube' % nm], 300)
        for tag, modes in (('main', ['com', 'rot', 'none']), ('all', ['com', 'rot', 'sc', 'none', 'rot+com', 'rot+com+sc'])):
            got = HandDetector.sampleRandomPoses(di, np.random.RandomState(9), *args, modes)
            assert got.shape == g['%s_%s' % (nm, tag)].shape and got.dtype == np.float32
            np.testing.assert_allclose(got, g['%s_%s' % (nm, tag)], rtol=0, atol=2e-6)       # poses are O(1) normalised coordinates


def test_rotate_points_3d_and_rot3d_pose_sampling_against_reference_outputs():
    """tests/golden/rot3d.npz (make_golden_r6.py): the reference's rotatePoints3D / rotatePoint3D / getRotationMatrix / transformPoint3D /
    getTransformationMatrix (data/transformations.py:34-45, 105-166) and HandDetector.sampleRandomPoses(rot3D=True)
    (util/handdetector.py:870, 891, 903) run here, with a documented stand-in for the one transforms3d function they import."""
    from data import transformations as T
    from data.importers import ICVLImporter, NYUImporter
    from util.handdetector import HandDetector
    g = _golden('rot3d.npz')
    for i in range(g['pts'].shape[0]):
        np.testing.assert_allclose(T.getRotationMatrix(*g['ang'][i]), g['R'][i], rtol=0, atol=1e-15)
        got = T.rotatePoints3D(g['pts'][i], g['ctr'][i], *g['ang'][i])
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, g['out'][i], rtol=0, atol=1e-4)        # coordinates are O(600) mm: a float32 ulp is 6e-5
        np.testing.assert_allclose(T.rotatePoint3D(g['pts'][i, 0].astype(np.float64), g['ctr'][i].astype(np.float64), *g['ang'][i]),
                                   g['out_point64'][i], rtol=0, atol=1e-10)
    for i in range(g['tp3_M'].shape[0]):
        np.testing.assert_allclose(T.transformPoint3D(g['pts'][i, 0], g['tp3_M'][i]), g['tp3_out'][i], rtol=1e-12, atol=1e-9)
    for a, want in zip(g['gtm_args'], g['gtm_out']):
        np.testing.assert_allclose(T.getTransformationMatrix(a[0:2], a[2], a[3:5], a[5]), want, rtol=1e-14, atol=1e-12)
    a = np.deg2rad(g['ang'][:, 0]), np.deg2rad(g['ang'][:, 1]), np.deg2rad(g['ang'][:, 2])
    np.testing.assert_allclose(T.euler_rxyz_matrix(*a), g['R'][:, :3, :3], rtol=0, atol=1e-15)           # the array form
    for nm, di in (('icvl', ICVLImporter('x')), ('nyu', NYUImporter('x'))):
        args = (g['%s_gt' % nm], g['%s_com' % nm], g['%s_cube' % nm], 300)
        for tag, modes in (('main', ['com', 'rot', 'none']), ('all', ['com', 'rot', 'sc', 'none', 'rot+com', 'rot+com+sc'])):
            got, ncom, ncube, _ = HandDetector.sampleRandomPoses(di, np.random.RandomState(9), *args, modes, retall=True, rot3D=True)
            assert got.shape == g['%s_%s' % (nm, tag)].shape and got.dtype == np.float32
            np.testing.assert_allclose(got, g['%s_%s' % (nm, tag)], rtol=0, atol=2e-6)
            np.testing.assert_array_equal(ncom, g['%s_%s_com' % (nm, tag)])
            # ('sc': float32 cube x float64 draw -- a float32 product under the reference's 2017 NumPy and here, a rounded float64 one under
            #  the NumPy 2 that generated the fixture: one float32 ulp)
            np.testing.assert_allclose(ncube, g['%s_%s_cube' % (nm, tag)], rtol=1.3e-7, atol=0)


def test_pca_projection_is_sklearns():
    """proj.transform(label) of poseregnettrainer.py:262 is scikit-learn's PCA.transform: (x - mean_) . components_^T."""
    from oracle import augment as A
    from sklearn.decomposition import PCA
    rng = np.random.RandomState(3)
    X = rng.normal(size=(500, 42)) @ rng.normal(size=(42, 42))
    pca = PCA(n_components=30).fit(X)
    x = rng.normal(size=(7, 42))
    for i in range(x.shape[0]):
        np.testing.assert_allclose(A.pca_transform(x[i], pca.mean_, pca.components_), pca.transform(x[i:i + 1]), rtol=1e-10, atol=1e-10)


def test_trainer_helpers_against_reference_outputs():
    """tests/golden/trainer.json: the reference's NetTrainerParams.lr_of_ep and NetTrainer.alignData (trainer/nettrainer.py:47-72,
    365-413) run here: learning-rate schedule, and the padding of the last minibatch with samples drawn by
    RandomState(number of samples) (or by repeating the last sample)."""
    import types
    from trainer.nettrainer import NetTrainer, NetTrainerParams
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'trainer.json')))
    p = NetTrainerParams()
    p.learning_rate = 0.001
    for ep, lr in g['lr_of_ep']:
        assert float(p.lr_of_ep(ep)) == lr and float(L.lr_of_ep(0.001, ep)) == lr
    for c in g['alignData']:
        dummy = types.SimpleNamespace(cfgParams=types.SimpleNamespace(pad_random=c['pad_random'], batch_size=4))
        data = np.arange(c['n'], dtype='float32').reshape(c['n'], 1) + 1.
        padded = NetTrainer.alignData(dummy, data, alignSize=c['align'])
        assert [float(v) for v in padded[:, 0]] == c['padded'], (c['n'], c['align'])


def test_init_values_against_reference_outputs():
    """tests/golden/init.npz: the reference's Layer.getInitVals (net/layer.py:70-124) -- the product's initialiser draws the
    same numbers from the same generator, and the oracle's init_params follows the same rules."""
    from net.layer import Layer
    g = _golden('init.npz')
    for tag, shape, mode, act, method, orth in (('conv_he', (8, 1, 5, 5), 'conv', 'ReLU', 'He', False), ('conv_he_res', (64, 16, 3, 3), 'conv', None, 'He', False),
                                                ('fc_he', (968, 64), 'fc', 'ReLU', None, False), ('fc_linear', (30, 42), 'fc', None, 'tanh', False),
                                                ('conv_xavier', (16, 8, 3, 3), 'conv', None, 'Xavier', False), ('fc_sigmoid', (20, 10), 'fc', 'sigmoid', None, False),
                                                ('conv_orth', (8, 4, 3, 3), 'conv', 'ReLU', 'He', True)):
        lay = Layer(np.random.RandomState(23455))
        got = lay.getInitVals(shape, mode, act_fn=act, method=method, orthogonal=orth)
        assert got.dtype == np.float32 and got.shape == g[tag].shape
        if orth:
            np.testing.assert_allclose(np.abs(got), np.abs(g[tag]), rtol=1e-4, atol=1e-6)     # SVD vectors: sign / LAPACK build
        else:
            assert np.array_equal(got, g[tag]), tag
        assert np.array_equal(lay.rng.uniform(size=3), g[tag + '_next'])
    # the oracle's net initialiser: conv-pool 1 of PoseRegNet is the first draw of RandomState(23455)
    onet = nets.build_poseregnet(type=0, batchSize=2, numJoints=1, nDims=30)
    P = nets.init_params(onet, np.random.RandomState(23455), np.float32)
    assert np.array_equal(P[0][0], g['conv_he'])


def test_evaluation_metrics_against_reference_outputs():
    """tests/golden/evaluation.json: the reference's HandposeEvaluation (util/handpose_evaluation.py:92-228) on seeded joints with a
    missing (NaN) joint."""
    from util.handpose_evaluation import HandposeEvaluation
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'evaluation.json')))
    gt, jt = np.asarray(g['gt']), np.asarray(g['joints'])
    hpe = HandposeEvaluation(list(gt), list(jt))
    r = dict(rtol=1e-12, atol=0)
    np.testing.assert_allclose(hpe.getMeanError(), g['mean'], **r)
    np.testing.assert_allclose(hpe.getStdError(), g['std'], **r)
    np.testing.assert_allclose(hpe.getMedianError(), g['median'], **r)
    np.testing.assert_allclose(hpe.getMaxError(), g['max'], **r)
    np.testing.assert_allclose(hpe.getMeanErrorOverSeq(), g['mean_over_seq'], **r)
    np.testing.assert_allclose(hpe.getMaxErrorOverSeq(), g['max_over_seq'], **r)
    np.testing.assert_allclose([hpe.getJointMeanError(j) for j in range(14)], g['joint_mean'], **r)
    np.testing.assert_allclose([hpe.getJointMaxError(j) for j in range(14)], g['joint_max'], **r)
    assert [[d, int(hpe.getNumFramesWithinMaxDist(d))] for d in (10, 20, 30, 40)] == g['within']
    np.testing.assert_allclose(L.mean_joint_error(gt, jt), g['mean'], **r)


def test_com_to_transform_against_the_reference_with_py2_division():
    """tests/golden/transform.npz: the reference's own HandDetector.comToTransform (util/handdetector.py:228-258) executed here
    with its two int / int crop-size expressions (lines 246, 249) written as the floor divisions Python 2 performed
    (tests/golden/make_golden_r2.py) -- pins the oracle's restatement, the product's host helper and, through the oracle, the
    crop transform the augmentation kernels compute."""
    from util.handdetector import HandDetector
    g = _golden('transform.npz')
    frame = np.full((240, 320), 500., np.float32)
    n_branch = [0, 0]
    for com, size, (fx, fy), ds, M, bounds in zip(g['com'], g['size'], g['fx'], g['dsize'], g['M'], g['bounds']):
        ds = (int(ds[0]), int(ds[1]))
        size = tuple(float(v) for v in size)
        np.testing.assert_array_equal(np.asarray(A.com_to_bounds(com, size, fx, fy), np.float64), bounds)
        np.testing.assert_allclose(A.com_to_transform(com, size, fx, fy, ds), M, rtol=0, atol=1e-12)
        hd = HandDetector(frame.copy(), fx, fy)
        np.testing.assert_allclose(hd.comToTransform(com, size, ds), M, rtol=0, atol=1e-12)
        n_branch[int((bounds[1] - bounds[0]) > (bounds[3] - bounds[2]))] += 1
        # the floor division is visible: true division gives another paste offset whenever the crop is not square
        wb, hb = bounds[1] - bounds[0], bounds[3] - bounds[2]
        if wb != hb and (min(wb, hb) * ds[0]) % max(wb, hb) != 0:
            assert (min(wb, hb) * ds[0] / max(wb, hb)) != (min(wb, hb) * ds[0] // max(wb, hb))
    assert min(n_branch) >= 5, n_branch                          # both branches (wb > hb, wb <= hb) are covered


def test_netbase_load_reads_a_python2_cpickle_checkpoint(tmp_path):
    """tests/golden/net_py2.pkl: a checkpoint in the byte layout Python 2's cPickle (protocol 2) gave NetBase.save
    (net/netbase.py:405-424) -- str keys as SHORT_BINSTRING, arrays as py2 NumPy reconstruct tuples with the raw data in a str --
    assembled by tests/golden/make_golden_r2.py.  NetBase.load must read it (encoding='latin1'); a plain pickle.load cannot."""
    import pickle
    import pickletools
    from net.batchnormlayer import BatchNormLayer, BatchNormLayerParams
    from net.convpoollayer import ConvPoolLayer, ConvPoolLayerParams
    from net.netbase import NetBase
    path = os.path.join(GOLD, 'net_py2.pkl')
    want = json.load(open(os.path.join(GOLD, 'net_py2.json')))
    raw = open(path, 'rb').read()
    ops_used = set(op.name for op, _, _ in pickletools.genops(raw))
    assert 'SHORT_BINSTRING' in ops_used and 'BINUNICODE' not in ops_used          # it really is the py2 layout
    with pytest.raises(Exception):
        pickle.loads(raw)                                                            # default ASCII decoding of the array bytes fails

    class _P(object):
        batch_size, inputDim, outputDim, layers = 4, (4, 1, 32, 32), (4, 8, 14, 14), []

    net = NetBase.__new__(NetBase)
    rng = np.random.RandomState(0)
    from hipdp.graph import Var
    x = Var('input', shape=(4, 1, 32, 32))
    l0 = ConvPoolLayer(rng, x, ConvPoolLayerParams(inputDim=(4, 1, 32, 32), nFilters=8, filterDim=(5, 5), poolsize=(2, 2), activation=None), layerNum=0)
    l1 = BatchNormLayer(rng, l0.output, BatchNormLayerParams(inputDim=l0.cfgParams.outputDim), layerNum=1)
    net.layers, net.cfgParams = [l0, l1], _P()
    net.load(path)
    for layer, key in ((l0, '0-values'), (l1, '1-values')):
        got = [p.get_value() for p in layer.params + layer.params_nontrained]
        assert len(got) == len(want['arrays'][key])
        for a, w in zip(got, want['arrays'][key]):
            assert list(a.shape) == w['shape'] and a.dtype == np.float32
            np.testing.assert_array_equal(a.astype(np.float64).ravel(), np.asarray(w['data']))
    # and what NetBase.save writes is read back by the same loader (protocol 2, as cPickle wrote it)
    out = str(tmp_path / 'resaved.pkl')
    net.save(out)
    assert pickle.load(open(out, 'rb'), encoding='latin1')['class'] == 'NetBase'
    l0.W.set_value(np.zeros((8, 1, 5, 5), np.float32))
    net.load(out)
    np.testing.assert_array_equal(l0.W.get_value().astype(np.float64).ravel(), np.asarray(want['arrays']['0-values'][0]['data']))


def test_sincos_cr_is_correctly_rounded():
    """oracle.augment.sincos_cr against mpmath at 400 bits: the rotation coefficients of rotateHand / rotatePoint2D are what the
    reference's libm (glibc 2.19, correctly rounded) returned -- not today's `< 1 ulp` numpy.cos, which differs in the last bit on a
    fraction of a percent of the arguments."""
    mpmath = pytest.importorskip('mpmath')
    from oracle import augment as A
    mpmath.mp.prec = 400
    rng = np.random.RandomState(0)
    vals = list(rng.uniform(-2 * np.pi, 2 * np.pi, 4000)) + [-d * np.pi / 180. for d in range(0, 360)] + \
        [d * np.pi / 180. for d in np.arange(0, 360, 0.5)] + [0.0, 1e-300, -1e-8, 1e-8, 7.9, -7.9]
    off_by_ulp = 0
    for a in vals:
        s, c = A.sincos_cr(a)
        assert s == float(mpmath.sin(mpmath.mpf(a))) and c == float(mpmath.cos(mpmath.mpf(a))), a
        off_by_ulp += int(np.sin(a) != s or np.cos(a) != c)
    assert off_by_ulp < 0.02 * len(vals)                        # libm / NumPy agree almost everywhere: it is the same function
