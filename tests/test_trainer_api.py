"""The drop-in boundary end to end, following /root/reference/src/main_nyu_posereg_embedding.py:97-158 step by step on
synthetic data: PoseRegNetTrainerParams / PoseRegNetTrainer.setData / addStaticData / addManagedData / compileFunctions /
train, then the script's post-training surgery (append the PCA-prior HiddenLayer, re-point net.output, save, reload,
computeOutput)."""
import os

import numpy as np
import pytest
from sklearn.decomposition import PCA

from data.importers import ICVLImporter
from hipdp import runtime as R
from net.hiddenlayer import HiddenLayer, HiddenLayerParams
from net.poseregnet import PoseRegNet, PoseRegNetParams
from net.resnet import ResNet, ResNetParams
from oracle import augment as A
from tests.backends import BACKENDS, get_runtime
from trainer.poseregnettrainer import PoseRegNetTrainer, PoseRegNetTrainerParams
from util.handdetector import HandDetector


def synth(n, size, J, seed):
    rng = np.random.RandomState(seed)
    cam = A.Camera.icvl()
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(rng, n, cam, cube=(250., 250., 250.), joints=J, dsize=size)
    return imgs[:, None], coms, cubes, Ms, gts


FAMILIES = {
    # the ResNet of BASELINE configs 2-4, and the DeepPose-style PoseRegNet the shipped mains literally build
    # (main_nyu_posereg_embedding.py:97-98); sizes are the smallest maps the two architectures admit
    'resnet': (ResNet, ResNetParams, 32),
    'poseregnet': (PoseRegNet, PoseRegNetParams, 48),
}


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('family', sorted(FAMILIES))
def test_main_script_flow(backend, family, tmp_path):
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    rng = np.random.RandomState(23455)
    Net, NetParams, size = FAMILIES[family]
    J, B, E = 16, 4, 8
    di = ICVLImporter('../data/ICVL/')
    train_data, train_com, train_cube, train_M, train_gt3Dcrop = synth(6, size, J, 1)
    val_data, _, val_cube, _, val_gt3Dcrop = synth(4, size, J, 2)
    train_gt3D = (train_gt3Dcrop / (train_cube[:, 2] / 2.)[:, None, None]).astype('float32')
    val_gt3D = (val_gt3Dcrop / (val_cube[:, 2] / 2.)[:, None, None]).astype('float32')
    aug_modes = ['com', 'rot', 'none']
    pca = PCA(n_components=E)
    pca.fit(HandDetector.sampleRandomPoses(di, rng, train_gt3Dcrop, train_com, train_cube, 200, aug_modes).reshape((-1, J * 3)))
    train_embed = pca.transform(train_gt3D.reshape((-1, J * 3))).astype('float32')
    val_embed = pca.transform(val_gt3D.reshape((-1, J * 3))).astype('float32')

    poseNetParams = NetParams(type=0, nChan=1, wIn=size, hIn=size, batchSize=B, numJoints=1, nDims=E)
    poseNet = Net(rng, cfgParams=poseNetParams)
    p = PoseRegNetTrainerParams()
    p.batch_size = B
    p.learning_rate = 0.001
    p.weightreg_factor = 0.0
    p.force_macrobatch_reload = True
    p.para_augment = True
    p.validation_frequency = 2
    p.snapshot_last = 1
    p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': aug_modes,
                                                             'hd': HandDetector(train_data[0, 0].copy(), abs(di.fx), abs(di.fy), importer=di),
                                                             'proj': pca}}
    with pytest.raises(ValueError):
        PoseRegNetTrainer(poseNet, object(), rng, str(tmp_path))
    tr = PoseRegNetTrainer(poseNet, p, rng, str(tmp_path))
    tr.setData(train_data, train_embed, val_data, val_embed)
    tr.addStaticData({'val_data_y3D': val_gt3D})
    tr.addStaticData({'pca_data': pca.components_.astype('float32'), 'mean_data': pca.mean_.astype('float32')})
    with pytest.raises(ValueError):
        tr.addManagedData({'train_data_cube': train_cube[:5]})
    tr.addManagedData({'train_data_cube': train_cube, 'train_data_com': train_com, 'train_data_M': train_M.astype('float32'),
                       'train_gt3Dcrop': train_gt3Dcrop})
    assert tr.getNumMiniBatches() == 2 and tr.train_data_xDB.shape[0] == 8           # padded with seeded random samples
    pad_rng = np.random.RandomState(6)
    assert np.array_equal(tr.train_data_xDB[6], train_data[pad_rng.randint(0, 6)])
    tr.compileFunctions(compileDebugFcts=False)
    x_before = tr.train_data_x.get_value().copy()
    costs, wvals, val_obs = tr.train(n_epochs=1)
    assert len(costs) == 2 and np.all(np.isfinite(costs))
    assert len(val_obs) == 3 and all(len(v) == 2 for v in val_obs)        # error, error_avg, error_max: initial + 1 validation
    x_after = tr.train_data_x.get_value()
    assert x_after.shape == x_before.shape and np.isfinite(x_after).all()
    assert (x_after != x_before).mean() > 0.01                               # the resident macro-batch was re-augmented in place
    assert x_after.min() >= -1.0 - 1e-5 and x_after.max() <= 1.0 + 1e-5
    assert os.path.exists(os.path.join(str(tmp_path), 'net_last.pkl'))

    # ---- post-training surgery of the script (main_nyu_posereg_embedding.py:143-158) ----
    out_embed = poseNet.computeOutput(val_data)
    assert out_embed.shape == (4, E)
    cfg = HiddenLayerParams(inputDim=(B, E), outputDim=(B, J * 3), activation=None)
    pcalayer = HiddenLayer(rng, poseNet.layers[-1].output, cfg, layerNum=len(poseNet.layers))
    pcalayer.W.set_value(pca.components_.astype('float32'))
    pcalayer.b.set_value(pca.mean_.astype('float32'))
    poseNet.layers.append(pcalayer)
    poseNet.output = pcalayer.output
    poseNet.cfgParams.numJoints = J
    poseNet.cfgParams.nDims = 3
    poseNet.cfgParams.outputDim = pcalayer.cfgParams.outputDim
    fn = os.path.join(str(tmp_path), 'network_prior.pkl')
    poseNet.save(fn)
    jts = poseNet.computeOutput(val_data)
    assert jts.shape == (4, J * 3)
    np.testing.assert_allclose(jts, out_embed.astype('f8') @ pca.components_ + pca.mean_, rtol=0, atol=2e-5)
    # reload into a freshly built net with the same surgery: identical outputs
    net2 = Net(np.random.RandomState(1), cfgParams=NetParams(type=0, nChan=1, wIn=size, hIn=size, batchSize=B, numJoints=1, nDims=E))
    l2 = HiddenLayer(rng, net2.layers[-1].output, HiddenLayerParams(inputDim=(B, E), outputDim=(B, J * 3), activation=None),
                     layerNum=len(net2.layers))
    net2.layers.append(l2)
    net2.output = l2.output
    net2.cfgParams.numJoints, net2.cfgParams.nDims, net2.cfgParams.outputDim = J, 3, l2.cfgParams.outputDim
    net2.load(fn)
    net2.setDeterministic()
    np.testing.assert_allclose(net2.computeOutput(val_data), jts, rtol=0, atol=1e-6)


def _sample_random_poses_loop(importer, rng, base_poses, base_com, base_cube, num_poses, aug_modes, sigma_com=5., sigma_sc=0.02, rot_range=180.):
    """The reference's per-sample loop (handdetector.py:805-909) written out with the importer's scalar functions: the
    yardstick for the vectorised HandDetector.sampleRandomPoses.  float32 arrays times float64 scalars stay float32, as in
    the NumPy the reference ran on."""
    from data.transformations import rotatePoints2D
    n = int(num_poses)
    new_poses = np.zeros((n, base_poses.shape[1], base_poses.shape[2]), dtype=base_poses.dtype)
    modes = rng.randint(0, len(aug_modes), n)
    ridxs = rng.randint(0, base_poses.shape[0], n)
    off = rng.randn(n, 3) * sigma_com
    sc = np.fabs(rng.randn(n) * sigma_sc + 1.)
    rot = rng.uniform(-rot_range, rot_range, size=(n, 3))
    for i in range(n):
        mode = aug_modes[modes[i]]
        cube, com3D, pose = base_cube[ridxs[i]], base_com[ridxs[i]], base_poses[ridxs[i]]
        if mode == 'com':
            nc = (com3D + off[i]).astype(np.float32)
            new_poses[i] = (pose + com3D - nc) / (cube[2] / 2.)
        elif mode == 'rot':
            joint_2D = importer.joints3DToImg(pose + com3D)
            data_2D = rotatePoints2D(joint_2D, importer.joint3DToImg(com3D)[0:2], rot[i, 0])
            new_poses[i] = (importer.jointsImgTo3D(data_2D) - com3D) / (cube[2] / 2.)
        elif mode == 'sc':
            new_poses[i] = pose / ((cube * np.float32(sc[i]))[2] / 2.)
        elif mode == 'none':
            new_poses[i] = pose / (cube[2] / 2.)
        else:
            nc = (com3D + off[i]).astype(np.float32)
            p = pose + com3D - nc
            if 'sc' in mode:
                p = p * np.float32(sc[i])
            joint_2D = importer.joints3DToImg(p + com3D)
            data_2D = rotatePoints2D(joint_2D, importer.joint3DToImg(nc)[0:2], rot[i, 0])
            new_poses[i] = (importer.jointsImgTo3D(data_2D) - com3D) / (cube[2] / 2.)
    return new_poses


@pytest.mark.parametrize('flip', [False, True])
def test_sample_random_poses_vectorised_equals_loop(flip):
    from data.importers import ICVLImporter, NYUImporter
    di = NYUImporter('../data/NYU/') if flip else ICVLImporter('../data/ICVL/')
    J = 14 if flip else 16
    cam = A.Camera.nyu() if flip else A.Camera.icvl()
    _, coms, cubes, _, gts = A.synthetic_augment_inputs(np.random.RandomState(4), 20, cam, cube=(300., 300., 300.), joints=J)
    cubes = (cubes * np.random.RandomState(5).uniform(0.8, 1.2, (20, 1))).astype(np.float32)
    modes = ['com', 'rot', 'sc', 'none', 'rot+com', 'rot+com+sc']
    got = HandDetector.sampleRandomPoses(di, np.random.RandomState(6), gts, coms, cubes, 600, modes)
    ref = _sample_random_poses_loop(di, np.random.RandomState(6), gts, coms, cubes, 600, modes)
    assert got.shape == ref.shape == (600, J, 3) and got.dtype == ref.dtype
    assert np.array_equal(got, ref)


def _small_trainer(rt, tmp_path, n_train, budget_mb=None, augment=False, numChunks=1, para_load=False, seed=23455):
    """A PoseRegNetTrainer on the small conv-pool PoseRegNet (48x48 crops, batch 4, labels = the 16 x 3 joints themselves, no PCA
    prior): the paging logic under test is the trainer's, the cheapest net keeps the emulator tier quick."""
    R.set_default_runtime(rt)
    rng = np.random.RandomState(seed)
    J, B, size = 16, 4, 48
    di = ICVLImporter('../data/ICVL/')
    x, com, cube, M, gt = synth(n_train, size, J, 1)
    vx, _, vcube, _, vgt = synth(4, size, J, 2)
    y = (gt / (cube[:, 2] / 2.)[:, None, None]).astype('float32').reshape(n_train, J * 3)
    vy = (vgt / (vcube[:, 2] / 2.)[:, None, None]).astype('float32').reshape(4, J * 3)
    net = PoseRegNet(rng, cfgParams=PoseRegNetParams(type=0, nChan=1, wIn=size, hIn=size, batchSize=B, numJoints=J, nDims=3))
    p = PoseRegNetTrainerParams()
    p.batch_size, p.learning_rate, p.validation_frequency, p.snapshot_last = B, 0.001, 1000, 1000
    p.para_load = para_load
    if augment:
        p.para_augment = True
        p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': ['com', 'rot', 'none'], 'proj': None,
                                                                 'hd': HandDetector(x[0, 0].copy(), abs(di.fx), abs(di.fy), importer=di)}}
    os.makedirs(str(tmp_path), exist_ok=True)
    tr = PoseRegNetTrainer(net, p, rng, str(tmp_path), numChunks=numChunks)
    if budget_mb is not None:
        tr.memorySize = budget_mb                                # what `free device memory / memory_factor` would have given
    tr.setData(x, y, vx, vy)
    tr.addManagedData({'train_data_cube': cube, 'train_data_com': com, 'train_data_M': M.astype('float32')})
    tr.compileFunctions(compileDebugFcts=False)
    return tr, net, (x, y, com, cube, M)


@pytest.mark.parametrize('backend', BACKENDS)
def test_paged_training_set_trains_like_the_resident_one(backend, tmp_path):
    """memory_factor-limited runs (nettrainer.py:259-276, 489-599): a training set larger than the device budget is split into
    macro-batches on the host (keyDB + the padded keyDBlast) and paged through a one-macro-batch device window.  Without
    augmentation the minibatch sequence is the same as with everything resident, so the trained weights must be bit-identical."""
    rt = get_runtime(backend)
    n = 22                                                          # 6 minibatches of 4; sample = 48*48*4 B = 9 KB
    res, net_r, _ = _small_trainer(rt, tmp_path / 'r', n)
    assert res.getNumMacroBatches() == 1
    costs_r, _, _ = res.train(n_epochs=1)
    w_r = [p.get_value().copy() for p in net_r.params]
    pag, net_p, (x, y, com, cube, M) = _small_trainer(rt, tmp_path / 'p', n, budget_mb=2.2 * 4 * 9216 / 1024. ** 2)
    nmb, spm = pag.getNumMacroBatches(), pag.getNumSamplesPerMacroBatch()
    assert nmb == 3 and spm == 8 and pag.getNumMiniBatches() == 6
    assert pag.train_data_xDB.shape[0] == 16 and pag.train_data_xDBlast.shape[0] == 8 and pag.train_data_x.shape[0] == 8
    assert 'train_data_x' in pag.managedVar and 'train_data_com' in pag.managedVar
    # the padded last macro-batch is filled from the WHOLE training set (alignData(fillData=...), nettrainer.py:263)
    pad_rng = np.random.RandomState(6)
    assert np.array_equal(pag.train_data_xDBlast[6], x[pad_rng.randint(0, n)])
    costs_p, _, _ = pag.train(n_epochs=1)
    # NOTE the reference pads the single resident macro-batch with RandomState(22) and the last paged one with RandomState(6):
    # the first five minibatches see identical data, the sixth differs by its two padding samples
    np.testing.assert_array_equal(np.asarray(costs_p[:5], np.float32), np.asarray(costs_r[:5], np.float32))
    assert np.isfinite(costs_p).all() and pag.currentMacroBatch == 2
    assert np.array_equal(pag.train_data_x.get_value(), pag.train_data_xDBlast)      # the window holds the last macro-batch
    # second epoch wraps around to macro-batch 0 (served by the prefetch where the runtime has a copy stream)
    pag.train(n_epochs=1)
    assert pag.currentMacroBatch == 2 and all(np.isfinite(p.get_value()).all() for p in net_p.params)
    assert len(w_r) == len(net_p.params)


@pytest.mark.parametrize('backend', BACKENDS)
def test_paged_training_set_with_device_augmentation(backend, tmp_path):
    """The same paging with the augmentation hook: every macro-batch is uploaded un-augmented, augmented on the device into the
    window (labels = joints: formed from label * cube_z / 2 on the device, poseregnettrainer.py:228-240), and its 'none'-mode
    samples come out untouched (rows of the right macro-batch: a mix-up of windows would change them)."""
    rt = get_runtime(backend)
    tr, net, (x, y, com, cube, M) = _small_trainer(rt, tmp_path, 22, budget_mb=2.2 * 4 * 9216 / 1024. ** 2, augment=True)
    assert tr.getNumMacroBatches() == 3
    seen = []
    orig = tr.augment_poses

    def spy(params, macro_idx, last, tidxs, idxs, new_data):
        orig(params, macro_idx, last, tidxs, idxs, new_data)
        src = tr.train_data_xDBlast if last else tr.train_data_xDB[macro_idx * 8:(macro_idx + 1) * 8]
        ysrc = tr.train_data_yDBlast if last else tr.train_data_yDB[macro_idx * 8:(macro_idx + 1) * 8]
        out, oy = tr.train_data_x.get_value(), tr.train_data_y.get_value()
        same = [i for i in range(8) if np.allclose(out[i], src[i], atol=2e-6)]
        seen.append((macro_idx, last, len(same)))
        assert np.isfinite(out).all() and out.min() >= -1 - 1e-5 and out.max() <= 1 + 1e-5
        # a sample whose label did not move is a 'none' draw: its crop is the uploaded one, bit for bit up to the f32 normalisation
        # round trip (the converse does not hold at 32x32: a CoM shift below one pixel = 7.8 mm moves the label only)
        for i in range(8):
            if np.allclose(oy[i], ysrc[i], atol=1e-6):
                assert i in same, (macro_idx, i)
        assert any(not np.allclose(out[i], src[i], atol=2e-6) for i in range(8))
        assert np.isfinite(oy).all() and np.abs(oy - ysrc).max() < 3.0
    tr.augment_poses = spy
    costs, _, _ = tr.train(n_epochs=1)
    assert [s[0] for s in seen] == [0, 1, 2] and seen[-1][1] is True and np.isfinite(costs).all()


@pytest.mark.parametrize('backend', BACKENDS)
def test_para_load_swaps_chunks_at_the_last_macro_batch(backend, tmp_path):
    """para_load (nettrainer.py:512-526, 630-655, 701-723): while a chunk trains, load_fun_params['fun'] prepares the next one
    on a host thread; at the chunk's last macro-batch the host arrays are replaced and the following chunk is requested."""
    rt = get_runtime(backend)
    with pytest.raises(ValueError):
        _small_trainer(rt, tmp_path, 8, para_load=True, numChunks=1)
    tr, net, (x, y, com, cube, M) = _small_trainer(rt, tmp_path, 8, para_load=True, numChunks=3)
    loaded = []

    def load_chunk(params, chunk_idx, last, data_queue):
        loaded.append(chunk_idx)
        for var, arr in data_queue.items():
            base = {'train_data_x': x, 'train_data_y': y, 'train_data_cube': cube, 'train_data_com': com, 'train_data_M': M}[var]
            arr[:] = np.roll(base, chunk_idx + 1, axis=0)          # a recognisable "next chunk"
    tr.load_chunk = load_chunk
    tr.cfgParams.load_fun_params = {'fun': 'load_chunk', 'args': {}}
    # with a single macro-batch the reference only reloads when force_macrobatch_reload asks for it: before the last minibatch of
    # every epoch (nettrainer.py:528)
    tr.cfgParams.force_macrobatch_reload = True
    tr.train(n_epochs=2)
    # first use (chunk 0 received, 1 requested), end of epoch 1 (chunk 1, 2 requested), end of epoch 2 (chunk 2, 0 requested)
    assert loaded[:4] == [0, 1, 2, 0] and tr.currentChunk == 2
    np.testing.assert_array_equal(tr.train_data_xDB, np.roll(x, 3, axis=0))
    np.testing.assert_array_equal(tr.train_data_x.get_value(), np.roll(x, 3, axis=0))
    np.testing.assert_array_equal(tr.train_data_com.get_value(), np.roll(com, 3, axis=0))
    assert tr._load_thread is None                                 # the worker was shut down by train()


@pytest.mark.parametrize('backend', BACKENDS)
def test_scalar_regression_target(backend, tmp_path):
    """numJoints == nDims == 1 (poseregnettrainer.py:84-85, 92-93, 115): y is a VECTOR, the (B, 1) output broadcasts against it, so cost
    and monitor run over all (i, j) pairs -- what the reference's graph computes.  Cost, gradient (through one ADAM step's direction)
    and validation error against NumPy's own broadcasting."""
    from hipdp import engine
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    rng = np.random.RandomState(23455)
    B, size = 4, 48
    net = PoseRegNet(rng, cfgParams=PoseRegNetParams(type=0, nChan=1, wIn=size, hIn=size, batchSize=B, numJoints=1, nDims=1))
    p = PoseRegNetTrainerParams()
    p.batch_size = B
    tr = PoseRegNetTrainer(net, p, rng, str(tmp_path))
    assert tr.loss_cfg == dict(kind='scalar')
    x = synth(B, size, 16, 3)[0]
    y = np.random.RandomState(4).normal(0, 0.3, B).astype(np.float32)
    net.setDeterministic()
    ev = engine.CompiledNet(net, train=False, runtime=rt, loss=tr.loss_cfg)
    cost, err = ev.evaluate(x, y)
    o = ev.out.buf.get().astype(np.float64)                    # (B, 1)
    d = o.reshape(B, 1) - y.astype(np.float64)                 # NumPy broadcasts like Theano here: (B, B)
    assert abs(cost - (d ** 2).mean(axis=1).mean()) < 1e-6 * max(1.0, (d ** 2).mean())
    assert abs(err - np.sqrt(d ** 2).mean(axis=1).mean()) < 1e-6
    net.unsetDeterministic()
    te = engine.CompiledNet(net, train=True, runtime=rt, loss=tr.loss_cfg)
    c, _ = te.cost_and_grads(x, y)
    ot = te.out.buf.get().astype(np.float64).reshape(B)
    np.testing.assert_allclose(te.out.grad.get().reshape(B), 2.0 / B * (ot - y.mean(dtype=np.float64)), rtol=1e-5, atol=1e-7)
    assert abs(c - ((ot[:, None] - y[None, :].astype(np.float64)) ** 2).mean()) < 1e-6 * max(1.0, c)


def test_parameter_and_weight_block_lists():
    """NetBase.params / params_filter / weights / weights_filter (/root/reference/src/net/netbase.py:157-216): `params` lists every shared
    variable of the layers once, without the ones on the block list; assigning a block list checks membership (UserWarning); the
    weights' block list is compared by `name` against the blocked `auto_name`s in the reference, i.e. it never hides anything -- kept."""
    from hipdp.graph import SharedParam
    net = PoseRegNet(np.random.RandomState(1), cfgParams=PoseRegNetParams(type=0, nChan=1, wIn=48, hIn=48, batchSize=2, numJoints=3, nDims=3))
    every = net.all_params
    assert [p.auto_name for p in net.params] == [p.auto_name for p in every] and len(every) == len(set(p.auto_name for p in every)) > 4
    assert net.params_filter == [] and net.weights_filter == []
    net.params_filter = [every[0], every[3]]
    assert [p.auto_name for p in net.params] == [p.auto_name for p in every if p not in (every[0], every[3])]
    assert net.all_params == every                                   # the block list does not touch the full enumeration
    stranger = SharedParam(np.zeros(3, np.float32), name='not_in_the_model')
    with pytest.raises(UserWarning, match='Param'):
        net.params_filter = [stranger]
    assert net.params_filter == [every[0], every[3]]                 # a rejected list leaves the old one in place
    net.params_filter = []
    assert len(net.params) == len(every)
    ws = net.all_weights
    assert 0 < len(ws) < len(every) and all(w in every for w in ws)
    net.weights_filter = [ws[0]]
    assert net.weights_filter == [ws[0]] and len(net.weights) == len(ws)      # the reference's name / auto_name comparison: a no-op
    with pytest.raises(UserWarning, match='Weight'):
        net.weights_filter = [stranger]
    assert str(net).startswith("Network configuration:\nLayer 0: ConvPoolLayer with ")
    assert str(net).count('\n') == len(net.layers) + 1 and str(net).endswith(' \n')
