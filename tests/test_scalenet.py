"""ScaleNet (the multi-scale CoM-refinement net of main_nyu_com_refine.py, /root/reference/src/net/scalenet.py:33-195): three
inputs, conv-pool towers, concatenation, FC + dropout -- engine against the oracle, on the emulator and on the GPU."""
import numpy as np
import pytest

from hipdp import engine
from hipdp import runtime as R
from net.scalenet import ScaleNet, ScaleNetParams
from oracle import nets
from tests.backends import BACKENDS, get_runtime

MM = 150.0


def make(rt, B, seed=23455):
    net = ScaleNet(np.random.RandomState(seed), cfgParams=ScaleNetParams(type=1, batchSize=B, numJoints=1, nDims=3))
    onet = nets.build_scalenet(batchSize=B, numJoints=1, nDims=3)
    P = nets.init_params(onet, np.random.RandomState(seed), np.float32)
    for i in P:
        P[i][1] = np.random.RandomState(seed + i).normal(0, 0.05, P[i][1].shape).astype(np.float32)
    for i, l in enumerate(net.layers):
        if i in P:
            for p, v in zip(l.params, P[i]):
                p.set_value(v)
    assert [tuple(l.cfgParams.outputDim) for l in net.layers] == [tuple(l['out_dim']) for l in onet['layers']]
    return net, onet, P


@pytest.mark.parametrize('backend', BACKENDS)
def test_scalenet_compute_output_matches_oracle(backend):
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    B = 3
    net, onet, P = make(rt, B)
    x = nets.synthetic_crops(np.random.RandomState(5), 5, 128, 128, np.float32)
    xs = nets.scalenet_inputs(x)
    assert [a.shape[2:] for a in xs] == [(128, 128), (64, 64), (32, 32)]
    net.setDeterministic()
    out = net.computeOutput(xs)                      # 5 samples in batches of 3: padded by repeating the last one
    ref = nets.compute_output(onet, nets.cast_params(P, np.float64), [a.astype(np.float64) for a in xs])
    assert out.shape == (5, 3)
    assert np.abs(out - ref).max() * MM < 1e-3
    with pytest.raises(ValueError):
        engine.CompiledNet(net, train=False, runtime=rt).forward(xs[:2])


@pytest.mark.parametrize('backend', BACKENDS)
def test_scalenet_train_gradients_match_oracle(backend):
    rt = get_runtime(backend)
    B = 4
    net, onet, P = make(rt, B)
    rng = np.random.RandomState(9)
    xs = nets.scalenet_inputs(nets.synthetic_crops(rng, B, 128, 128, np.float32))
    y = rng.normal(0, 0.3, (B, 3)).astype(np.float32)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    cost, out = eng.cost_and_grads(xs, y)
    masks = {i: eng.dropout_masks[id(l)][0].get().astype(np.float64) for i, l in enumerate(net.layers) if id(l) in eng.dropout_masks}
    assert len(masks) == 2
    c_ref, G_ref, _, out_ref = nets.cost_and_grads(onet, nets.cast_params(P, np.float64), [a.astype(np.float64) for a in xs],
                                                   y.astype(np.float64), True, masks)
    assert np.abs(out - out_ref).max() * MM < 1e-3
    assert abs(cost - c_ref) < 1e-5 * abs(c_ref)
    gmax = max(np.abs(G_ref[i][s]).max() for i in G_ref for s in range(2))
    for i in G_ref:
        for s in range(2):
            got = eng.store.read_grad(net.layers[i].params[s])
            np.testing.assert_allclose(got, G_ref[i][s], rtol=0, atol=2e-4 * max(np.abs(G_ref[i][s]).max(), 5e-3 * gmax),
                                       err_msg='layer %d slot %d' % (i, s))
    c2 = eng.train_step(xs, y, 1e-3)
    assert np.isfinite(c2)


def _centre(x, k):
    H, W = x.shape[2], x.shape[3]
    h, w = H // k, W // k
    xs, ys = int(H / 2 - h / 2), int(W / 2 - w / 2)
    return np.ascontiguousarray(x[:, :, ys:ys + w, xs:xs + h])


@pytest.mark.parametrize('backend', BACKENDS)
def test_com_refine_script_flow(backend, tmp_path):
    """main_nyu_com_refine.py:148-192 step by step on synthetic data: ScaleNetTrainer.setData / addStaticData /
    addManagedData / compileFunctions / train, then the trained net as HandDetector's refineNet (cropArea3D docom=True)."""
    from data.importers import ICVLImporter
    from oracle import augment as A
    from trainer.scalenettrainer import ScaleNetTrainer, ScaleNetTrainerParams
    from util.handdetector import HandDetector
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    rng = np.random.RandomState(23455)
    di = ICVLImporter('../data/ICVL/')
    cam = A.Camera.icvl()
    B, n_train, n_val = 4, 6, 4
    imgs, coms, cubes, Ms, gts = A.synthetic_augment_inputs(np.random.RandomState(1), n_train + n_val, cam, cube=(250., 250., 250.), joints=16)
    data = imgs[:, None].astype('float32')
    gt3D = (gts / (cubes[:, 2] / 2.)[:, None, None]).astype('float32')
    train_data, val_data = data[:n_train], data[n_train:]
    net = ScaleNet(rng, cfgParams=ScaleNetParams(type=1, nChan=1, wIn=128, hIn=128, batchSize=B, resizeFactor=2, numJoints=1, nDims=3))
    p = ScaleNetTrainerParams()
    p.use_early_stopping = False
    p.batch_size = B
    p.learning_rate = 0.0005
    p.weightreg_factor = 0.0001
    p.force_macrobatch_reload = True
    p.para_augment = True
    p.validation_frequency = 2
    p.snapshot_last = 1
    p.augment_fun_params = {'fun': 'augment_poses', 'args': {'normZeroOne': False, 'di': di, 'aug_modes': ['com', 'rot', 'none'],
                                                             'hd': HandDetector(train_data[0, 0].copy(), abs(di.fx), abs(di.fy), importer=di)}}
    with pytest.raises(ValueError):
        ScaleNetTrainer(net, object(), rng, str(tmp_path))
    tr = ScaleNetTrainer(net, p, rng, str(tmp_path))
    tr.setData(train_data, gt3D[:n_train, di.crop_joint_idx, :], val_data, gt3D[n_train:, di.crop_joint_idx, :])
    tr.addStaticData({'val_data_x1': _centre(val_data, 2), 'val_data_x2': _centre(val_data, 4)})
    tr.addManagedData({'train_data_x1': _centre(train_data, 2), 'train_data_x2': _centre(train_data, 4)})
    tr.addManagedData({'train_data_com': coms[:n_train], 'train_data_cube': cubes[:n_train], 'train_data_M': Ms[:n_train].astype('float32'),
                       'train_gt3D': gt3D[:n_train]})
    tr.compileFunctions()
    costs, _, val_errs = tr.train(n_epochs=1)
    assert len(costs) == 2 and np.all(np.isfinite(costs))
    # the resident macro-batch was re-augmented, and the two small inputs are the centre crops of the augmented crop
    x = tr.train_data_x.get_value()
    assert np.array_equal(tr.train_data_x1.get_value(), _centre(x, 2)) and np.array_equal(tr.train_data_x2.get_value(), _centre(x, 4))
    assert (x[:n_train] != train_data).mean() > 0.01
    assert tr.train_data_y.get_value().shape == (8, 3) and np.isfinite(tr.train_data_y.get_value()).all()
    # the net refines a crop centre: cropArea3D(docom=True) with refineNet goes crop -> CoM -> ScaleNet offset -> crop
    frames, fcoms = A.synthetic_frames(np.random.RandomState(3), 2, cam, 240, 320, (250., 250., 250.))
    net.setDeterministic()
    hd = HandDetector(frames[1].copy(), abs(di.fx), abs(di.fy), importer=di, refineNet=net)
    crop, M, com = hd.cropArea3D(com=fcoms[1], size=(250., 250., 250.), dsize=(128, 128), docom=True)
    assert crop.shape == (128, 128) and M.shape == (3, 3) and np.isfinite(com).all()
    hd0 = HandDetector(frames[1].copy(), abs(di.fx), abs(di.fy), importer=di)
    crop0, _, com0 = hd0.cropArea3D(com=fcoms[1], size=(250., 250., 250.), dsize=(128, 128), docom=True)
    off = hd.refineCoM(crop0, (250., 250., 250.), com0)
    expect = di.joint3DToImg(off + di.jointImgTo3D(com0))
    np.testing.assert_allclose(com, expect, rtol=0, atol=1e-3)


@pytest.mark.parametrize('backend', BACKENDS)
def test_scalenet_shared_conv_towers(backend):
    """ScaleNetParams(shared_conv=True) (scalenet.py:176-180): towers 2 and 3 run on the first tower's filters and biases.  One
    parameter slot per shared weight, forward on all three towers from it, and a gradient that is the SUM over the towers --
    against the oracle; one ADAM step then moves the single copy."""
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    B = 4
    net = ScaleNet(np.random.RandomState(23455), cfgParams=ScaleNetParams(type=1, batchSize=B, numJoints=1, nDims=3, shared_conv=True))
    assert net.layers[3].W is net.layers[0].W and net.layers[7].b is net.layers[1].b and net.layers[8].W is net.layers[2].W
    assert len(net.params) == 2 * 3 + 2 * 3                       # three shared conv layers + three FC layers, each (W, b)
    onet = nets.build_scalenet(batchSize=B, numJoints=1, nDims=3, shared_conv=True)
    P = nets.init_params(onet, np.random.RandomState(23455), np.float32)
    assert P[3] is P[0] and P[8] is P[2]
    for i in (0, 1, 2, 9, 11, 13):
        P[i][1] = np.random.RandomState(100 + i).normal(0, 0.05, P[i][1].shape).astype(np.float32)
    for i, l in enumerate(net.layers):
        if i in P and 'share' not in onet['layers'][i]:
            for p, v in zip(l.params, P[i]):
                p.set_value(v)
    # the same RandomState draws as the reference: copyLayer layers consume nothing, so the FC weights follow the first tower's
    np.testing.assert_array_equal(net.layers[9].W.get_value(), P[9][0])
    rng = np.random.RandomState(9)
    xs = nets.scalenet_inputs(nets.synthetic_crops(rng, B, 128, 128, np.float32))
    y = rng.normal(0, 0.3, (B, 3)).astype(np.float32)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    assert len(eng.store.slots) == 12
    cost, out = eng.cost_and_grads(xs, y)
    masks = {i: eng.dropout_masks[id(l)][0].get().astype(np.float64) for i, l in enumerate(net.layers) if id(l) in eng.dropout_masks}
    # the float64 oracle with the device's own ReLU / pool-tie decisions (tests/pinning.py): a float32 near-tie in one pooling
    # window would otherwise route one gradient differently and move 200 of a filter bank's 1 600 elements by half a percent
    from oracle import torch_ref
    from tests.pinning import device_masks
    c_ref, G_all, out_ref = _torch_with_dropout(torch_ref, onet, nets.cast_params(P, np.float64), xs, y, masks, device_masks(eng, net))
    G_ref = {i: g for i, g in G_all.items() if 'share' not in onet['layers'][i] and onet['layers'][i]['kind'] != 'dropout'}
    assert np.abs(out - out_ref).max() * MM < 1e-3 and abs(cost - c_ref) < 1e-5 * abs(c_ref)
    assert sorted(G_ref) == [0, 1, 2, 9, 11, 13]
    gmax = max(np.abs(G_ref[i][s]).max() for i in G_ref for s in range(2))
    for i in G_ref:
        for s in range(2):
            got = eng.store.read_grad(net.layers[i].params[s])
            np.testing.assert_allclose(got, G_ref[i][s], rtol=0, atol=2e-4 * max(np.abs(G_ref[i][s]).max(), 5e-3 * gmax),
                                       err_msg='layer %d slot %d' % (i, s))
    # the shared gradient really is a sum: the first tower alone gives something else
    _, G_solo, _, _ = nets.cost_and_grads(nets.build_scalenet(batchSize=B, numJoints=1, nDims=3), {i: [a.astype(np.float64) for a in P[i]] for i in P},
                                          [a.astype(np.float64) for a in xs], y.astype(np.float64), True, masks)
    assert np.abs(G_solo[0][0] - G_ref[0][0]).max() > 1e-3 * np.abs(G_ref[0][0]).max()
    # ... and the numpy oracle's sum over the towers agrees with autograd's accumulation into the shared leaves
    _, G_np, _, _ = nets.cost_and_grads(onet, nets.cast_params(P, np.float64), [a.astype(np.float64) for a in xs], y.astype(np.float64), True, masks)
    assert sorted(G_np) == sorted(G_ref)
    w0 = net.layers[0].W.get_value().copy()
    assert np.isfinite(eng.train_step(xs, y, 1e-3))
    w1 = net.layers[0].W.get_value()
    assert np.abs(w1 - w0).max() > 1e-4 and np.array_equal(net.layers[6].W.get_value(), w1)


def _torch_with_dropout(torch_ref, onet, P64, xs, y, dropout_masks, pin):
    """torch_ref.forward models dropout in training mode as the identity: apply the device's masks by folding them into the
    ReLU decision of the HiddenLayer in front of each DropoutLayer (mask * relu(pre) == pre * (mask & pass) for 0/1 masks)."""
    import torch
    pin = dict(pin)
    for i, l in enumerate(onet['layers']):
        if l['kind'] == 'dropout':
            src = l['src'][1]
            pin[src] = np.asarray(pin[src], bool) & (np.asarray(dropout_masks[i]) > 0)
    T = torch_ref.to_torch(P64, torch.float64)
    xt = [torch.as_tensor(a, dtype=torch.float64) for a in xs]
    out, _ = torch_ref.forward(onet, T, xt, True, pin)
    cost = ((out - torch.tensor(y, dtype=torch.float64)) ** 2).sum(dim=1).mean()
    cost.backward()
    G = {}
    for i in T:
        if T[i][0].grad is not None:
            G[i] = [T[i][0].grad.numpy(), T[i][1].grad.numpy()]
    return float(cost.detach()), G, out.detach().numpy()


@pytest.mark.parametrize('backend', BACKENDS)
def test_scalenet_twin_runs_on_the_first_nets_parameters(backend):
    """ScaleNet(rng, cfgParams=, twin=net) (scalenet.py:136, 178): every layer is a copyLayer of the twin's -- the same regressor at
    another batch size.  One device copy of the weights: training the first net moves the twin's outputs, and the twin draws nothing
    from the RandomState."""
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    net, onet, P = make(rt, 4)
    rng = np.random.RandomState(1)
    state = rng.get_state()[1].copy()
    twin = ScaleNet(rng, cfgParams=ScaleNetParams(type=1, batchSize=2, numJoints=1, nDims=3), twin=net)
    assert all(a.W is b.W and a.b is b.b for a, b in zip(net.layers, twin.layers) if hasattr(a, 'W'))
    assert np.array_equal(rng.get_state()[1], state)
    x = nets.synthetic_crops(np.random.RandomState(5), 4, 128, 128, np.float32)
    xs = nets.scalenet_inputs(x)
    net.setDeterministic()
    twin.setDeterministic()
    o4, o2 = net.computeOutput(xs), twin.computeOutput(xs)                # batches of 4 and of 2
    assert np.abs(o4 - o2).max() * MM < 1e-4
    assert twin._param_store is net._param_store
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    eng.train_step(xs, np.random.RandomState(2).normal(0, 0.3, (4, 3)).astype(np.float32), 1e-2)
    n4, n2 = net.computeOutput(xs), twin.computeOutput(xs)
    assert np.abs(n4 - o4).max() > 1e-4 and np.abs(n4 - n2).max() * MM < 1e-4
