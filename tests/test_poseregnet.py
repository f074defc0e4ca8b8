"""PoseRegNet (the net the shipped main_*_posereg_embedding.py scripts build, /root/reference/src/net/poseregnet.py:60-143):
generic conv-pool kernels + FC + dropout through the engine against the oracle, same bodies on the emulator and on the GPU."""
import numpy as np
import pytest

from hipdp import engine
from net.poseregnet import PoseRegNet, PoseRegNetParams
from oracle import nets
from tests.backends import BACKENDS, get_runtime

MM = 150.0


def make(rt, type_, B, size, nJ, nD, seed=23455):
    net = PoseRegNet(np.random.RandomState(seed), cfgParams=PoseRegNetParams(type=type_, wIn=size, hIn=size, batchSize=B, numJoints=nJ, nDims=nD))
    onet = nets.build_poseregnet(type=type_, wIn=size, hIn=size, batchSize=B, numJoints=nJ, nDims=nD)
    P = nets.init_params(onet, np.random.RandomState(seed), np.float32)
    for i, l in enumerate(onet['layers']):            # non-zero biases so that bias-after-pool is exercised
        if i in P:
            P[i][1] = np.random.RandomState(seed + i).normal(0, 0.05, P[i][1].shape).astype(np.float32)
    for i, l in enumerate(net.layers):
        if i in P:
            for p, v in zip(l.params, P[i]):
                p.set_value(v)
    return net, onet, P


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('type_', [0, 11])
def test_poseregnet_eval_forward_matches_oracle(backend, type_):
    rt = get_runtime(backend)
    B = 3
    net, onet, P = make(rt, type_, B, 128, 14, 3)
    x = nets.synthetic_crops(np.random.RandomState(5), B, 128, 128, np.float32)
    eng = engine.CompiledNet(net, train=False, runtime=rt)
    out = eng.forward(x)
    ref, _ = nets.forward(onet, nets.cast_params(P, np.float64), x.astype(np.float64), False)
    assert out.shape == ref.shape == (B, 42)
    assert np.abs(out - ref).max() * MM < 1e-3          # the north-star bar: 1e-3 mm on a 300 mm cube


@pytest.mark.parametrize('backend', BACKENDS)
def test_poseregnet_train_forward_backward_matches_oracle(backend):
    rt = get_runtime(backend)
    B = 4
    net, onet, P = make(rt, 0, B, 128, 1, 30)
    rng = np.random.RandomState(8)
    x = nets.synthetic_crops(rng, B, 128, 128, np.float32)
    y = rng.normal(0, 0.3, (B, 30)).astype(np.float32)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    cost, out = eng.cost_and_grads(x, y)
    # the device drew the dropout masks; the oracle replays them (the reference's MRG stream is not reproduced)
    masks = {}
    for i, l in enumerate(net.layers):
        if id(l) in eng.dropout_masks:
            m = eng.dropout_masks[id(l)][0].get()
            assert set(np.unique(m)) <= {0.0, 1.0} and 0.5 < m.mean() < 0.9
            masks[i] = m.astype(np.float64)
    assert len(masks) == 2
    P64 = nets.cast_params(P, np.float64)
    c_ref, G_ref, _, out_ref = nets.cost_and_grads(onet, P64, x.astype(np.float64), y.astype(np.float64), True, masks)
    assert np.abs(out - out_ref).max() * MM < 1e-3
    assert abs(cost - c_ref) < 1e-5 * abs(c_ref)
    gmax = max(np.abs(G_ref[i][s]).max() for i in G_ref for s in range(2))
    for i in G_ref:
        for s in range(2):
            got = eng.store.read_grad(net.layers[i].params[s])
            np.testing.assert_allclose(got, G_ref[i][s], rtol=0, atol=2e-4 * max(np.abs(G_ref[i][s]).max(), 5e-3 * gmax),
                                       err_msg='layer %d slot %d' % (i, s))
    # a second step draws different masks
    m0 = [eng.dropout_masks[id(l)][0].get().copy() for l in net.layers if id(l) in eng.dropout_masks]
    eng.train_step(x, y, 1e-3)
    eng.train_step(x, y, 1e-3)
    m1 = [eng.dropout_masks[id(l)][0].get() for l in net.layers if id(l) in eng.dropout_masks]
    assert all((a != b).any() for a, b in zip(m0, m1))
    assert m0[0].shape == m0[1].shape and (m1[0] != m1[1]).any()          # the two layers use different streams


@pytest.mark.parametrize('backend', BACKENDS)
def test_joint_regression_cost_and_monitors(backend):
    """Direct joint regression (numJoints > 1): cost = mean_n mean_j sum_d (out - y)^2 (poseregnettrainer.py:97), its gradients,
    and the validation monitor mean_n mean_j ||out - y|| (poseregnettrainer.py:127-129)."""
    rt = get_runtime(backend)
    B, J = 4, 14
    net, onet, P = make(rt, 0, B, 128, J, 3)
    rng = np.random.RandomState(10)
    x = nets.synthetic_crops(rng, B, 128, 128, np.float32)
    y = rng.normal(0, 0.3, (B, J * 3)).astype(np.float32)
    loss = dict(kind='joints', numJoints=J, nDims=3)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=loss)
    cost, out = eng.cost_and_grads(x, y)
    masks = {i: eng.dropout_masks[id(l)][0].get().astype(np.float64) for i, l in enumerate(net.layers) if id(l) in eng.dropout_masks}
    c_ref, G_ref, _, out_ref = nets.cost_and_grads(onet, nets.cast_params(P, np.float64), x.astype(np.float64), y.astype(np.float64), True,
                                                   masks, joints=(J, 3))
    assert abs(cost - c_ref) < 1e-5 * abs(c_ref) and np.abs(out - out_ref).max() * MM < 1e-3
    gmax = max(np.abs(G_ref[i][s]).max() for i in G_ref for s in range(2))
    for i in G_ref:
        for s in range(2):
            np.testing.assert_allclose(eng.store.read_grad(net.layers[i].params[s]), G_ref[i][s], rtol=0,
                                       atol=2e-4 * max(np.abs(G_ref[i][s]).max(), 5e-3 * gmax), err_msg='layer %d slot %d' % (i, s))
    ev = engine.CompiledNet(net, train=False, runtime=rt, loss=loss)
    c_eval, err = ev.evaluate(x, y)
    o_eval, _ = nets.forward(onet, nets.cast_params(P, np.float64), x.astype(np.float64), False)
    d = (o_eval - y).reshape(B, J, 3)
    assert abs(c_eval - (d * d).sum(axis=2).mean(axis=1).mean()) < 1e-5 * c_eval
    assert abs(err - np.sqrt((d * d).sum(axis=2)).mean(axis=1).mean()) < 1e-5 * err
