"""Full-size parity on the MI355X (BASELINE config 2's geometry: 128x128 crops, the 47-layer ResNet): forward joints and
train-step gradients against the float64 oracle at a batch the oracle finishes in seconds, and size-independent
properties at the full batch of 128 (determinism, batch-composition independence of the deterministic forward, the
PCA-prior layer as an exact affine map of the embedding, ADAM's first step = -lr * sign(g))."""
import numpy as np
import pytest

from hipdp import engine
from hipdp import runtime as R
from oracle import nets
from tests.backends import get_runtime
from tests.test_engine import MM, grads_from_store, make_net

pytestmark = pytest.mark.gpu


def test_forward_joints_within_1e3_mm_at_128():
    rt = get_runtime('hip')
    R.set_default_runtime(rt)
    B = 8
    net, onet, P = make_net(rt, 1, B, 128, 14, 3)              # type 1: 30-D bottleneck + PCA-prior layer, NYU's 14 joints
    x = nets.synthetic_crops(np.random.RandomState(5), 11, 128, 128, np.float32)
    net.setDeterministic()
    out = net.computeOutput(x)                                 # 11 frames, batch 8: padded by repeating the last frame
    ref = nets.compute_output(onet, nets.cast_params(P, np.float64), x.astype(np.float64))
    assert out.shape == (11, 42) and np.abs(ref).max() > 0.05
    err_mm = np.abs(out - ref).max() * MM
    assert err_mm < 1e-3, err_mm                               # the north-star bar on a 300 mm cube


def test_train_gradients_match_oracle_at_128():
    rt = get_runtime('hip')
    B = 8
    net, onet, P = make_net(rt, 0, B, 128, 1, 30)
    rng = np.random.RandomState(6)
    x = nets.synthetic_crops(rng, B, 128, 128, np.float32)
    y = rng.normal(0, 0.3, (B, 30)).astype(np.float32)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    cost, out = eng.cost_and_grads(x, y)
    c_ref, G_ref, _, out_ref = nets.cost_and_grads(onet, nets.cast_params(P, np.float64), x.astype(np.float64), y.astype(np.float64))
    assert np.abs(out - out_ref).max() * MM < 1e-3
    assert abs(cost - c_ref) < 1e-5 * abs(c_ref)
    G = grads_from_store(eng, net)
    gmax = max(np.abs(G_ref[i][s]).max() for i in G_ref for s in range(2))
    # the same graph evaluated by the oracle in float32 (what Theano's floatX = float32 computes) calibrates how much of
    # the distance to float64 is float32 itself after 190 layers
    _, G32, _, _ = nets.cost_and_grads(onet, P, x, y)
    # float32 rounding flips the ReLU mask of the few activations that sit within ~1e-7 of the kink (out of 10^7 per batch).
    # One flip in stage 4 (a few hundred values per channel, BatchNorm backward downstream) moves every upstream gradient by
    # a fraction of a percent -- in the reference's float32 graph just as here: the float32 ORACLE is ~5e-3 (L2, relative)
    # away from float64 on most tensors, and which element flips differs between any two float32 evaluations.  So the bar
    # is the float32 oracle's own worst distance to float64: every device tensor within 4x of it (1e-3 where nothing flips).
    def rel_err(Gx, i, s):
        ref = G_ref[i][s]
        return np.linalg.norm(Gx[i][s] - ref) / max(np.linalg.norm(ref), 5e-3 * gmax * np.sqrt(ref.size))

    worst32 = max(rel_err(G32, i, s) for i in G_ref for s in range(2))
    for i in G_ref:
        for s in range(2):
            r = rel_err(G, i, s)
            assert r < max(1e-3, 4 * worst32), ('layer %d slot %d' % (i, s), r, worst32)
    assert worst32 < 2e-2


def test_full_batch_properties():
    rt = get_runtime('hip')
    R.set_default_runtime(rt)
    B = 128
    net, onet, P = make_net(rt, 1, 8, 128, 14, 3)              # parameters / running statistics of a calibrated net ...
    from net.resnet import ResNet, ResNetParams
    big = ResNet(np.random.RandomState(1), cfgParams=ResNetParams(type=1, nChan=1, wIn=128, hIn=128, batchSize=B, numJoints=14, nDims=3))
    for lb, ls in zip(big.layers, net.layers):                 # ... copied into the batch-128 net
        for pb, ps in zip(lb.params + lb.params_nontrained, ls.params + ls.params_nontrained):
            pb.set_value(ps.get_value())
    x = nets.synthetic_crops(np.random.RandomState(9), B, 128, 128, np.float32)
    big.setDeterministic()
    net.setDeterministic()
    o1 = big.computeOutput(x)
    o2 = big.computeOutput(x)
    assert np.array_equal(o1, o2)                              # deterministic kernels: bit-identical replays
    o8 = net.computeOutput(x)                                  # the same frames in batches of 8
    assert np.abs(o1 - o8).max() * MM < 1e-4                   # deterministic-mode BN: a frame's joints do not depend on its batch
    perm = np.random.RandomState(3).permutation(B)
    assert np.abs(big.computeOutput(x[perm]) - o1[perm]).max() * MM < 1e-4
    # the last layer is the PCA prior: joints = embedding . W + b exactly (hiddenlayer.py:136-139 with activation None)
    emb_layer = big.layers[-2]
    eng = engine.CompiledNet(big, train=False, runtime=rt)
    eng.forward(x)
    emb = None
    for t in eng.tensors:
        if t.name == 'fc%d' % emb_layer.layerNum:
            emb = t.buf.get()
    W, b = big.layers[-1].W.get_value().astype('f8'), big.layers[-1].b.get_value().astype('f8')
    np.testing.assert_allclose(o1, emb.astype('f8') @ W + b, rtol=0, atol=2e-6 * max(1.0, np.abs(o1).max()))


def test_first_adam_step_is_lr_sign_at_full_batch():
    """ADAM's first update is -lr * g/(|g| + eps') for every parameter (m_hat = g, v_hat = g^2, optimizer.py:78-88): a
    size-independent check of the whole bs128 train step (forward, backward, reductions, update) on the device."""
    rt = get_runtime('hip')
    B = 128
    from net.resnet import ResNet, ResNetParams
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, nChan=1, wIn=128, hIn=128, batchSize=B, numJoints=1, nDims=30))
    rng = np.random.RandomState(4)
    x = nets.synthetic_crops(rng, B, 128, 128, np.float32)
    y = rng.normal(0, 0.3, (B, 30)).astype(np.float32)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    w0 = eng.store.w.get().copy()
    lr = 1e-3
    cost = eng.train_step(x, y, lr)
    g = eng.store.g.get()[:eng.store.n_w]
    w1 = eng.store.w.get()
    assert np.isfinite(cost) and np.isfinite(g).all()
    step = (w1 - w0)[:eng.store.n_w].astype('f8')
    g8 = g.astype('f8')
    expect = -lr * g8 / (np.abs(g8) + 1e-8)
    # w1 - w0 is formed from two float32 weights: allow their rounding (|w| <= ~4) next to 0.2 % of lr
    np.testing.assert_allclose(step, expect, rtol=0, atol=2e-3 * lr + 2.0 ** -22 * np.abs(w0[:eng.store.n_w]).max())
    assert (np.abs(g8) > 1e-6).mean() > 0.5 and np.abs(step).max() <= lr * (1 + 1e-3) + 1e-6
