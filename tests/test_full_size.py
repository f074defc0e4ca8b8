"""Full-size parity on the MI355X (BASELINE config 2's geometry: 128x128 crops, the 47-layer ResNet).

* forward joints within 1e-3 mm of the float64 oracle;
* every parameter gradient of the train step against the float64 oracle evaluated on the device's own ReLU / max-pool
  decisions (tests/pinning.py) at 2e-4 of the tensor scale -- at batch 8 AND at the benchmarked batch of 128, whose launch
  plans pick other tiles / split-K factors than the small batches do;
* every distinct GEMM problem (M, N, K, layouts, tile, split-K, prologue / epilogue flags) that the bs128 plans launch, re-run
  stand-alone on random operands against a float64 matmul;
* size-independent properties at the full batch (determinism, batch-composition independence of the deterministic forward,
  the PCA-prior layer as an exact affine map of the embedding, ADAM's first step = -lr * sign(g))."""
import numpy as np
import pytest

from hipdp import engine
from hipdp import runtime as R
from oracle import nets, torch_ref
from tests import gemm_cases
from tests.backends import get_runtime
from tests.pinning import device_masks
from tests.test_engine import MM, bad_gradients, grads_from_store, make_net, zero_gradient_bounds

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


def _gradients_vs_pinned_oracle(B, seed):
    rt = get_runtime('hip')
    net, onet, P = make_net(rt, 0, B, 128, 1, 30, calib_batch=8)
    rng = np.random.RandomState(seed)
    x = nets.synthetic_crops(rng, B, 128, 128, np.float32)
    y = rng.normal(0, 0.3, (B, 30)).astype(np.float32)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    cost, out = eng.cost_and_grads(x, y)
    c_ref, G_ref, out_ref = torch_ref.cost_and_grads(onet, nets.cast_params(P, np.float64), x.astype(np.float64), y.astype(np.float64),
                                                     masks=device_masks(eng, net))
    assert np.abs(out - out_ref).max() * MM < 1e-3
    assert abs(cost - c_ref) < 1e-5 * abs(c_ref)
    G = grads_from_store(eng, net)
    bad = bad_gradients(G, G_ref, zero_tol=zero_gradient_bounds(eng, net, onet, G_ref))       # 2e-4 of each tensor's scale
    gmax = max(np.abs(G_ref[i][s]).max() for i in G_ref for s in range(2))
    assert not bad, [(i, s, 'max |err| %.3g, |ref| max %.3g, largest gradient %.3g' % (np.abs(G[i][s] - G_ref[i][s]).max(), np.abs(G_ref[i][s]).max(), gmax))
                     for i, s in bad[:6]]
    return eng


def test_train_gradients_match_oracle_at_128():
    _gradients_vs_pinned_oracle(8, 6)


def test_train_gradients_match_oracle_at_benchmarked_batch_128():
    """The batch bench.py times.  gemm_plan / wgrad_plan choose tiles and split-K from M = batch x pixels, so this is the only
    place the bs128 instantiations (FC1's 128-row weight-streaming tiles, 256-slice split-K, bm = 128 3x3 tiles) meet an oracle."""
    eng = _gradients_vs_pinned_oracle(128, 16)
    tiles = set((l.keep[0].bm, l.keep[0].bn) for _, l in eng.all_launches() if l.fn is eng.rt.lib.dpp_gemm)
    n_fc = sum(l.fn is eng.rt.lib.dpp_fc_gemm for _, l in eng.all_launches())
    n_fcw = sum(l.fn is eng.rt.lib.dpp_fc_wgrad_stream for _, l in eng.all_launches())
    # FC1 forward / data gradient on the three-stage weight-streaming kernel and its filter gradient on the row stream (defaults), or on
    # dpp_gemm's 128 x 64 tile
    assert (n_fc == 2 and n_fcw == 1) or n_fc == 3 or (128, 64) in tiles


def test_bf16_train_gradients_match_the_bf16_oracle_at_benchmarked_batch_128():
    """`bench.py --dtype bf16` at its own size (bs128, 128x128): every gradient of the bf16 step at the 2e-4 bar against the oracle
    that rounds the same operands to bfloat16 in each pass, on the device's own decisions and rounded operands."""
    from tests.test_engine import bf16_gradients_vs_pinned_oracle
    rt = get_runtime('hip')
    B = 128
    net, onet, P = make_net(rt, 0, B, 128, 1, 30, calib_batch=8)
    rng = np.random.RandomState(17)
    x = nets.synthetic_crops(rng, B, 128, 128, np.float32)
    y = rng.normal(0, 0.3, (B, 30)).astype(np.float32)
    bf16_gradients_vs_pinned_oracle(rt, net, onet, P, x, y)


def test_every_bs128_gemm_instantiation_against_float64():
    """(M, N, K, layouts, tile, split-K, prologue / epilogue flags) exactly as the bs128 forward / backward plans emit them,
    each re-run stand-alone on random operands (tests/gemm_cases.py)."""
    rt = get_runtime('hip')
    from net.resnet import ResNet, ResNetParams
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, nChan=1, wIn=128, hIn=128, batchSize=128, numJoints=1, nDims=30))
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    checked, skipped = gemm_cases.check_all(rt, eng)
    assert len(checked) >= 20, (len(checked), len(skipped))
    tiles = set((k[11], k[12], k[14]) for _, k in checked)
    assert any(t[2] >= 32 for t in tiles)                                                   # deep split-K is among them
    # FC1's three GEMMs: on the three-stage weight-streaming kernel (dpp_fc_gemm, keyed with tile -1) by default, else dpp_gemm's 128 x 64 tile
    fc = [k for _, k in checked if k[11] == -1]
    n_fcw = sum(l.fn is rt.lib.dpp_fc_wgrad_stream for _, l in eng.all_launches())      # (held to the oracle by the gradient tests above
    assert (len(fc) == 2 and n_fcw == 1) or len(fc) == 3 or any(t[:2] == (128, 64) for t in tiles), (fc, tiles)   # and tests/test_gemm.py)


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
    size-independent check of the update on the device's own bs128 gradient (the gradient itself is checked against the
    oracle by test_train_gradients_match_oracle_at_benchmarked_batch_128)."""
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
