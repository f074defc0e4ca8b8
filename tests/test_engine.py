"""End-to-end parity of the compiled ResNet (net/ API -> hipdp.engine -> HIP kernels) against the float64 oracle
on identical seeded weights / inputs: deterministic forward (computeOutput), training forward + cost, every
parameter gradient, and a few full train steps (ADAM + BN running statistics).

Tolerances: the north-star bar is 1e-3 mm on the 3-D joint output, i.e. 1e-3 / 150 = 6.7e-6 in normalised units
(cube_z / 2 = 150 mm for NYU); outputs here are O(0.1-1)."""
import numpy as np
import pytest

from hipdp import heuristics  # noqa: E402
from hipdp import engine, ops
from net.resnet import ResNet, ResNetParams
from oracle import layers as L
from oracle import nets, torch_ref
from tests.backends import BACKENDS, get_runtime
from tests.pinning import device_masks

MM = 150.0          # normalised unit -> mm for a 300 mm NYU cube


def make_net(rt, type_, batch, size, numJoints, nDims, seed=23455, calib_batch=None):
    rng = np.random.RandomState(seed)
    cfg = ResNetParams(type=type_, nChan=1, wIn=size, hIn=size, batchSize=batch, numJoints=numJoints, nDims=nDims)
    net = ResNet(rng, cfgParams=cfg)
    onet = nets.build_resnet(type=type_, wIn=size, hIn=size, batchSize=batch, numJoints=numJoints, nDims=nDims)
    P = nets.init_params(onet, np.random.RandomState(seed), np.float32)
    P = nets.perturb_bn(P, onet, np.random.RandomState(seed + 1))
    # running statistics of a "trained" net: the batch statistics of a calibration batch (random running stats would not
    # normalise anything and the deterministic forward would blow up to 1e6)
    xc = nets.synthetic_crops(np.random.RandomState(seed + 2), calib_batch or batch, size, size, np.float64)   # (a small one keeps full-batch tests quick)
    ones = {i: np.ones(l['out_dim']) for i, l in enumerate(onet['layers']) if l['kind'] == 'dropout'}     # calibration only
    _, cache = nets.forward(onet, nets.cast_params(P, np.float64), xc, True, ones)
    for i, l in enumerate(onet['layers']):
        if l['kind'] == 'bn':
            P[i][2], P[i][3] = cache[i][1].astype(np.float32), cache[i][2].astype(np.float32)
    # last layer scaled so that outputs are O(0.3) like normalised joint coordinates
    last = max(P)
    out_c, _ = nets.forward(onet, nets.cast_params(P, np.float64), xc, True, ones)
    P[last][0] = (P[last][0] * (0.3 / max(1e-6, np.abs(out_c).max()))).astype(np.float32)
    for i, l in enumerate(net.layers):
        if i in P:
            for p, v in zip(l.params + l.params_nontrained, P[i]):
                p.set_value(v)
    return net, onet, P


def bad_gradients(G, G_ref, slots=(0, 1), zero_tol=None):
    """Tensors whose gradient misses the float32 round-off bound (2e-4 of the tensor's scale; conv biases in front of a
    BatchNorm have an exactly-zero gradient, so a floor relative to the largest gradient applies).  zero_tol: {(layer, slot):
    per-channel absolute bound} for tensors whose true gradient is identically zero (see zero_gradient_bounds)."""
    gmax = max(np.abs(G_ref[i][s]).max() for i in G_ref for s in range(2))
    bad = []
    for i in G_ref:
        for s in slots:
            tol = 2e-4 * max(np.abs(G_ref[i][s]).max(), 5e-3 * gmax)
            if zero_tol is not None and (i, s) in zero_tol:
                tol = np.maximum(tol, zero_tol[(i, s)])
            if (np.abs(G[i][s] - G_ref[i][s]) > tol).any():
                bad.append((i, s))
    return bad


def zero_gradient_bounds(eng, net, onet, G_ref):
    """A conv bias in front of a BatchNorm has no gradient: sum_pixels dX = scale * (sum G - n * c1 - c2 * sum xhat) = 0.  In
    float32 (the reference's floatX as much as the kernels') c1 = sum(G) / n is a ROUNDED mean, so the sum comes out as
    ~eps32 * scale * |sum G| (+ the summation noise of n terms) instead of 0 -- at batch 128 that exceeds the generic floor of
    bad_gradients.  Bound per channel: 16 * eps32 * scale * |dbeta|, with dbeta = sum G from the oracle and scale = gamma *
    inv_std from the device's BatchNorm state; the second source is the rounded batch mean inside xhat (sum xhat = n * inv_std *
    (mean - mean_f32) != 0), worth eps32 * scale * |dgamma| * |mean| * inv_std."""
    eps32 = float(np.finfo(np.float32).eps)
    out = {}
    for j, l in enumerate(onet['layers']):
        if l['kind'] == 'bn' and l['src'][0] == 'layer' and onet['layers'][l['src'][1]]['kind'] in ('conv', 'convpool'):
            i = l['src'][1]
            b = eng.bn_states[id(net.layers[j])]
            scale = np.abs(b.scale.get()[:b.C]).astype(np.float64)
            mean, inv_std = np.abs(b.mean.get()[:b.C]).astype(np.float64), np.abs(b.inv_std.get()[:b.C]).astype(np.float64)
            dbeta, dgamma = np.abs(G_ref[j][0]).astype(np.float64), np.abs(G_ref[j][1]).astype(np.float64)
            # sum G - n*c1: c1 rounded; c2 * sum xhat: xhat uses the ROUNDED mean, so sum xhat = n * inv_std * (mean error)
            # (the constant covers the summation noise of the n terms as well: which realisation of it a build gets depends on the
            #  summation order of every kernel downstream -- 64 held for the LDS-tiled stage-1 kernels, the row-streaming ones drew 1.1x that,
            #  round 4's wave-autonomous 1x1 kernels 1.3x of 128 on the stem at batch 128 and 2.2x in the bf16 mode at 256x256, where the
            #  bias gradient is no longer ~0: sum xhat of a bf16-STORED tensor is not 0, and the oracle models that value, 4e-3 there)
            out[(i, 1)] = 320 * eps32 * scale * (dbeta + dgamma * (1.0 + mean * inv_std))
    return out


def gradients_match_on_every_input(run, seeds=(6, 7, 8)):
    """`run(seed)` -> list of bad tensors, evaluated with the oracle's ReLU / max-pool decisions pinned to the device's own
    (tests/pinning.py): on these deliberately tiny nets (16-64 values per BatchNorm channel) one activation within float32
    rounding of the ReLU kink would otherwise flip its mask in one of the two evaluations and move a whole channel's gradient
    by percents.  Pinned, the tight comparison has to hold on EVERY input."""
    failures = [(seed, bad[:4]) for seed, bad in ((seed, run(seed)) for seed in seeds) if bad]
    assert not failures, "gradients off: %r" % (failures,)


def grads_from_store(eng, net):
    G = {}
    for i, l in enumerate(net.layers):
        if l.params:
            G[i] = [eng.store.read_grad(p) for p in l.params]
    return G


@pytest.mark.parametrize('backend', BACKENDS)
def test_resnet_forward_eval_matches_oracle(backend):
    rt = get_runtime(backend)
    net, onet, P = make_net(rt, 1, 4, 32, 14, 3)
    rng = np.random.RandomState(5)
    x = nets.synthetic_crops(rng, 6, 32, 32, np.float32)
    from hipdp import runtime as R
    R.set_default_runtime(rt)
    net.setDeterministic()
    out = net.computeOutput(x)                      # 6 samples, batch 4 -> padded by repeating the last sample
    ref = nets.compute_output(onet, nets.cast_params(P, np.float64), x.astype(np.float64))
    assert out.shape == (6, 42)
    assert net.computeOutput(x[:0]).shape == (0, 42)                   # an empty test set: no batch, an empty result
    assert np.array_equal(net.computeOutput(x[:1]), out[:1])           # a single sample: the batch is that sample repeated
    err_mm = np.abs(out - ref).max() * MM
    assert np.abs(ref).max() > 0.05
    # The bar is 1e-3 mm.  On this deliberately tiny net (batch 4, 2x2 maps: BN statistics from 16 values) float32
    # itself is the limit: the oracle evaluated in float32 (what Theano's floatX=float32 graph computes) is ~0.8e-3 mm
    # away from float64, so the HIP path is required to stay within 3x that float32 noise floor here; the full-size
    # 128x128 / batch-128 parity is measured on the GPU by tests/test_full_size.py.
    noise_mm = np.abs(nets.compute_output(onet, P, x) - ref).max() * MM
    assert err_mm < max(1e-3, 3 * noise_mm), (err_mm, noise_mm)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('type_', [0, 1, 'lazy', 'lazy2', 'lazy3', 'fc1stream', 'early_reduce', 'no_variants', 'conv3stream', 'bn_fuse'])
def test_resnet_train_forward_backward_matches_oracle(backend, type_, monkeypatch):
    """'lazy': type 0 with the opt-in plan in which the 1x1 convolutions form the gradient through a BatchNorm from (G, x) in
    their operand prologue (heuristics.LAZY_BN_BWD) instead of reading a tensor written by bn_bwd_apply.
    'fc1stream': the HiddenLayer behind the last conv map on the weight-streaming kernel (dpp_fc_gemm: forward with split-K,
    data gradient, weight gradient with the BN+ReLU prologue) -- the full-size nets take that path for FC1, this 32x32 net
    (K = 1 024) only with the threshold lowered.
    'early_reduce': the filter / bias gradient partials reduced in several dpp_reduce_multi launches on the gradient branch while the
    pass runs (heuristics.EARLY_REDUCE_BYTES; the bs128 net flushes every 16 MB) instead of one launch at its end.
    'conv3stream': a 64x64 net, whose stage-1 maps are 16 pixels wide, with the opt-in heuristics.CONV3_STREAM_C: its 16 -> 16 3x3 layers
    and their data gradients run on the barrier-free dpp_conv3x3_stream (measured no faster than the tiled kernel: off by default)."""
    size = 32
    if type_ == 'bn_fuse':          # finalize + apply of the BatchNorm backward in one launch (opt-in heuristics.BN_BWD_FUSE_MAX_BLOCKS)
        monkeypatch.setattr(heuristics, 'BN_BWD_FUSE_MAX_BLOCKS', 128)
        type_ = 0
    if type_ == 'conv3stream':
        size, type_ = 64, 0
        monkeypatch.setattr(heuristics, 'CONV3_STREAM_C', (16, 32))
    if type_ == 'lazy2' and backend == 'emu':
        pytest.skip("the opt-in lazy2 plan (measured slower, off by default) is exercised on the GPU tier only; 'lazy' covers the mode-4 operand here")
    if type_ in ('lazy', 'lazy2', 'lazy3'):       # lazy2: only the data gradient does, and leaves the tensor it forms for the filter gradient
        # (lazy3: lazy2 restricted to the data gradients that run on the wave-autonomous kernel, dpp_gemm variant 4)
        monkeypatch.setattr(heuristics, 'LAZY_BN_BWD', {'lazy': 1, 'lazy2': 2, 'lazy3': 3}[type_])
        type_ = 0
    if type_ == 'early_reduce':
        monkeypatch.setattr(heuristics, 'EARLY_REDUCE_BYTES', 1 << 14)
        type_ = 0
    no_variants = type_ == 'no_variants'
    if no_variants:
        # the C side rules the K-split / 16-column stream kernels out for every launch (as misaligned buffers or
        # DPP_GEMM_WIDE_EPILOGUE=0 would): the engine describes those convolutions with the generic tile instead of failing the build
        monkeypatch.setattr(ops, 'gemm_variant_rows', lambda rt, launch: 0)
        type_ = 0
    if type_ == 'fc1stream':
        monkeypatch.setattr(heuristics, 'FC1_MIN_K', 512)
        monkeypatch.setattr(heuristics, 'FC1_STREAM', '1')
        type_ = 0
    rt = get_runtime(backend)
    nJ, nD = (1, 30) if type_ == 0 else (14, 3)
    net, onet, P = make_net(rt, type_, 4, size, nJ, nD, **(dict(calib_batch=4) if size != 32 else {}))
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    nfa = sum(l.fn is rt.lib.dpp_bn_bwd_finalize_apply for _, l in eng.all_launches())
    assert (nfa >= 20) if heuristics.BN_BWD_FUSE_MAX_BLOCKS else (nfa == 0), nfa
    n3s = sum(l.fn is rt.lib.dpp_conv3x3_stream for _, l in eng.all_launches())
    assert n3s == (10 if size == 64 else 0), n3s          # 5 stage-1 blocks: forward + data gradient each
    variants = [l.keep[0].variant for _, l in eng.all_launches() if l.fn is rt.lib.dpp_gemm]
    assert (not any(v in (2, 3, 4) for v in variants)) if no_variants else (any(v in (2, 3) for v in variants) and any(v == 4 for v in variants))
    if heuristics.LAZY_BN_BWD:
        assert any(l.name.startswith('dgrad1x1') and l.keep[0].actA.mode == 4 for l in eng.bwd.launches())
    if heuristics.FC1_MIN_K == 512:
        assert sum(l.fn is rt.lib.dpp_fc_gemm for _, l in eng.all_launches()) >= 4            # forward (+ split-K), data gradient, ...
        assert sum(l.fn is rt.lib.dpp_fc_wgrad_stream for _, l in eng.all_launches()) >= 1    # ... and the filter gradients on the row stream
    if heuristics.EARLY_REDUCE_BYTES == 1 << 14:
        assert sum(l.name == 'reduce_multi_early' for l in eng.bwd.launches()) >= 3
    P64 = nets.cast_params(P, np.float64)

    def run(seed):
        rng = np.random.RandomState(seed)
        x = nets.synthetic_crops(rng, 4, size, size, np.float32)
        y = rng.normal(0, 0.3, (4, nJ * nD)).astype(np.float32)
        cost, out = eng.cost_and_grads(x, y)          # train mode: batch statistics, so the moving running stats do not matter
        c_ref, G_ref, out_ref = torch_ref.cost_and_grads(onet, P64, x.astype(np.float64), y.astype(np.float64), masks=device_masks(eng, net))
        assert np.abs(out - out_ref).max() * MM < 1e-3
        assert abs(cost - c_ref) < 1e-5 * abs(c_ref)
        return bad_gradients(grads_from_store(eng, net), G_ref)

    gradients_match_on_every_input(run)


@pytest.mark.parametrize('backend', BACKENDS)
def test_resnet_train_steps_match_oracle(backend):
    """train_model = forward + backward + ADAM + running-statistics EMA.  ADAM normalises every gradient to a step of
    about +-lr, so weights whose exact gradient is (nearly) zero move by the sign of float32 round-off -- in the
    reference's float32 graph as much as here -- and trajectories of a float32 and a float64 run separate at the 1e-2
    level after a few steps.  Hence: (1) the first cost is compared tightly, (2) the update itself is checked exactly
    by feeding the DEVICE gradients to the oracle's ADAM, (3) the 3-step cost curve is compared loosely."""
    rt = get_runtime(backend)
    net, onet, P = make_net(rt, 0, 4, 32, 1, 30)
    rng = np.random.RandomState(7)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    P64 = nets.cast_params(P, np.float64)
    state = nets.new_adam_state(onet, P64)
    costs, refs = [], []
    for step in range(3):
        x = nets.synthetic_crops(rng, 4, 32, 32, np.float32)
        y = rng.normal(0, 0.3, (4, 30)).astype(np.float32)
        if step == 0:
            w_before = {i: [p.get_value().copy() for p in l.params] for i, l in enumerate(net.layers) if l.params}
        costs.append(eng.train_step(x, y, 1e-3))
        if step == 0:
            # (2) device weights == oracle ADAM applied to the device's own gradients, float32 arithmetic
            G = grads_from_store(eng, net)
            for i, l in enumerate(net.layers):
                for sl, p in enumerate(l.params):
                    w, g = [w_before[i][sl].copy()], [G[i][sl]]
                    m, v = [np.zeros_like(w[0])], [np.zeros_like(w[0])]
                    L.adam_step(w, g, m, v, 1.0, np.float32(1e-3))
                    np.testing.assert_allclose(p.get_value(), w[0], rtol=0, atol=2e-6 * max(1.0, np.abs(w[0]).max()),
                                               err_msg='layer %d slot %d' % (i, sl))
        c, _ = nets.train_step(onet, P64, state, x.astype(np.float64), y.astype(np.float64), np.float32(1e-3))
        refs.append(c)
        if step == 0:
            # running BN statistics follow the reference's EMA of mean and inv_std (identical weights at step 0)
            for i, l in enumerate(net.layers):
                if l.__class__.__name__ == 'BatchNormLayer':
                    np.testing.assert_allclose(l.mean.get_value(), P64[i][2], rtol=1e-5, atol=1e-5)
                    np.testing.assert_allclose(l.inv_std.get_value(), P64[i][3], rtol=2e-5)
    assert abs(costs[0] - refs[0]) < 1e-5 * refs[0]
    np.testing.assert_allclose(costs, refs, rtol=3e-2)
    # deterministic forward after training, through the re-used parameter store: compare with the oracle evaluated on
    # the DEVICE's trained weights
    xt = nets.synthetic_crops(rng, 4, 32, 32, np.float32)
    from hipdp import runtime as R
    R.set_default_runtime(rt)
    net.setDeterministic()
    out = net.computeOutput(xt)
    Pd = {i: [np.asarray(p.get_value(), np.float64) for p in l.params + l.params_nontrained] for i, l in enumerate(net.layers)
          if l.params}
    ref = nets.compute_output(onet, Pd, xt.astype(np.float64))
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('backend', BACKENDS)
def test_resnet_type3_dropout_narrow_stages(backend):
    """ResNet type 3 (resnet.py:243-288): stages 3-4 narrowed to 128 filters -- stage 3 becomes identity blocks without
    down-sampling -- and a DropoutLayer behind each 1024-wide layer; gradients against the oracle with the device's masks.
    The oracle evaluates with the device's dropout masks AND the device's ReLU decisions (tests/pinning.py), so the tight
    comparison holds on every input."""
    rt = get_runtime(backend)
    net, onet, P = make_net(rt, 3, 4, 32, 1, 30)
    assert [l.__class__.__name__ for l in net.layers].count('DropoutLayer') == 2
    from hipdp import runtime as R
    R.set_default_runtime(rt)
    x0 = nets.synthetic_crops(np.random.RandomState(7), 4, 32, 32, np.float32)
    # deterministic mode first (the train-mode passes below move the running statistics): dropout is the 0.7 scale
    # (dropoutlayer.py:100-104)
    net.setDeterministic()
    ref = nets.compute_output(onet, nets.cast_params(P, np.float64), x0.astype(np.float64))
    noise = np.abs(nets.compute_output(onet, P, x0) - ref).max() * MM
    assert np.abs(net.computeOutput(x0) - ref).max() * MM < max(1e-3, 3 * noise)
    net.unsetDeterministic()
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    for seed in (7, 8, 9):
        rng = np.random.RandomState(seed)
        x = nets.synthetic_crops(rng, 4, 32, 32, np.float32)
        y = rng.normal(0, 0.3, (4, 30)).astype(np.float32)
        cost, out = eng.cost_and_grads(x, y)
        masks = {i: eng.dropout_masks[id(l)][0].get().astype(np.float64) for i, l in enumerate(net.layers) if id(l) in eng.dropout_masks}
        assert len(masks) == 2
        pin = {k: v for k, v in device_masks(eng, net).items() if onet['layers'][k]['kind'] != 'convpool'}
        Pn = {i: [p.get_value() for p in l.params + l.params_nontrained] for i, l in enumerate(net.layers) if l.params}   # running stats moved
        c_ref, G_ref, _, out_ref = nets.cost_and_grads(onet, nets.cast_params(Pn, np.float64), x.astype(np.float64), y.astype(np.float64), True, masks,
                                                       masks=pin)
        assert np.abs(out - out_ref).max() * MM < 1e-3
        assert abs(cost - c_ref) < 1e-5 * abs(c_ref)
        bad = bad_gradients(grads_from_store(eng, net), G_ref)
        assert not bad, (seed, bad[:4])


@pytest.mark.parametrize('backend', BACKENDS)
def test_weight_decay_cost_and_gradients(backend):
    """weightreg_factor (poseregnettrainer.py:101-107): cost += wd * sum(W^2) over conv / FC weights for nets without dropout,
    and 2 * wd * W in their gradients; a net with dropout ignores it."""
    rt = get_runtime(backend)
    wd = 1e-3
    net, onet, P = make_net(rt, 0, 4, 32, 1, 30)
    rng = np.random.RandomState(12)
    x = nets.synthetic_crops(rng, 4, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (4, 30)).astype(np.float32)
    P64 = nets.cast_params(P, np.float64)
    # validation cost first (the training pass below moves the running statistics)
    ev = engine.CompiledNet(net, train=False, runtime=rt, loss=dict(kind='embedding'), weight_decay=wd)
    c_ev, _ = ev.evaluate(x, y)
    o_ev, _ = nets.forward(onet, P64, x.astype(np.float64), False)
    reg = sum(wd * (P64[i][0] ** 2).sum() for i, l in enumerate(onet['layers']) if l['kind'] in ('conv', 'convpool', 'fc'))
    assert abs(c_ev - (((o_ev - y) ** 2).sum(axis=1).mean() + reg)) < 1e-5 * c_ev
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), weight_decay=wd)
    # ONE launch each for the regulariser and its gradient (all weights as segments of the flat buffer): the per-layer version -- 67
    # launches for the ResNet, a single-workgroup sum over FC1's 16.8 M weights among them -- made the default trainer
    # (weightreg_factor = 0.001, nettrainer.py:52) six times slower than a step without (tools/trainer_throughput.py)
    assert [l.name for l in eng.lossplan.launches()].count('sumsq_multi') == 1 and not any(l.name == 'sumsq' for l in eng.lossplan.launches())
    assert sum(l.name == 'axpy_multi' for l in eng.bwd.launches()) == 1

    def run(seed):
        r = np.random.RandomState(seed)
        xs = nets.synthetic_crops(r, 4, 32, 32, np.float32)
        ys = r.normal(0, 0.3, (4, 30)).astype(np.float32)
        cost, _ = eng.cost_and_grads(xs, ys)
        pin = {k: v for k, v in device_masks(eng, net).items() if onet['layers'][k]['kind'] != 'convpool'}      # (no exact pool ties on random inputs)
        c_ref, G_ref, _, _ = nets.cost_and_grads(onet, P64, xs.astype(np.float64), ys.astype(np.float64), weight_decay=wd, masks=pin)
        c_plain, _, _, _ = nets.cost_and_grads(onet, P64, xs.astype(np.float64), ys.astype(np.float64), masks=pin)
        assert c_ref - c_plain > 1e-3 * c_plain                    # the regulariser is visible in the cost
        assert abs(cost - c_ref) < 1e-5 * abs(c_ref)
        return bad_gradients(grads_from_store(eng, net), G_ref, slots=(0,))

    gradients_match_on_every_input(run, seeds=(12, 13, 14))
    # dropout nets ignore the factor
    net3, onet3, P3 = make_net(rt, 3, 4, 32, 1, 30)
    e3 = engine.CompiledNet(net3, train=False, runtime=rt, loss=dict(kind='embedding'), weight_decay=wd)
    c3, _ = e3.evaluate(x, y)
    o3, _ = nets.forward(onet3, nets.cast_params(P3, np.float64), x.astype(np.float64), False)
    assert abs(c3 - ((o3 - y) ** 2).sum(axis=1).mean()) < 1e-5 * c3


def bf16_gradients_vs_pinned_oracle(rt, net, onet, P, x, y, bar=2e-4):
    """One training forward + backward of CompiledNet(bf16=True) against the float64 oracle that rounds THE SAME operands to
    bfloat16 in each of the three passes (oracle/torch_ref.py:_QBilinear, configured from the net's own launches by
    tests/pinning.py:device_quant) on the device's own ReLU / pooling decisions and rounded forward operands: forward output at the
    1e-3 mm bar, cost at 1e-5, every parameter gradient at `bar` of its tensor's scale -- the bars of the fp32 path."""
    from tests.pinning import device_grad_pins, device_quant, device_store, store_agreement
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), bf16=True)
    cost, out = eng.cost_and_grads(x, y)
    quant = device_quant(eng, net)
    assert any(q['fwd'] for q in quant.values()) and any(q['dgrad'] for q in quant.values())
    # bf16 STORAGE (the default of the bf16 mode): the device's own stored tensors are pinned like its rounded MFMA operands ...
    store = device_store(eng, net) if eng.store16 else None
    assert (store is not None and len(store) >= 10) == bool(heuristics.BF16_STORE)
    own, own_g = {}, {}
    gpins = device_grad_pins(eng, net) if eng.grad16 else None
    c_ref, G_ref, out_ref = torch_ref.cost_and_grads(onet, nets.cast_params(P, np.float64), x.astype(np.float64), y.astype(np.float64),
                                                     masks=device_masks(eng, net), quant=quant, store=store, stored_out=own, grad_pins=gpins,
                                                     stored_grads=own_g)
    if gpins is not None:
        # the same un-pinned check for the bf16-stored gradient tensors (where the net is big enough to have any: maps of > 256 rows)
        # (the device forms the BatchNorm-backward sums from the values AS STORED, so the oracle on the pinned tensors reproduces c1 / c2;
        # what is left are elements within float32 round-off of a bfloat16 rounding boundary: the neighbouring value, or -- for an
        # element that is the result of a cancellation -- float32 noise at the tensor's scale)
        for (kind, i), mine in own_g.items():
            pin = gpins[0][i] if kind == 'g' else gpins[1][i]
            mine = np.asarray(mine, np.float32)
            # (as tests/pinning.py:store_agreement: an element that is noise against its tensor -- a channel whose gradients are all masked
            # has c1 = c2 = rounding noise, its dX values are 1e-12 on a tensor of 1e-2 and their bfloat16 neighbours are arbitrary --
            # counts as agreeing; round 6 met one on the MI355X once the data gradients ran on bf16 MFMA, tools/bf16_agreement.py)
            same = (mine == pin) | (np.abs(mine - pin) <= np.float32(2e-6) * np.abs(pin).max())
            bound = np.maximum(np.abs(pin) * np.float32(2.0 ** -7), np.float32(2e-5) * np.abs(pin).max())
            assert same.mean() >= 0.995 and (np.abs(mine - pin)[~same] <= bound[~same]).all(), (kind, i, float(same.mean()))
    if store is not None:
        # ... and the pins themselves are checked WITHOUT pinning: the float64 oracle's own rounding of what it would have stored equals
        # the device's stored tensor everywhere except where the value sits within float32 round-off of a bfloat16 rounding boundary
        agree = store_agreement(store, own)
        assert min(agree.values()) >= 0.998 and np.mean(list(agree.values())) >= 0.9995, sorted(agree.items(), key=lambda kv: kv[1])[:4]
    assert np.abs(out - out_ref).max() * MM < 1e-3, np.abs(out - out_ref).max() * MM
    assert abs(cost - c_ref) < 1e-5 * abs(c_ref)
    G = grads_from_store(eng, net)
    gmax = max(np.abs(G_ref[i][s]).max() for i in G_ref for s in range(2))
    ztol = zero_gradient_bounds(eng, net, onet, G_ref)
    bad = []
    for i in G_ref:
        for s in range(2):
            tol = bar * max(np.abs(G_ref[i][s]).max(), 5e-3 * gmax)
            if (i, s) in ztol:
                tol = np.maximum(tol, ztol[(i, s)])
            if (np.abs(G[i][s] - G_ref[i][s]) > tol).any():
                bad.append((i, s, float(np.abs(G[i][s] - G_ref[i][s]).max() / max(np.abs(G_ref[i][s]).max(), 5e-3 * gmax))))
    assert not bad, bad[:8]
    return eng, quant, (cost, out, G), (c_ref, out_ref, G_ref)


@pytest.mark.parametrize('backend', BACKENDS)
def test_bf16_step_matches_the_bf16_oracle(backend, monkeypatch):
    """CompiledNet(bf16=True) (BASELINE config 5) on a small net: every gradient at the 2e-4 bar against the oracle that rounds the
    same operands; the bf16 kernels really ran (the result differs from the fp32 engine's), and the UNROUNDED oracle is far outside
    the bar (the comparison has teeth)."""
    monkeypatch.setattr(heuristics, 'FC1_MIN_K', 512)
    rt = get_runtime(backend)
    net, onet, P = make_net(rt, 0, 4, 32, 1, 30)
    rng = np.random.RandomState(6)
    x = nets.synthetic_crops(rng, 4, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (4, 30)).astype(np.float32)
    eng, quant, (c16, o16, G16), (c_ref, o_ref, G_ref) = bf16_gradients_vs_pinned_oracle(rt, net, onet, P, x, y)
    assert any(l.fn is rt.lib.dpp_conv3x3_bf16 for _, l in eng.all_launches())
    assert any(l.fn is rt.lib.dpp_fc_gemm and l.args[1] == 1 for _, l in eng.all_launches())
    e32 = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), bf16=False)
    c32, o32 = e32.cost_and_grads(x, y)
    assert np.abs(o16 - o32).max() > 100 * np.abs(o16 - o_ref).max()          # bf16 is not fp32 ...
    _, G_plain, _ = torch_ref.cost_and_grads(onet, nets.cast_params(P, np.float64), x.astype(np.float64), y.astype(np.float64),
                                             masks=device_masks(eng, net))
    far = max(np.abs(G16[i][0] - G_plain[i][0]).max() / np.abs(G_plain[i][0]).max() for i in G_plain if onet['layers'][i]['kind'] == 'conv')
    assert far > 20 * 2e-4, far                                               # ... and the plain oracle would not pass the bar


@pytest.mark.parametrize('backend', BACKENDS)
def test_bf16_gradient_storage_on_maps_large_enough_to_use_it(backend, monkeypatch):
    """The bf16 mode also stores the GRADIENTS of the activation tensors as bfloat16 -- for maps of more than 256 rows (engine.
    _grad_dtype).  A 64x64 net at batch 4 has such maps in stages 1 and 2 (1 024 and 256+ rows): the masked gradients G and the
    BatchNorm-backward outputs dX there really are bf16 on the device, every parameter gradient still meets the 2e-4 bar against the
    oracle on the device's own stored operands, the oracle's own roundings agree with the device's stored gradients, and with
    BF16_GRADS off the gradients come out float32 and different."""
    from tests.pinning import device_grad_pins
    monkeypatch.setattr(heuristics, 'FC1_MIN_K', 512)
    rt = get_runtime(backend)
    net, onet, P = make_net(rt, 0, 4, 64, 1, 30, calib_batch=4)
    rng = np.random.RandomState(16)
    x = nets.synthetic_crops(rng, 4, 64, 64, np.float32)
    y = rng.normal(0, 0.3, (4, 30)).astype(np.float32)
    eng, quant, (c16, o16, G16), _ = bf16_gradients_vs_pinned_oracle(rt, net, onet, P, x, y)
    assert eng.grad16
    gp, dv = device_grad_pins(eng, net)
    assert len(gp) >= 6 and len(dv) >= 6, (len(gp), len(dv))
    monkeypatch.setattr(heuristics, 'BF16_GRADS', False)
    e2 = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), bf16=True)
    assert e2.store16 and not e2.grad16
    e2.cost_and_grads(x, y)
    assert device_grad_pins(e2, net) == ({}, {})
    G2 = grads_from_store(e2, net)
    assert any(not np.array_equal(G16[i][0], G2[i][0]) for i in G16)


@pytest.mark.parametrize('backend', BACKENDS)
def test_early_fc1_adam_step_equals_plain_step(backend, monkeypatch):
    """step_plan updates the FC1 weight inside the backward pass (gradient branch, after the two kernels that use it, weight decay
    first) and the rest of the flat buffer at the end: two whole steps must leave EVERY parameter, both ADAM moments and the cost
    bit-identical to backward-then-one-ADAM."""
    rt = get_runtime(backend)
    monkeypatch.setattr(heuristics, 'EARLY_BUCKET_MIN', 1 << 18)           # the 32x32 test net's FC1 (1 M weights) qualifies
    wd = 1e-3
    state = {}
    for early in (True, False):
        monkeypatch.setattr(heuristics, 'EARLY_ADAM', early)
        net, _, _ = make_net(rt, 0, 4, 32, 1, 30)
        eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'), weight_decay=wd)
        assert eng._early_adam is not None
        bwd, upd = eng._early_adam_plans()
        assert (bwd is not eng.bwd) == early
        if early:
            assert len(bwd) == len(eng.bwd) + 1 and [o.name for o in upd.launches()].count('adam') == 2
            names = [o.name for o in bwd.launches()]
            assert names.index('adam_fc1') < len(names) // 4                      # early in the pass, not in its tail
        costs = []
        for seed in (5, 6):
            r = np.random.RandomState(seed)
            costs.append(eng.train_step(nets.synthetic_crops(r, 4, 32, 32, np.float32), r.normal(0, 0.3, (4, 30)).astype(np.float32), 1e-3))
        state[early] = (costs, eng.store.w.get(), eng.store.m.get(), eng.store.v.get(), eng.hyper.get())
    for a, b in zip(state[True], state[False]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize('backend', BACKENDS)
def test_projection_shortcuts_beside_the_chain_and_in_it(backend, monkeypatch):
    """The projection shortcut of a residual block runs on the second stream beside the bottleneck (the bottleneck's exit convolution
    absorbs the add); a step whose second stream is busy during the forward pass (step_plan(early=): the cascade of bench.py) keeps the
    shortcuts in the chain; with DPP_SIDE_SHORTCUT=0 the later convolution absorbs the add as in rounds 1-3.  Two steps leave
    bit-identical parameters in the first two cases (the same kernels in another order) and the same to round-off in the third."""
    from hipdp.ops import Fork, Join, Plan
    rt = get_runtime(backend)
    state = {}
    for mode in ('beside', 'early', 'off'):
        monkeypatch.setattr(heuristics, 'SIDE_SHORTCUT', mode != 'off')
        net, _, _ = make_net(rt, 0, 4, 32, 1, 30)
        eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
        side_convs = [o.name for (o, side) in eng.fwd.ops if side and getattr(o, 'name', '').startswith('conv1x1')]
        assert len(side_convs) == (0 if mode == 'off' else 3) and len(eng._beside) == len(side_convs)
        plan = eng.step_plan()
        if mode == 'early':
            scratch = rt.alloc(64)
            other = Plan('other work')
            other.add(ops.fill_zero(rt, scratch))
            plan = eng.step_plan(early=other)
            fwd_side = [getattr(o, 'name', '') for (o, side) in plan.ops if side]
            assert not any(n.startswith('conv1x1') for n in fwd_side) and 'fill_zero' in fwd_side
            nfj = lambda pl: sum(isinstance(o, (Fork, Join)) for o, _ in pl.ops)            # noqa: E731
            assert nfj(plan) == nfj(eng.step_plan()) - 6 + 1                                # three fork / join pairs gone, one fork for `other`
        for seed in (5, 6):
            r = np.random.RandomState(seed)
            eng.set_input(nets.synthetic_crops(r, 4, 32, 32, np.float32))
            eng.y_in.set(r.normal(0, 0.3, (4, 30)).astype(np.float32))
            eng.set_lr(1e-3)
            plan.run(rt)
        rt.synchronize()
        state[mode] = (eng.store.w.get().copy(), float(eng.cost.get()[0]))
    assert np.array_equal(state['beside'][0], state['early'][0]) and state['beside'][1] == state['early'][1]
    # (ADAM normalises every gradient to a step of ~lr: a parameter whose gradient is ~0 may step the other way in either pass)
    d = np.abs(state['beside'][0] - state['off'][0])
    assert d.max() <= 2 * 2 * 1e-3 * 1.01 and np.mean(d > 1e-4) < 0.02, (d.max(), np.mean(d > 1e-4))
    assert abs(state['beside'][1] - state['off'][1]) < 1e-4 * abs(state['off'][1])       # (the second step's cost, on weights one ADAM step apart)


@pytest.mark.parametrize('backend', BACKENDS)
def test_engines_of_a_rebuilt_parameter_store_refuse_to_run(backend):
    """Appending the PCA-prior layer to a trained net (what the mains do, /root/reference/src/main_nyu_posereg_embedding.py:175-190)
    rebuilds the device parameter store: the values survive, the net evaluates through the new layer, and an engine compiled before
    the change raises instead of computing with buffers nobody updates any more."""
    from hipdp import runtime as R
    from net.hiddenlayer import HiddenLayer, HiddenLayerParams
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    net, onet, P = make_net(rt, 0, 4, 32, 1, 30)
    x = nets.synthetic_crops(np.random.RandomState(5), 4, 32, 32, np.float32)
    net.setDeterministic()
    emb = net.computeOutput(x)
    old = engine.CompiledNet(net, train=False, runtime=rt)
    assert np.array_equal(old.forward(x), emb)
    rng = np.random.RandomState(3)
    comp, mean = rng.normal(0, 1, (30, 42)).astype(np.float32), rng.normal(0, 1, 42).astype(np.float32)
    prior = HiddenLayer(rng, net.layers[-1].output, HiddenLayerParams(inputDim=(4, 30), outputDim=(4, 42), activation=None), layerNum=len(net.layers))
    prior.W.set_value(comp)
    prior.b.set_value(mean)
    net.layers.append(prior)
    net.output = prior.output
    net.cfgParams.numJoints, net.cfgParams.nDims, net.cfgParams.outputDim = 14, 3, (4, 42)
    joints = net.computeOutput(x)
    np.testing.assert_allclose(joints, emb.astype(np.float64) @ comp + mean, rtol=0, atol=1e-4)
    with pytest.raises(RuntimeError, match='compile the net again'):
        old.forward(x)


@pytest.mark.parametrize('backend', BACKENDS)
def test_async_checkpoint_writes_the_bytes_save_writes(backend, tmp_path):
    """NetBase.saveAsync (the epoch loop's snapshot off the training thread): the file is byte for byte what save()
    writes for the parameters AT THE CALL -- training goes on while the worker thread pickles."""
    from hipdp import runtime as R
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    net, onet, P = make_net(rt, 0, 4, 32, 1, 30)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    rng = np.random.RandomState(7)
    x = nets.synthetic_crops(rng, 4, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (4, 30)).astype(np.float32)
    eng.train_step(x, y, 1e-2)
    net.save(str(tmp_path / 'sync.pkl'))
    net.saveAsync(str(tmp_path / 'async.pkl'))
    eng.train_step(x, y, 1e-2)                     # the live parameters move on while the snapshot is written
    net.saveAsync(str(tmp_path / 'async2.pkl'))    # (joins the first one)
    net.joinSave()
    a, b, c = (open(str(tmp_path / f), 'rb').read() for f in ('sync.pkl', 'async.pkl', 'async2.pkl'))
    assert a == b and len(a) > 1000 and c != a
    net.save(str(tmp_path / 'sync2.pkl'))
    assert open(str(tmp_path / 'sync2.pkl'), 'rb').read() == c


@pytest.mark.parametrize('backend', BACKENDS)
def test_async_checkpoint_survives_a_full_scratch_directory_and_a_dead_writer(backend, tmp_path, monkeypatch, capsys):
    """a failure of the asynchronous hand-over (Docker's 64 MB /dev/shm, a writer process that dies) must not end a
    run that save() would have survived: the snapshot falls back to the synchronous writer (same bytes), leaves no scratch files behind,
    and joinSave() does not raise.  A snapshot dropped by skip_if_busy is written once the running writer finishes."""
    import subprocess
    import threading
    from hipdp import runtime as R
    from net.netbase import NetBase
    rt = get_runtime(backend)
    if not hasattr(rt, 'download_async'):
        pytest.skip('runtime without asynchronous downloads: saveAsync is save()')
    R.set_default_runtime(rt)
    net, onet, P = make_net(rt, 0, 4, 32, 1, 30)
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    rng = np.random.RandomState(7)
    x = nets.synthetic_crops(rng, 4, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (4, 30)).astype(np.float32)
    eng.train_step(x, y, 1e-2)
    net.save(str(tmp_path / 'sync.pkl'))
    want = open(str(tmp_path / 'sync.pkl'), 'rb').read()
    # (1) no scratch directory has room
    monkeypatch.setattr(NetBase, '_scratch_dir', staticmethod(lambda need: None))
    net.saveAsync(str(tmp_path / 'a1.pkl'))
    net.joinSave(strict=True)
    assert open(str(tmp_path / 'a1.pkl'), 'rb').read() == want
    # (2) the writer process fails; scratch files are cleaned up
    scratch = tmp_path / 'scratch'
    scratch.mkdir()
    monkeypatch.setattr(NetBase, '_scratch_dir', staticmethod(lambda need: str(scratch)))
    real_run = subprocess.run

    def dead(*a, **k):
        return subprocess.CompletedProcess(a, 1, stdout=b'killed', stderr=b'')
    monkeypatch.setattr(subprocess, 'run', dead)
    net.saveAsync(str(tmp_path / 'a2.pkl'))
    net.joinSave(strict=True)
    assert open(str(tmp_path / 'a2.pkl'), 'rb').read() == want and list(scratch.iterdir()) == []
    assert 'writing it synchronously' in capsys.readouterr().out
    # (3) both ways fail: reported, not raised (unless strict)
    net.saveAsync(str(tmp_path / 'no_such_dir' / 'a3.pkl'))
    net.joinSave()
    assert 'was NOT written' in capsys.readouterr().out
    net.saveAsync(str(tmp_path / 'no_such_dir' / 'a3.pkl'))
    with pytest.raises(OSError):
        net.joinSave(strict=True)
    # (4) a request dropped while the writer is busy is written by joinSave()
    gate = threading.Event()

    def slow(*a, **k):
        gate.wait(30)
        return real_run(*a, **k)
    monkeypatch.setattr(subprocess, 'run', slow)
    assert net.saveAsync(str(tmp_path / 'b1.pkl')) is True
    assert net.saveAsync(str(tmp_path / 'last.pkl'), skip_if_busy=True) is False
    gate.set()
    net.joinSave(strict=True)
    assert open(str(tmp_path / 'b1.pkl'), 'rb').read() == want and open(str(tmp_path / 'last.pkl'), 'rb').read() == want


@pytest.mark.parametrize('backend', BACKENDS)
def test_device_weight_snapshot_and_bulk_checkpoint_values(backend, tmp_path):
    """The epoch loop's "best weights so far" stay on the device (NetBase.deviceWeightSnapshot / restoreDeviceWeightSnapshot: two
    device-to-device copies instead of the reference's host copy of every array, nettrainer.py:871-876), and a checkpoint reads the
    whole parameter store with two device -> host copies: both must give exactly what the per-parameter path gives."""
    import pickle
    from hipdp import runtime as R
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    net, onet, P = make_net(rt, 0, 4, 32, 1, 30)
    assert net.deviceWeightSnapshot() is None                      # no device store before the first compile
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    rng = np.random.RandomState(7)
    x = nets.synthetic_crops(rng, 4, 32, 32, np.float32)
    y = rng.normal(0, 0.3, (4, 30)).astype(np.float32)
    eng.train_step(x, y, 1e-2)
    before = [[p.get_value() for p in l.params + l.params_nontrained] for l in net.layers]
    snap = net.deviceWeightSnapshot()
    net.save(str(tmp_path / 'a.pkl'))
    saved = pickle.load(open(str(tmp_path / 'a.pkl'), 'rb'))
    for l, vals in zip(net.layers, before):
        if vals:
            got = saved['%d-values' % l.layerNum]
            assert len(got) == len(vals) and all(np.array_equal(a, b) and a.shape == b.shape and a.dtype == b.dtype for a, b in zip(got, vals))
    eng.train_step(x, y, 1e-2)
    moved = [[p.get_value() for p in l.params + l.params_nontrained] for l in net.layers]
    assert any(not np.array_equal(a, b) for va, vb in zip(before, moved) for a, b in zip(va, vb))
    snap2 = net.deviceWeightSnapshot(snap)                          # reuses the buffers of the earlier snapshot
    assert snap2[1][0].ptr == snap[1][0].ptr
    eng.train_step(x, y, 1e-2)
    final_nt = [[p.get_value() for p in l.params_nontrained] for l in net.layers]
    net.restoreDeviceWeightSnapshot(snap2)
    # the reference's early stopping restores `weightVals` = the TRAINED parameters (nettrainer.py:871-876, 893-895); the BatchNorm
    # running statistics keep their final values -- on the device path exactly as on the host path
    after = [[p.get_value() for p in l.params] for l in net.layers]
    want = [vals[:len(l.params)] for l, vals in zip(net.layers, moved)]
    assert all(np.array_equal(a, b) for va, vb in zip(want, after) for a, b in zip(va, vb))
    after_nt = [[p.get_value() for p in l.params_nontrained] for l in net.layers]
    assert all(np.array_equal(a, b) for va, vb in zip(final_nt, after_nt) for a, b in zip(va, vb))
    assert any(not np.array_equal(a, b) for l, va, vb in zip(net.layers, moved, after_nt) for a, b in zip(va[len(l.params):], vb))
