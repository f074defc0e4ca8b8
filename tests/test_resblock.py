"""The fused deterministic-mode bottleneck block (csrc/resblock.hip, dpp_resblock_eval) against the float64 oracle layers composed
the way res_block composes them (/root/reference/src/net/resnet.py:349-414 with BatchNorm in deterministic mode,
/root/reference/src/net/batchnormlayer.py:158-159).  Emulator in the CPU tier, the real kernel through the C ABI with `-m gpu`."""
import numpy as np
import pytest

from hipdp import layout, ops
from oracle import layers as L
from tests.backends import BACKENDS, get_runtime


def _bn_params(rng, C):
    return dict(mean=rng.normal(size=C) * 0.3, inv_std=rng.uniform(0.5, 2.0, C), gamma=rng.uniform(0.5, 1.5, C), beta=rng.normal(size=C) * 0.3)


def _bnrelu(x, p):
    s = lambda v: v[None, :, None, None]          # noqa: E731
    return np.maximum((x - s(p['mean'])) * s(p['gamma'] * p['inv_std']) + s(p['beta']), 0)


def _block_ref(x, P, stride, proj):
    h = _bnrelu(x, P['bn0'])
    c1 = L.conv2d_fwd(h, P['W1'], P['b1'], (stride, stride), 'half')
    c2 = L.conv2d_fwd(_bnrelu(c1, P['bn1']), P['W2'], P['b2'], (1, 1), 'half')
    c3 = L.conv2d_fwd(_bnrelu(c2, P['bn2']), P['W3'], P['b3'], (1, 1), 'half')
    if not proj:
        return x + c3
    return c3 + L.conv2d_fwd(h, P['Wsc'], P['bsc'], (stride, stride), 'half')


def _run(rt, x, P, stride, proj, Nb, Cout):
    N, Cin, H, W = x.shape
    f = lambda a: rt.upload(np.ascontiguousarray(a, np.float32))          # noqa: E731
    bns = []
    for k in ('bn0', 'bn1', 'bn2'):
        bns.append(ops.bn_eval(*(f(P[k][q]) for q in ('mean', 'inv_std', 'gamma', 'beta'))))
    X = f(layout.nchw_to_nhwc(x))
    Ho, Wo = -(-H // stride), -(-W // stride)
    Y = rt.alloc((N, Ho, Wo, Cout), zero=False)
    kw = {}
    if proj:
        kw = dict(Wsc=f(layout.conv_w_to_kernel(P['Wsc'])), bsc=f(P['bsc']))
    op = ops.resblock_eval(rt, X, N, H, W, Cin, stride, Cout, Nb, bns[0], bns[1], bns[2], f(layout.conv_w_to_kernel(P['W1'])), f(P['b1']),
                           f(layout.conv_w_to_kernel(P['W2'])), f(P['b2']), f(layout.conv_w_to_kernel(P['W3'])), f(P['b3']), Y, **kw)
    op(rt.stream)
    rt.synchronize()
    return layout.nhwc_to_nchw(Y.get())


def _params(rng, Cin, Nb, Cout, proj):
    P = dict(bn0=_bn_params(rng, Cin), bn1=_bn_params(rng, Nb), bn2=_bn_params(rng, Nb),
             W1=rng.normal(size=(Nb, Cin, 1, 1)) * (2.0 / Cin) ** 0.5, b1=rng.normal(size=Nb) * 0.1,
             W2=rng.normal(size=(Nb, Nb, 3, 3)) * (2.0 / (9 * Nb)) ** 0.5, b2=rng.normal(size=Nb) * 0.1,
             W3=rng.normal(size=(Cout, Nb, 1, 1)) * (2.0 / Nb) ** 0.5, b3=rng.normal(size=Cout) * 0.1)
    if proj:
        P['Wsc'] = rng.normal(size=(Cout, Cin, 1, 1)) * (2.0 / Cin) ** 0.5
        P['bsc'] = rng.normal(size=Cout) * 0.1
    return P


# (N, H, W, Nb): identity blocks of the three stage shapes -- whole tiles, ragged tiles (H, W no multiple of the tile), maps smaller
# than one tile, several tiles per image in both directions
IDENTITY = [(2, 8, 16, 16), (1, 19, 21, 16), (3, 4, 4, 16), (2, 8, 8, 32), (1, 11, 13, 32), (2, 2, 2, 32), (3, 8, 8, 64), (1, 9, 10, 64),
            (2, 2, 2, 64), (1, 16, 16, 64)]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', IDENTITY, ids=lambda c: 'x'.join(str(v) for v in c))
def test_identity_block_matches_the_oracle_layers(backend, cfg):
    rt = get_runtime(backend)
    N, H, W, Nb = cfg
    C4 = 4 * Nb
    assert rt.lib.dpp_resblock_eval_ok(C4, C4, Nb, 1, 0) == 1
    rng = np.random.RandomState(3 + H * W + Nb)
    x = rng.normal(size=(N, C4, H, W))
    P = _params(rng, C4, Nb, C4, False)
    ref = _block_ref(x, P, 1, False)
    out = _run(rt, x, P, 1, False, Nb, C4)
    # three chained f32 products of depth Cin, 9 Nb, Nb: round-off bound relative to the output scale
    tol = 4e-6 * np.sqrt(C4 + 9 * Nb + Nb) * np.abs(ref).max()
    np.testing.assert_allclose(out, ref, rtol=0, atol=tol)


# (N, H, W, Cin, Nb, stride): the three projection blocks of the ResNet (32 -> 64, 64 -> 128, 128 -> 256, stride 2), odd map sizes (the
# strided output is ceil(H / 2)), a stride-1 projection
PROJECTION = [(2, 16, 16, 32, 16, 2), (1, 13, 19, 32, 16, 2), (2, 16, 16, 64, 32, 2), (1, 9, 7, 64, 32, 2), (2, 16, 16, 128, 64, 2),
              (3, 5, 6, 128, 64, 2), (1, 8, 8, 64, 32, 1)]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', PROJECTION, ids=lambda c: 'x'.join(str(v) for v in c))
def test_projection_block_matches_the_oracle_layers(backend, cfg):
    rt = get_runtime(backend)
    N, H, W, Cin, Nb, stride = cfg
    Cout = 4 * Nb
    assert rt.lib.dpp_resblock_eval_ok(Cin, Cout, Nb, stride, 1) == 1
    rng = np.random.RandomState(5 + H * W + Nb)
    x = rng.normal(size=(N, Cin, H, W))
    P = _params(rng, Cin, Nb, Cout, True)
    ref = _block_ref(x, P, stride, True)
    out = _run(rt, x, P, stride, True, Nb, Cout)
    tol = 4e-6 * np.sqrt(2 * Cin + 9 * Nb + Nb) * np.abs(ref).max()
    np.testing.assert_allclose(out, ref, rtol=0, atol=tol)


@pytest.mark.parametrize('backend', BACKENDS)
def test_a_pixel_does_not_depend_on_its_batch(backend):
    """Same frame alone and as the last of a batch of 5: bit-identical rows (tile geometry and summation order follow from the layer
    shape alone -- what lets computeOutput's padded last batch and tests/test_full_size.py's 8-vs-128 comparison hold)."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(77)
    x = rng.normal(size=(5, 64, 10, 12))
    P = _params(rng, 64, 16, 64, False)
    full = _run(rt, x, P, 1, False, 16, 64)
    one = _run(rt, x[4:5], P, 1, False, 16, 64)
    assert np.array_equal(full[4:5], one)


def test_unsupported_shapes_are_refused():
    rt = get_runtime('emu')
    assert rt.lib.dpp_resblock_eval_ok(64, 64, 24, 1, 0) == 0          # bottleneck width
    assert rt.lib.dpp_resblock_eval_ok(32, 64, 16, 1, 0) == 0          # identity needs Cin == Cout
    assert rt.lib.dpp_resblock_eval(None, None) == 10001


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('type_', [0, 1, 2, 3, 4])
def test_every_resnet_type_lowers_its_twenty_blocks_to_one_launch_each(backend, type_):
    """hipdp/evalfuse.py on the graphs the reference's ResNet builds (/root/reference/src/net/resnet.py:120-336): types 0-4 (30-D bottleneck,
    dropout variants, the narrowed stages of type 3 whose stage-3 / 4 openers are IDENTITY blocks because the width does not change).  The
    deterministic engine must recognise all twenty blocks, and its output must equal the layer-by-layer engine's to float32 round-off."""
    from hipdp import engine
    from hipdp import runtime as R
    from net.resnet import ResNet, ResNetParams
    from oracle import nets
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    nJ, nD = (14, 3) if type_ in (1, 4) else (1, 30)
    net = ResNet(np.random.RandomState(23455 + type_), cfgParams=ResNetParams(type=type_, wIn=32, hIn=32, batchSize=2, numJoints=nJ, nDims=nD))
    rng = np.random.RandomState(9)
    for l in net.layers:                      # running statistics / affine parameters away from their (0, 1) initial values
        for p_ in getattr(l, 'params_nontrained', []):
            v = p_.get_value()
            p_.set_value((v + rng.normal(0, 0.1, v.shape) * (1.0 if 'mean' in p_.name else 0.2)).astype(np.float32))
    net.setDeterministic()
    x = nets.synthetic_crops(np.random.RandomState(5), 2, 32, 32, np.float32)
    fused = engine.CompiledNet(net, train=False, runtime=rt)
    plain = engine.CompiledNet(net, train=False, runtime=rt, fuse_blocks=False)
    assert len(fused.fused_blocks) == 20 and len(plain.fused_blocks) == 0
    n_proj = sum(m['shortcut'] is not None for m in fused.fused_blocks)
    assert n_proj == (3 if type_ != 3 else 2)              # type 3: stages 3 and 4 keep the width of stage 2 (128), so only stages 1 and 2 open with a projection
    a, b = fused.forward(x), plain.forward(x)
    assert np.isfinite(a).all() and np.abs(b).max() > 0
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-5 * np.abs(b).max())
    assert len(fused.fwd) < len(plain.fwd) - 30


# ---- bf16 mode of the fused block (round 6, BASELINE config 5's deterministic forward) ------------------------------------------------
def _q(a):
    """round to bfloat16 (nearest even), back in float64: what a bf16-stored tensor / a bf16 MFMA operand holds"""
    return L.bf16_round(np.asarray(a, np.float32)).astype(np.float64)


def _block_ref_bf16(x_stored, P, stride, proj, Nb):
    """The layer-by-layer bf16 path of the engine, restated: every conv output it materialises is bf16-stored (rounded where stored), every
    product rounds both operands (the activation after its prologue; hipdp/engine.py:_gemm_prec: every 1x1 product of a block runs on bf16
    MFMA operands), the shortcut's output is a stored tensor of its own."""
    h = _q(_bnrelu(x_stored, P['bn0']))
    c1 = _q(L.conv2d_fwd(h, _q(P['W1']), P['b1'], (stride, stride), 'half'))
    c2 = _q(L.conv2d_fwd(_q(_bnrelu(c1, P['bn1'])), _q(P['W2']), P['b2'], (1, 1), 'half'))
    a2 = _bnrelu(c2, P['bn2'])
    c3 = L.conv2d_fwd(_q(a2), _q(P['W3']), P['b3'], (1, 1), 'half')
    if not proj:
        return _q(x_stored + c3)
    sc = _q(L.conv2d_fwd(h, _q(P['Wsc']), P['bsc'], (stride, stride), 'half'))
    return _q(c3 + sc)


def _run16(rt, x, P, stride, proj, Nb, Cout, x16):
    from tests.test_bf16_store import bf16_bits, widen
    N, Cin, H, W = x.shape
    f = lambda a: rt.upload(np.ascontiguousarray(a, np.float32))          # noqa: E731
    bns = [ops.bn_eval(*(f(P[k][q]) for q in ('mean', 'inv_std', 'gamma', 'beta'))) for k in ('bn0', 'bn1', 'bn2')]
    xh = layout.nchw_to_nhwc(x).astype(np.float32)
    X = rt.upload(bf16_bits(xh)) if x16 else f(xh)
    Ho, Wo = -(-H // stride), -(-W // stride)
    Y = rt.alloc((N, Ho, Wo, Cout), np.uint16, zero=False)
    kw = dict(Wsc=f(layout.conv_w_to_kernel(P['Wsc'])), bsc=f(P['bsc'])) if proj else {}
    op = ops.resblock_eval(rt, X, N, H, W, Cin, stride, Cout, Nb, bns[0], bns[1], bns[2], f(layout.conv_w_to_kernel(P['W1'])), f(P['b1']),
                           f(layout.conv_w_to_kernel(P['W2'])), f(P['b2']), f(layout.conv_w_to_kernel(P['W3'])), f(P['b3']), Y, **kw)
    assert rt.lib.dpp_resblock_eval_check(op.args[0]) == 0
    op(rt.stream)
    rt.synchronize()
    return layout.nhwc_to_nchw(widen(Y.get()).reshape(N, Ho, Wo, Cout))


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(2, 8, 16, 64, 16, 1, 0, 1), (1, 11, 13, 128, 32, 1, 0, 1), (2, 8, 8, 256, 64, 1, 0, 1), (2, 16, 16, 32, 16, 2, 1, 0),
                                 (1, 9, 7, 64, 32, 2, 1, 1), (2, 16, 16, 128, 64, 2, 1, 1)], ids=lambda c: 'x'.join(str(v) for v in c))
def test_bf16_mode_block_matches_the_layer_by_layer_rounding_model(backend, cfg):
    """dpp_resblock_eval with a bf16-stored output (DPP_ST_C: the bf16 MODE; input bf16-stored, or float32 for the block behind the
    stem) against the float64 restatement of what the layer-by-layer bf16 engine computes (rounding points above).  The products are
    exact (bf16 x bf16 in float32), the sums differ from float64 by float32 round-off, so an intermediate within that of a bfloat16 rounding
    boundary lands on the other neighbour and moves the outputs it feeds: nearly every output element is THE bfloat16 of the reference,
    the rest within a couple of bfloat16 steps at the tensor's scale."""
    rt = get_runtime(backend)
    N, H, W, Cin, Nb, stride, proj, x16 = cfg
    Cout = 4 * Nb
    rng = np.random.RandomState(11 + H * W + Nb)
    x = rng.normal(size=(N, Cin, H, W))
    xs = _q(x) if x16 else np.asarray(x, np.float32).astype(np.float64)
    P = _params(rng, Cin, Nb, Cout, bool(proj))
    ref = _block_ref_bf16(xs, P, stride, bool(proj), Nb)
    out = _run16(rt, x, P, stride, bool(proj), Nb, Cout, bool(x16))
    same = out == ref
    assert same.mean() > 0.97, same.mean()
    assert np.abs(out - ref).max() <= 3 * 2.0 ** -8 * np.abs(ref).max()
    # and it is NOT the float32 block rounded at the end: the intermediate roundings are really there
    f32 = _block_ref(xs, P, stride, bool(proj))
    assert (out == _q(f32)).mean() < 0.9
    # refused: a bf16-stored input with a float32 output
    f = lambda a: rt.upload(np.ascontiguousarray(a, np.float32))          # noqa: E731
    from tests.test_bf16_store import bf16_bits
    bns = [ops.bn_eval(*(f(P[k][q]) for q in ('mean', 'inv_std', 'gamma', 'beta'))) for k in ('bn0', 'bn1', 'bn2')]
    kw = dict(Wsc=f(layout.conv_w_to_kernel(P['Wsc'])), bsc=f(P['bsc'])) if proj else {}
    bad = ops.resblock_eval(rt, rt.upload(bf16_bits(layout.nchw_to_nhwc(x))), N, H, W, Cin, stride, Cout, Nb, bns[0], bns[1], bns[2],
                            f(layout.conv_w_to_kernel(P['W1'])), f(P['b1']), f(layout.conv_w_to_kernel(P['W2'])), f(P['b2']),
                            f(layout.conv_w_to_kernel(P['W3'])), f(P['b3']), rt.alloc((N, -(-H // stride), -(-W // stride), Cout), zero=False), **kw)
    assert rt.lib.dpp_resblock_eval_check(bad.args[0]) == 10002


@pytest.mark.parametrize('backend', BACKENDS)
def test_bf16_deterministic_forward_fuses_its_blocks(backend, monkeypatch):
    """CompiledNet(train=False, bf16=True) lowers its twenty blocks to one launch each as the float32 engine does (rounds 4-5 fell back to
    ~6 launches per block), and its output stays within the distance two layer-by-layer bf16 evaluations with different summation orders
    have from each other: compared with the layer-by-layer bf16 engine (DPP_EVAL_FUSE_BF16 = 0, the path whose every product is pinned
    against the oracle in tests/test_configs.py) relative to the spread of the bf16 path from the float32 path."""
    from hipdp import engine, heuristics
    from hipdp import runtime as R
    from net.resnet import ResNet, ResNetParams
    from oracle import nets
    rt = get_runtime(backend)
    R.set_default_runtime(rt)
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, wIn=32, hIn=32, batchSize=2, numJoints=1, nDims=30))
    rng = np.random.RandomState(9)
    for l in net.layers:
        for p_ in getattr(l, 'params_nontrained', []):
            v = p_.get_value()
            p_.set_value((v + rng.normal(0, 0.1, v.shape) * (1.0 if 'mean' in p_.name else 0.2)).astype(np.float32))
    net.setDeterministic()
    x = nets.synthetic_crops(np.random.RandomState(5), 2, 32, 32, np.float32)
    fused = engine.CompiledNet(net, train=False, runtime=rt, bf16=True)
    assert len(fused.fused_blocks) == 20 and fused.store16
    monkeypatch.setattr(heuristics, 'EVAL_FUSE_BF16', False)
    plain = engine.CompiledNet(net, train=False, runtime=rt, bf16=True)
    assert len(plain.fused_blocks) == 0 and len(fused.fwd) < len(plain.fwd) - 30
    f32 = engine.CompiledNet(net, train=False, runtime=rt, bf16=False)
    a, b, c = fused.forward(x), plain.forward(x), f32.forward(x)
    assert np.isfinite(a).all()
    spread = np.abs(b - c).max()                    # what bf16 rounding does to this net at all
    assert spread > 0 and np.abs(a - b).max() <= 1.5 * spread and np.abs(a - c).max() <= 2.5 * spread
