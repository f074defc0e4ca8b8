"""bf16 STORAGE of activation tensors (ABI v9, DPP_ST_*; BASELINE config 5) at the kernel level.

The contract is crisp, so the tests are exact: a kernel that READS a bf16-stored tensor must give, bit for bit, what the same kernel
gives on the float32 tensor holding the widened values; a kernel that WRITES one must leave round-to-nearest-even(bfloat16) of the
float32 value the float32 kernel writes -- and its fused BatchNorm statistics must be those of the UNROUNDED values (identical to the
float32 kernel's).  Every entry point that can see an activation tensor is covered: the four dpp_gemm kernels (generic tile, K-split,
16-column stream, wave-autonomous strips) in forward / data-gradient / filter-gradient roles, dpp_conv3x3 (f32 and bf16 operands) and
its two filter-gradient kernels, the 1x1 / FC filter-gradient streams, dpp_fc_gemm on both of its kernels, the BatchNorm kernels and
the stem."""
import numpy as np
import pytest

from hipdp import ops
from hipdp.lib import Act
from tests.backends import BACKENDS, get_runtime


def bf16_bits(a):
    """float32 -> bfloat16 bit patterns (uint16), round to nearest even."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def widen(bits):
    return (np.ascontiguousarray(bits, np.uint16).astype(np.uint32) << 16).view(np.float32)


def both(rt, a):
    """(buffer holding `a` rounded to bf16 as bf16 bits, float32 buffer holding the same values widened)"""
    b = bf16_bits(a)
    return rt.upload(b), rt.upload(widen(b).reshape(np.shape(a)))


class _BN(object):
    pass


def _bn_coeffs(rt, rng, C):
    bn = _BN()
    vals = (rng.normal(0, 0.3, C), rng.uniform(0.5, 1.5, C), rng.normal(0, 0.3, C), rng.uniform(0.5, 1.5, C))
    bn.mean, bn.scale, bn.beta_buf, bn.inv_std = (rt.upload(v.astype(np.float32)) for v in vals)
    return bn


FWD_CASES = [  # (K, N, M, tile, variant)
    (64, 32, 128, (64, 32, 4), 0), (64, 64, 128, (64, 64, 4), 0), (16, 64, 256, (64, 16, 4), 0), (40, 24, 100, (0, 0, 0), 0),
    (256, 64, 96, (32, 64, 4), 2), (128, 32, 96, (32, 32, 4), 2), (64, 16, 128, (128, 16, 4), 3),
    (64, 256, 96, (32, 64, 4), 4), (32, 128, 128, (64, 64, 4), 4), (16, 64, 256, (128, 64, 4), 4)]


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', FWD_CASES)
def test_gemm_forward_reads_and_writes_bf16_tensors(backend, cfg):
    """1x1 convolution forward: A (with the BatchNorm + ReLU prologue), the residual and the output bf16-stored."""
    rt = get_runtime(backend)
    K, N, M, tile, variant = cfg
    rng = np.random.RandomState(61)
    X16, X32 = both(rt, rng.normal(size=(M, K)) * 2 + 1)
    R16, R32 = both(rt, rng.normal(size=(M, N)))
    Wk = rt.upload((rng.normal(size=(N, K)) * 0.3).astype(np.float32))
    mean, scale, beta = (rt.upload(rng.normal(size=K).astype(np.float32)) for _ in range(3))
    bias = rt.upload(rng.normal(size=N).astype(np.float32))
    rows = tile[0] if tile[0] else 64
    nblk = -(-M // rows)
    out = {}
    for tag, X, R, dt in (('f32', X32, R32, np.float32), ('bf16', X16, R16, np.uint16)):
        Y = rt.alloc((M, N), dt, zero=False)
        stats = rt.alloc((nblk, 2, N), zero=False) if tile[0] else None
        L = ops.gemm(rt, X, Wk, Y, M, N, K, 1, 1, K, K, N, actA=ops.act(Act.BN_RELU, mean, scale, beta, K), bias=bias, residual=R, tile=tile,
                     variant=variant, epi=ops.epilogue(stats=stats) if stats is not None else None)
        if variant:
            assert ops.gemm_variant_rows(rt, L) == tile[0], "variant %d refuses the %s layout" % (variant, tag)
        L(rt.stream)
        rt.synchronize()
        out[tag] = (Y.get(), stats.get() if stats is not None else None)
    if variant == 3:
        # round 6: on bf16-stored tensors the 16-column stream runs with its product transposed in the accumulators
        # (gemm_stream16t_kernel: 8-byte epilogue accesses) -- the same sums in another order, so "the rounding of the float32 kernel's
        # value" holds up to float32 round-off in front of the rounding, and the statistics are those of the unrounded values to round-off
        f = out['f32'][0]
        want = widen(bf16_bits(f)).reshape(f.shape)
        got = widen(out['bf16'][0]).reshape(f.shape)
        assert (got == want).mean() > 0.999 and np.abs(got - f).max() <= np.abs(f).max() * 2.0 ** -8
        np.testing.assert_allclose(out['bf16'][1], out['f32'][1], rtol=0, atol=2e-6 * np.abs(out['f32'][1]).max())
        return
    assert np.array_equal(out['bf16'][0], bf16_bits(out['f32'][0]))
    if out['f32'][1] is not None:
        assert np.array_equal(out['bf16'][1], out['f32'][1])           # statistics of the UNROUNDED values


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(64, 32, 128, (64, 32, 4), 0), (256, 64, 96, (32, 64, 4), 2), (16, 64, 128, (64, 64, 4), 3), (64, 256, 96, (32, 64, 4), 4),
                                 (32, 128, 128, (64, 64, 4), 4)])
def test_gemm_data_gradient_reads_a_bf16_batchnorm_input(backend, cfg):
    """Data gradient with the fused BatchNorm-backward epilogue: bn_x is the (bf16-stored) forward tensor, everything else float32."""
    rt = get_runtime(backend)
    K, N, M, tile, variant = cfg
    rng = np.random.RandomState(62)
    dY = rt.upload(rng.normal(size=(M, K)).astype(np.float32))
    W2 = rt.upload((rng.normal(size=(K, N)) * 0.3).astype(np.float32))
    share = rng.normal(size=(M, N)).astype(np.float32)
    bx16, bx32 = both(rt, rng.normal(size=(M, N)))
    bn = _bn_coeffs(rt, rng, N)
    nb = M // tile[0]
    out = {}
    for tag, bx in (('f32', bx32), ('bf16', bx16)):
        dH = rt.upload(share)
        part = rt.alloc((nb, 2, N), zero=False)
        L = ops.gemm(rt, dY, W2, dH, M, N, K, 1, 0, K, N, N, residual=dH, tile=tile, variant=variant,
                     epi=ops.epilogue(bn=bn, bn_x=bx, bn_relu=True, bn_partial=part))
        if variant:
            assert ops.gemm_variant_rows(rt, L) == tile[0]
        L(rt.stream)
        rt.synchronize()
        out[tag] = (dH.get(), part.get())
    if variant == 3:          # (the transposed-accumulator stream on the bf16-stored BatchNorm input: other summation order, see above)
        np.testing.assert_allclose(out['bf16'][0], out['f32'][0], rtol=0, atol=2e-6 * np.abs(out['f32'][0]).max())
        np.testing.assert_allclose(out['bf16'][1], out['f32'][1], rtol=0, atol=2e-6 * np.abs(out['f32'][1]).max())
        return
    assert np.array_equal(out['bf16'][0], out['f32'][0]) and np.array_equal(out['bf16'][1], out['f32'][1])


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(16, 64, 256, 1), (64, 16, 256, 1), (64, 256, 128, 1), (32, 128, 192, 2), (24, 40, 100, 1)])
def test_filter_gradients_read_bf16_activations(backend, cfg):
    """dW = dY^T . act(X) with X bf16-stored: the generic split-K layout of dpp_gemm and the row stream of dpp_wgrad_stream."""
    rt = get_runtime(backend)
    Co, Ci, M, stride = cfg
    rng = np.random.RandomState(63)
    rows = M * stride * stride
    X16, X32 = both(rt, rng.normal(size=(rows, Ci)) + 0.5)
    dY = rt.upload(rng.normal(size=(M, Co)).astype(np.float32))
    mean, scale, beta = (rt.upload(rng.normal(size=Ci).astype(np.float32)) for _ in range(3))
    act = ops.act(Act.BN_RELU, mean, scale, beta, Ci)
    mp = None
    if stride == 2:
        from hipdp.lib import RowMap
        Ho = Wo = int(round((M // 3) ** 0.5))
        assert 3 * Ho * Wo == M
        mp = RowMap.strided(2, Ho, Wo, 2 * Ho, 2 * Wo)
    got = {}
    for tag, X in (('f32', X32), ('bf16', X16)):
        part = rt.alloc((3, Co, Ci), zero=False)
        ops.gemm(rt, dY, X, None, Co, Ci, M, 0, 0, Co, Ci, Ci, mapB=mp, actB=act, splitk=3, partial=part)(rt.stream)
        res = [part]
        nsl = rt.lib.dpp_wgrad_stream_slices(Co, Ci, M, 32)
        if nsl > 0:
            p2 = rt.alloc((nsl, Co, Ci), zero=False)
            ops.wgrad_stream(rt, dY, Co, X, Ci, M, 32, p2, mapX=mp, actX=act)(rt.stream)
            res.append(p2)
        rt.synchronize()
        got[tag] = [r.get() for r in res]
    assert all(np.array_equal(a, b) for a, b in zip(got['bf16'], got['f32']))
    assert np.abs(got['f32'][0]).max() > 0.1


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(2, 8, 8, 64, 64, 64), (3, 16, 16, 16, 16, 64), (2, 12, 20, 32, 32, 128), (1, 8, 8, 64, 32, 64)])
@pytest.mark.parametrize('precision', [0, 1])
def test_conv3x3_on_bf16_stored_tensors(backend, cfg, precision):
    """dpp_conv3x3 / dpp_conv3x3_bf16: forward with bf16 X, residual and Y (+ fused statistics); data gradient reading a bf16 bn_x;
    both filter-gradient kernels reading a bf16 X."""
    rt = get_runtime(backend)
    N, H, W, Ci, Co, bm = cfg
    rng = np.random.RandomState(64)
    X16, X32 = both(rt, rng.normal(size=(N, H, W, Ci)) + 0.3)
    R16, R32 = both(rt, rng.normal(size=(N, H, W, Co)))
    Wk = rt.upload((rng.normal(size=(Co, 9, Ci)) * 0.2).astype(np.float32))
    bias = rt.upload(rng.normal(size=Co).astype(np.float32))
    mean, scale, beta = (rt.upload(v.astype(np.float32)) for v in (rng.normal(size=Ci) * 0.3, rng.uniform(0.5, 1.5, Ci), rng.normal(size=Ci) * 0.3))
    act = ops.act(Act.BN_RELU, mean, scale, beta, Ci)
    th, tw, img = (__import__('ctypes').c_int() for _ in range(3))
    nblk = rt.lib.dpp_conv3x3_tiling(N, H, W, bm, th, tw, img)
    full = H % th.value == 0 and W % tw.value == 0 and (N % img.value == 0 or nblk == 1)
    out = {}
    for tag, X, R, dt in (('f32', X32, R32, np.float32), ('bf16', X16, R16, np.uint16)):
        Y = rt.alloc((N, H, W, Co), dt, zero=False)
        stats = rt.alloc((nblk, 2, Co), zero=False) if full else None
        ops.conv3x3(rt, X, N, H, W, Ci, Wk, Co, Y, actX=act, bias=bias, residual=R, bm=bm, epi=ops.epilogue(stats=stats) if full else None,
                    precision=precision)(rt.stream)
        rt.synchronize()
        out[tag] = (Y.get(), stats.get() if full else None)
    assert np.array_equal(out['bf16'][0], bf16_bits(out['f32'][0]))
    if full:
        assert np.array_equal(out['bf16'][1], out['f32'][1])
    # data gradient: dY f32 -> dX f32, the epilogue reads the bf16-stored BatchNorm input of the layer below
    dY = rt.upload(rng.normal(size=(N, H, W, Co)).astype(np.float32))
    Wd = rt.alloc((Ci, 9, Co), zero=False)
    ops.conv3x3_wtrans(rt, Wk, Co, Ci, Wd)(rt.stream)
    bn = _bn_coeffs(rt, rng, Ci)
    nb2 = rt.lib.dpp_conv3x3_tiling(N, H, W, bm, None, None, None)
    got = {}
    for tag, bx in (('f32', X32), ('bf16', X16)):
        dX = rt.alloc((N, H, W, Ci), zero=False)
        part = rt.alloc((nb2, 2, Ci), zero=False)
        ops.conv3x3(rt, dY, N, H, W, Co, Wd, Ci, dX, bm=bm, epi=ops.epilogue(bn=bn, bn_x=bx, bn_relu=True, bn_partial=part), precision=precision)(rt.stream)
        rt.synchronize()
        got[tag] = (dX.get(), part.get())
    assert np.array_equal(got['bf16'][0], got['f32'][0]) and np.array_equal(got['bf16'][1], got['f32'][1])
    if precision:
        return
    # filter gradients (f32 MFMA in both modes): LDS-tiled kernel and row stream
    wg = {}
    for tag, X in (('f32', X32), ('bf16', X16)):
        nb3 = rt.lib.dpp_conv3x3_wgrad_blocks(N, H, W, Ci, Co, 64)
        part = rt.alloc((nb3, Co, 9, Ci), zero=False)
        ops.conv3x3_wgrad(rt, X, N, H, W, Ci, dY, Co, part, actX=act, bm=64)(rt.stream)
        res = [part]
        nsl = rt.lib.dpp_wgrad3_stream_slices(Co, Ci, N, H, W, 20)
        if nsl > 0:
            p2 = rt.alloc((nsl, Co, 9, Ci), zero=False)
            ops.wgrad3_stream(rt, dY, Co, X, Ci, N, H, W, 20, p2, actX=act)(rt.stream)
            res.append(p2)
        rt.synchronize()
        wg[tag] = [r.get() for r in res]
    assert all(np.array_equal(a, b) for a, b in zip(wg['bf16'], wg['f32']))


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('precision', [0, 1])
@pytest.mark.parametrize('shape', [(128, 128, 1024), (24, 40, 200)])
def test_fc1_reads_a_bf16_stored_map(backend, precision, shape):
    """The HiddenLayer behind the last conv map: forward (A = the flattened bf16 map with its BatchNorm + ReLU prologue) on both kernels
    of dpp_fc_gemm (whole 128-row tiles: the three-stage stream; ragged: the double-buffered one), the weight gradient on dpp_fc_gemm
    (A = X^T) and on dpp_fc_wgrad_stream."""
    rt = get_runtime(backend)
    Nb, Nout, K = shape
    Cc = 8 if K == 200 else 32
    rng = np.random.RandomState(65)
    X16, X32 = both(rt, rng.normal(size=(Nb, K)) + 0.2)
    Wm = rt.upload((rng.normal(size=(K, Nout)) * 0.1).astype(np.float32))
    mean, scale, beta = (rt.upload(rng.normal(size=Cc).astype(np.float32)) for _ in range(3))
    act = ops.act(Act.BN_RELU, mean, scale, beta, Cc)
    bias = rt.upload(rng.normal(size=Nout).astype(np.float32))
    dY = rt.upload(rng.normal(size=(Nb, Nout)).astype(np.float32))
    got = {}
    for tag, X in (('f32', X32), ('bf16', X16)):
        Y = rt.alloc((Nb, Nout), zero=False)
        ops.fc_gemm(rt, X, Wm, Y, Nb, Nout, K, 1, 0, K, Nout, Nout, actA=act, bias=bias, precision=precision)(rt.stream)
        dW = rt.alloc((K, Nout), zero=False)
        ops.fc_gemm(rt, X, dY, dW, K, Nout, Nb, 0, 0, K, Nout, Nout, actA=act, precision=precision)(rt.stream)
        res = [Y, dW]
        if precision == 0 and rt.lib.dpp_fc_wgrad_stream_ok(Nb, K, Nout):
            dW2 = rt.alloc((K, Nout), zero=False)
            ops.fc_wgrad_stream(rt, X, dY, dW2, Nb, K, Nout, actX=act)(rt.stream)
            res.append(dW2)
        # and the generic dpp_gemm (what batches that are no multiple of 128 take)
        Y2 = rt.alloc((Nb, Nout), zero=False)
        ops.gemm(rt, X, Wm, Y2, Nb, Nout, K, 1, 0, K, Nout, Nout, actA=act, bias=bias)(rt.stream)
        res.append(Y2)
        rt.synchronize()
        got[tag] = [r.get() for r in res]
    assert all(np.array_equal(a, b) for a, b in zip(got['bf16'], got['f32']))
    assert np.abs(got['f32'][0]).max() > 0.1


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(96, 16, 32), (200, 64, 48), (64, 256, 32)])
def test_batchnorm_kernels_read_bf16_tensors(backend, cfg):
    """bn_stats_partial (the non-fused statistics path), bn_bwd_reduce and bn_bwd_apply on a bf16-stored x."""
    rt = get_runtime(backend)
    M, Cc, rpb = cfg
    rng = np.random.RandomState(66)
    x16, x32 = both(rt, rng.normal(size=(M, Cc)) * 1.5 + 3.0)
    nb = -(-M // rpb)
    bn = _bn_coeffs(rt, rng, Cc)
    c1, c2 = (rt.upload((rng.normal(size=Cc) * 0.1).astype(np.float32)) for _ in range(2))
    dA = rt.upload(rng.normal(size=(M, Cc)).astype(np.float32))
    add = rt.upload(rng.normal(size=(M, Cc)).astype(np.float32))
    got = {}
    for tag, x in (('f32', x32), ('bf16', x16)):
        part = rt.alloc((nb, 2, Cc), zero=False)
        ops.bn_stats_partial(rt, x, M, Cc, rpb, part)(rt.stream)
        G, part2 = rt.alloc((M, Cc), zero=False), rt.alloc((nb, 2, Cc), zero=False)
        ops.bn_bwd_reduce(rt, dA, x, M, Cc, bn.mean, bn.inv_std, bn.scale, bn.beta_buf, 1, G, rpb, part2)(rt.stream)
        dX, cs = rt.alloc((M, Cc), zero=False), rt.alloc((nb, Cc), zero=False)
        ops.bn_bwd_apply(rt, G, x, M, Cc, bn.mean, bn.inv_std, bn.scale, c1, c2, dX, add=add, rpb=rpb, colsum=cs)(rt.stream)
        rt.synchronize()
        got[tag] = [b.get() for b in (part, G, part2, dX, cs)]
    assert all(np.array_equal(a, b) for a, b in zip(got['bf16'], got['f32']))


@pytest.mark.parametrize('backend', BACKENDS)
def test_stem_writes_a_bf16_tensor(backend):
    rt = get_runtime(backend)
    N, H, W, Co = 2, 32, 32, 32
    rng = np.random.RandomState(67)
    x = rng.uniform(-1, 1, size=(N, H, W)).astype(np.float32)
    x[:, :, W // 2:] = 1.0
    X, Wk, b = rt.upload(x), rt.upload((rng.normal(size=(Co, 25)) * 0.3).astype(np.float32)), rt.upload(rng.normal(size=Co).astype(np.float32))
    out = {}
    for tag, dt in (('f32', np.float32), ('bf16', np.uint16)):
        Y = rt.alloc((N, H // 2, W // 2, Co), dt, zero=False)
        arg = rt.alloc((N, H // 2, W // 2, Co), np.uint8, zero=False)
        stats = rt.alloc((N * 4, 2, Co), zero=False)
        ops.stem_fwd(rt, X, N, H, W, Wk, b, Co, Y, arg, stats)(rt.stream)
        rt.synchronize()
        out[tag] = (Y.get(), arg.get(), stats.get())
    assert np.array_equal(out['bf16'][0], bf16_bits(out['f32'][0]))
    assert np.array_equal(out['bf16'][1], out['f32'][1]) and np.array_equal(out['bf16'][2], out['f32'][2])


@pytest.mark.parametrize('backend', BACKENDS)
def test_unsupported_storage_combinations_are_refused(backend):
    """What the kernels do not take in the storage mode is refused, not mis-read: mixed C / residual storage, the mode-4 operand,
    a bf16 B operand on the variants, unknown bits."""
    rt = get_runtime(backend)
    M, N, K = 64, 64, 64
    A16 = rt.alloc((M, K), np.uint16)
    A32, Wk = rt.alloc((M, K)), rt.alloc((N, K))
    Y16, Y32 = rt.alloc((M, N), np.uint16), rt.alloc((M, N))
    with pytest.raises(ValueError):
        ops.gemm(rt, A32, Wk, Y16, M, N, K, 1, 1, K, K, N, residual=Y32)
    L = ops.gemm(rt, A32, rt.alloc((N, K), np.uint16), Y32, M, N, K, 1, 1, K, K, N, tile=(32, 64, 4), variant=4)
    assert ops.gemm_variant_rows(rt, L) == 0
    L = ops.gemm(rt, A16, Wk, Y32, M, N, K, 1, 1, K, K, N)
    L.keep[0].store = 64
    with pytest.raises(Exception):
        L(rt.stream)
    bn = _BN()
    bn.mean, bn.scale = rt.alloc(K), rt.alloc(K)
    L = ops.gemm(rt, A16, rt.alloc((K, N)), Y32, M, N, K, 1, 0, K, N, N, actA=ops.act_bn_bwd(bn, rt.alloc(K), rt.alloc(K), rt.alloc((M, K)), K))
    with pytest.raises(Exception):
        L(rt.stream)


# ---- bf16-stored GRADIENTS of the activation tensors (the A / C roles of the backward calls) ------------------------------------------
@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(64, 32, 128, (64, 32, 4), 0), (256, 64, 96, (32, 64, 4), 2), (64, 16, 128, (128, 16, 4), 3), (16, 64, 128, (64, 64, 4), 3),
                                 (64, 256, 96, (32, 64, 4), 4), (32, 128, 128, (64, 64, 4), 4), (16, 64, 256, (128, 64, 4), 4)])
def test_gemm_data_gradient_with_bf16_gradients(backend, cfg):
    """Data gradient on bf16-stored gradients: dY (A) read as bf16, the masked gradient written as bf16 onto an earlier bf16 share
    (residual = C), bn_x bf16 -- the written tensor is the rounded float32 result, the BatchNorm-backward sums are those of the
    UNROUNDED values."""
    rt = get_runtime(backend)
    K, N, M, tile, variant = cfg
    rng = np.random.RandomState(71)
    dY16, dY32 = both(rt, rng.normal(size=(M, K)))
    W2 = rt.upload((rng.normal(size=(K, N)) * 0.3).astype(np.float32))
    share = rng.normal(size=(M, N))
    bx16, bx32 = both(rt, rng.normal(size=(M, N)))
    bn = _bn_coeffs(rt, rng, N)
    nb = M // tile[0]
    out = {}
    for tag, dY, bx in (('f32', dY32, bx32), ('bf16', dY16, bx16)):
        s16, s32 = both(rt, share)
        dH = s32 if tag == 'f32' else s16
        part = rt.alloc((nb, 2, N), zero=False)
        L = ops.gemm(rt, dY, W2, dH, M, N, K, 1, 0, K, N, N, residual=dH, tile=tile, variant=variant,
                     epi=ops.epilogue(bn=bn, bn_x=bx, bn_relu=True, bn_partial=part))
        if variant:
            assert ops.gemm_variant_rows(rt, L) == tile[0]
        L(rt.stream)
        rt.synchronize()
        out[tag] = (dH.get(), part.get())
    assert np.array_equal(out['bf16'][0], bf16_bits(out['f32'][0]))
    # the BatchNorm-backward sums of a bf16-stored gradient are those of the values AS STORED (bn_bwd_apply's c1 / c2 are then exactly
    # the means of the G it reads): against float64 sums of the widened stored tensor
    Gs = widen(out['bf16'][0]).astype('f8').reshape(M, N)
    bx = widen(bx16.get()).astype('f8').reshape(M, N)
    xhat = (bx - bn.mean.get().astype('f8')) * bn.inv_std.get().astype('f8')
    p = out['bf16'][1].reshape(2, N, nb).astype('f8').sum(axis=2)
    np.testing.assert_allclose(p[0], Gs.sum(0), rtol=0, atol=2e-6 * np.abs(Gs).sum(0).max())
    np.testing.assert_allclose(p[1], (Gs * xhat).sum(0), rtol=0, atol=2e-6 * np.abs(Gs * xhat).sum(0).max())
    assert not np.array_equal(out['bf16'][1], out['f32'][1])


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(16, 64, 256, 1), (64, 16, 256, 1), (64, 256, 128, 1), (32, 128, 192, 2)])
def test_filter_gradients_read_bf16_gradients(backend, cfg):
    """dW = dY^T . act(X) with BOTH operands bf16-stored (and each alone): the generic layout and the row stream."""
    rt = get_runtime(backend)
    Co, Ci, M, stride = cfg
    rng = np.random.RandomState(72)
    rows = M * stride * stride
    X16, X32 = both(rt, rng.normal(size=(rows, Ci)) + 0.5)
    Y16, Y32 = both(rt, rng.normal(size=(M, Co)))
    mean, scale, beta = (rt.upload(rng.normal(size=Ci).astype(np.float32)) for _ in range(3))
    act = ops.act(Act.BN_RELU, mean, scale, beta, Ci)
    mp = None
    if stride == 2:
        from hipdp.lib import RowMap
        Ho = Wo = int(round((M // 3) ** 0.5))
        mp = RowMap.strided(2, Ho, Wo, 2 * Ho, 2 * Wo)
    got = {}
    for tag, X, dY in (('f32', X32, Y32), ('both', X16, Y16), ('dy', X32, Y16)):
        part = rt.alloc((3, Co, Ci), zero=False)
        ops.gemm(rt, dY, X, None, Co, Ci, M, 0, 0, Co, Ci, Ci, mapB=mp, actB=act, splitk=3, partial=part)(rt.stream)
        res = [part]
        nsl = rt.lib.dpp_wgrad_stream_slices(Co, Ci, M, 32)
        if nsl > 0:
            p2 = rt.alloc((nsl, Co, Ci), zero=False)
            ops.wgrad_stream(rt, dY, Co, X, Ci, M, 32, p2, mapX=mp, actX=act)(rt.stream)
            res.append(p2)
        rt.synchronize()
        got[tag] = [r.get() for r in res]
    for tag in ('both', 'dy'):
        assert all(np.array_equal(a, b) for a, b in zip(got[tag], got['f32'])), tag


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(2, 8, 8, 64, 64), (3, 16, 16, 16, 16), (2, 12, 20, 32, 32)])
def test_conv3x3_filter_gradients_read_bf16_gradients(backend, cfg):
    rt = get_runtime(backend)
    N, H, W, Ci, Co = cfg
    rng = np.random.RandomState(73)
    X16, X32 = both(rt, rng.normal(size=(N, H, W, Ci)) + 0.3)
    Y16, Y32 = both(rt, rng.normal(size=(N, H, W, Co)))
    mean, scale, beta = (rt.upload(v.astype(np.float32)) for v in (rng.normal(size=Ci) * 0.3, rng.uniform(0.5, 1.5, Ci), rng.normal(size=Ci) * 0.3))
    act = ops.act(Act.BN_RELU, mean, scale, beta, Ci)
    wg = {}
    for tag, X, dY in (('f32', X32, Y32), ('both', X16, Y16), ('dy', X32, Y16)):
        nb3 = rt.lib.dpp_conv3x3_wgrad_blocks(N, H, W, Ci, Co, 64)
        part = rt.alloc((nb3, Co, 9, Ci), zero=False)
        ops.conv3x3_wgrad(rt, X, N, H, W, Ci, dY, Co, part, actX=act, bm=64)(rt.stream)
        res = [part]
        nsl = rt.lib.dpp_wgrad3_stream_slices(Co, Ci, N, H, W, 20)
        if nsl > 0:
            p2 = rt.alloc((nsl, Co, 9, Ci), zero=False)
            ops.wgrad3_stream(rt, dY, Co, X, Ci, N, H, W, 20, p2, actX=act)(rt.stream)
            res.append(p2)
        rt.synchronize()
        wg[tag] = [r.get() for r in res]
    for tag in ('both', 'dy'):
        assert all(np.array_equal(a, b) for a, b in zip(wg[tag], wg['f32'])), tag


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(96, 16, 32), (200, 64, 48), (64, 256, 32)])
def test_batchnorm_backward_on_bf16_gradients(backend, cfg):
    """bn_bwd_reduce (dA / G bf16, in place) and bn_bwd_apply in every storage combination of G and dX (+ the added gradient)."""
    rt = get_runtime(backend)
    M, Cc, rpb = cfg
    rng = np.random.RandomState(74)
    x16, x32 = both(rt, rng.normal(size=(M, Cc)) * 1.5 + 3.0)
    nb = -(-M // rpb)
    bn = _bn_coeffs(rt, rng, Cc)
    c1, c2 = (rt.upload((rng.normal(size=Cc) * 0.1).astype(np.float32)) for _ in range(2))
    dA = rng.normal(size=(M, Cc))
    G16, G32 = both(rt, rng.normal(size=(M, Cc)))
    A16, A32 = both(rt, rng.normal(size=(M, Cc)))
    # reduce: the masked gradient is written rounded, the sums are those of the unrounded values
    red = {}
    for tag in ('f32', 'bf16'):
        d16, d32 = both(rt, dA)
        d = d32 if tag == 'f32' else d16
        part2 = rt.alloc((nb, 2, Cc), zero=False)
        ops.bn_bwd_reduce(rt, d, x16, M, Cc, bn.mean, bn.inv_std, bn.scale, bn.beta_buf, 1, d, rpb, part2)(rt.stream)
        rt.synchronize()
        red[tag] = (d.get(), part2.get())
    assert np.array_equal(red['bf16'][0], bf16_bits(red['f32'][0]))
    Gs = widen(red['bf16'][0]).astype('f8').reshape(M, Cc)             # sums of the values as stored
    p = red['bf16'][1].reshape(2, Cc, nb).astype('f8').sum(axis=2)
    np.testing.assert_allclose(p[0], Gs.sum(0), rtol=0, atol=2e-6 * np.abs(Gs).sum(0).max())
    ref = {}
    for g16 in (False, True):
        for o16 in (False, True):
            dX, cs = rt.alloc((M, Cc), np.uint16 if o16 else np.float32, zero=False), rt.alloc((nb, Cc), zero=False)
            ops.bn_bwd_apply(rt, G16 if g16 else G32, x16, M, Cc, bn.mean, bn.inv_std, bn.scale, c1, c2, dX, add=A16 if o16 else A32, rpb=rpb,
                             colsum=cs)(rt.stream)
            rt.synchronize()
            ref[(g16, o16)] = (dX.get(), cs.get())
    base = ref[(False, False)]
    assert np.array_equal(ref[(True, False)][0], base[0]) and np.array_equal(ref[(True, False)][1], base[1])
    for g16 in (False, True):
        assert np.array_equal(ref[(g16, True)][0], bf16_bits(base[0]))
        # column sums (the bias gradient of the producing conv) of a bf16-stored dX: of the values as stored
        stored = widen(ref[(g16, True)][0]).astype('f8').reshape(M, Cc)
        np.testing.assert_allclose(ref[(g16, True)][1].astype('f8').sum(0), stored.sum(0), rtol=0, atol=2e-6 * np.abs(stored).sum(0).max())


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(64, 256, 96, 32), (32, 128, 128, 64), (64, 64, 64, 32)])
@pytest.mark.parametrize('stored', [False, True])
def test_expand_kernel_on_bf16_mfma_operands(backend, cfg, stored):
    """dpp_gemm variant 4 with precision = 1: both operands rounded to bfloat16 (the activation AFTER its BatchNorm + ReLU prologue),
    f32 accumulation on v_mfma_f32_16x16x32_bf16 -- forward and data gradient against the float64 product of the ROUNDED operands
    (the rounding is the whole difference to the f32 kernel, so the f32 tolerance applies), on float32 and on bf16-stored tensors;
    K = 16 is refused by this kernel (the other dpp_gemm kernels take the precision since round 6:
    test_gemm_tile_ksplit_and_stream_kernels_on_bf16_mfma_operands)."""
    rt = get_runtime(backend)
    K, N, M, rpw = cfg
    rng = np.random.RandomState(81)
    X16, X32 = both(rt, rng.normal(size=(M, K)) * 2 + 1)
    R16, R32 = both(rt, rng.normal(size=(M, N)))
    Wk = (rng.normal(size=(N, K)) * 0.3).astype(np.float32)
    mean, scale, beta = (rng.normal(size=K).astype(np.float32) for _ in range(3))
    bias = rng.normal(size=N).astype(np.float32)
    X, R = (X16, R16) if stored else (X32, R32)
    xv, rv = widen(X16.get()).reshape(M, K), widen(R16.get()).reshape(M, N)
    Y = rt.alloc((M, N), np.uint16 if stored else np.float32, zero=False)
    nblk = M // rpw
    stats = rt.alloc((nblk, 2, N), zero=False)
    L = ops.gemm(rt, X, rt.upload(Wk), Y, M, N, K, 1, 1, K, K, N, actA=ops.act(Act.BN_RELU, rt.upload(mean), rt.upload(scale), rt.upload(beta), K),
                 bias=rt.upload(bias), residual=R, tile=(rpw, 64, 4), variant=4, epi=ops.epilogue(stats=stats), precision=1)
    assert ops.gemm_variant_rows(rt, L) == rpw
    L(rt.stream)
    rt.synchronize()
    dx = (xv - mean).astype(np.float32)
    a = np.maximum((dx.astype('f8') * scale.astype('f8') + beta.astype('f8')).astype(np.float32), 0)        # dpp_act4's arithmetic
    ref = widen(bf16_bits(a)).astype('f8') @ widen(bf16_bits(Wk)).astype('f8').T + bias + rv
    got = widen(Y.get()).reshape(M, N) if stored else Y.get()
    tol = 3e-6 * np.sqrt(K) * np.abs(ref).max()
    if stored:
        assert np.abs(got - ref).max() <= np.abs(ref).max() * 2.0 ** -8 + tol        # within one bf16 rounding of the f32 result ...
        assert (got == widen(bf16_bits(ref.astype(np.float32)))).mean() > 0.99          # ... and almost always THE rounding of it
    else:
        np.testing.assert_allclose(got, ref, rtol=0, atol=tol)
    p = stats.get().reshape(2, N, nblk)
    np.testing.assert_allclose(p[0].mean(axis=1), ref.mean(0), rtol=0, atol=1e-5 * np.abs(ref).max())
    # data gradient: dX = bf16(dY) . bf16(W2) (+ share), BatchNorm-backward epilogue
    W2 = (rng.normal(size=(K, N)) * 0.3).astype(np.float32)
    dY16, dY32 = both(rt, rng.normal(size=(M, K)))
    bnx = rng.normal(size=(M, N)).astype(np.float32)
    bn = _bn_coeffs(rt, rng, N)
    dH = rt.alloc((M, N), zero=False)
    part = rt.alloc((nblk, 2, N), zero=False)
    L = ops.gemm(rt, dY16 if stored else dY32, rt.upload(W2), dH, M, N, K, 1, 0, K, N, N, tile=(rpw, 64, 4), variant=4, precision=1,
                 epi=ops.epilogue(bn=bn, bn_x=rt.upload(bnx), bn_relu=True, bn_partial=part))
    assert ops.gemm_variant_rows(rt, L) == rpw
    L(rt.stream)
    rt.synchronize()
    keep = ((bnx.astype('f8') - bn.mean.get()) * bn.scale.get() + bn.beta_buf.get()) >= 0
    g = np.where(keep, widen(dY16.get()).reshape(M, K).astype('f8') @ widen(bf16_bits(W2)).astype('f8'), 0.0)
    np.testing.assert_allclose(dH.get(), g, rtol=0, atol=3e-6 * np.sqrt(K) * np.abs(g).max())
    # K = 16 (round 6): one 32-deep step whose upper half is zero
    x16k = (rng.normal(size=(M, 16))).astype(np.float32)
    w16k = (rng.normal(size=(64, 16)) * 0.3).astype(np.float32)
    Y16k = rt.alloc((M, 64), zero=False)
    L16 = ops.gemm(rt, rt.upload(x16k), rt.upload(w16k), Y16k, M, 64, 16, 1, 1, 16, 16, 64, tile=(32, 64, 4), variant=4, precision=1)
    assert ops.gemm_variant_rows(rt, L16) == 32
    L16(rt.stream)
    rt.synchronize()
    r16 = widen(bf16_bits(x16k)).astype('f8') @ widen(bf16_bits(w16k)).astype('f8').T
    np.testing.assert_allclose(Y16k.get(), r16, rtol=0, atol=3e-6 * 4 * np.abs(r16).max())
    with pytest.raises(Exception):          # the row-stream kernel (variant 1) has no bf16 path
        ops.gemm(rt, X32, rt.upload(Wk), rt.alloc((M, N)), M, N, K, 1, 1, K, K, N, precision=1, variant=1, tile=(64, 64, 4))(rt.stream)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('stored', [False, True])
@pytest.mark.parametrize('case', ['tile_k64_n32', 'tile_k128_n32', 'tile_k40_ragged', 'ksplit256', 'ksplit128', 'stream_64_16', 'stream_16_64', 'stream_64_16_walk',
                                  'stream_16_64_walk'])
def test_gemm_tile_ksplit_and_stream_kernels_on_bf16_mfma_operands(backend, stored, case):
    """dpp_gemm variants 0, 2 and 3 with precision = 1 (round 6, VERDICT r5 item 1(a): bf16 MFMA operands for gemm_kernel, gemm_ksplit_kernel
    and gemm_stream16_kernel): forward (K-contiguous activations with the BatchNorm + ReLU prologue x K-contiguous filters), data gradient
    (dY x W[K][N]) and filter gradient (both operands pixel-major, reduction over pixels, split K) against the float64 product of the
    operands ROUNDED to bfloat16 -- the activation after its prologue -- on float32 and on bf16-stored tensors.
    /root/reference/src/net/convlayer.py:230-240, T.grad at /root/reference/src/trainer/poseregnettrainer.py:110-111."""
    rt = get_runtime(backend)
    K, N, M, variant, tile = {'tile_k64_n32': (64, 32, 192, 0, (64, 32, 4)), 'tile_k128_n32': (128, 32, 128, 0, (64, 16, 4)),
                              'tile_k40_ragged': (40, 24, 100, 0, (0, 0, 0)), 'ksplit256': (256, 64, 64, 2, (32, 64, 4)),
                              'ksplit128': (128, 32, 96, 2, (32, 32, 4)), 'stream_64_16': (64, 16, 256, 3, (128, 16, 4)),
                              'stream_16_64': (16, 64, 128, 3, (64, 64, 4)),
                              # more row blocks than the 512 workgroups of a launch: the transposed stream kernel walks (bf16-stored tensors only)
                              'stream_64_16_walk': (64, 16, 128 * 515, 3, (128, 16, 4)), 'stream_16_64_walk': (16, 64, 64 * 517, 3, (64, 64, 4))}[case]
    rng = np.random.RandomState(83)
    q = lambda v: widen(bf16_bits(np.asarray(v, np.float32))).astype('f8')                  # noqa: E731
    X16, X32 = both(rt, rng.normal(size=(M, K)) * 2 + 1)
    X = X16 if stored else X32
    xv = widen(X16.get()).reshape(M, K)
    Wk = (rng.normal(size=(N, K)) * 0.3).astype(np.float32)
    mean, scale, beta = (rng.normal(size=K).astype(np.float32) for _ in range(3))
    bias = rng.normal(size=N).astype(np.float32)
    act = ops.act(Act.BN_RELU, rt.upload(mean), rt.upload(scale), rt.upload(beta), K)
    a = np.maximum(((xv - mean).astype(np.float32).astype('f8') * scale.astype('f8') + beta.astype('f8')).astype(np.float32), 0)
    # forward
    Y = rt.alloc((M, N), zero=False)
    L = ops.gemm(rt, X, rt.upload(Wk), Y, M, N, K, 1, 1, K, K, N, actA=act, bias=rt.upload(bias), tile=tile, variant=variant, precision=1)
    if variant:
        assert ops.gemm_variant_rows(rt, L) > 0
    L(rt.stream)
    rt.synchronize()
    ref = q(a) @ q(Wk).T + bias
    np.testing.assert_allclose(Y.get(), ref, rtol=0, atol=3e-6 * np.sqrt(K) * np.abs(ref).max())
    # the float32 kernel on the same problem differs by the rounding of the operands (a guard that the bf16 path really ran)
    Y0 = rt.alloc((M, N), zero=False)
    ops.gemm(rt, X, rt.upload(Wk), Y0, M, N, K, 1, 1, K, K, N, actA=act, bias=rt.upload(bias), tile=tile, variant=variant)(rt.stream)
    rt.synchronize()
    assert np.abs(Y0.get() - ref).max() > 20 * 3e-6 * np.sqrt(K) * np.abs(ref).max()
    # data gradient of the twin layer: dX[M][N] = dY[M][K] . W2[K][N]
    W2 = (rng.normal(size=(K, N)) * 0.3).astype(np.float32)
    dY16, dY32 = both(rt, rng.normal(size=(M, K)))
    dH = rt.alloc((M, N), zero=False)
    L = ops.gemm(rt, dY16 if stored else dY32, rt.upload(W2), dH, M, N, K, 1, 0, K, N, N, tile=tile, variant=variant, precision=1)
    if variant:
        assert ops.gemm_variant_rows(rt, L) > 0
    L(rt.stream)
    rt.synchronize()
    g = widen(dY16.get()).reshape(M, K).astype('f8') @ q(W2)
    np.testing.assert_allclose(dH.get(), g, rtol=0, atol=3e-6 * np.sqrt(K) * np.abs(g).max())
    if case.endswith('_walk'):
        return                                    # (the walking cases are about the stream kernel's forward / data gradient above)
    if rt.lib.dpp_wgrad_stream_bf16_ok(N, K) and M % 64 == 0:
        # the row-stream filter gradient of the same layer on bf16 operands (dpp_wgrad_stream_bf16, the stage-1 shapes)
        G16s, G32s = both(rt, rng.normal(size=(M, N)))
        rpw = 32
        nsl = rt.lib.dpp_wgrad_stream_slices(N, K, M, rpw)
        assert nsl > 0
        parts = rt.alloc((nsl, N, K), zero=False)
        dWs = rt.alloc((N, K), zero=False)
        ops.wgrad_stream(rt, G16s if stored else G32s, N, X, K, M, rpw, parts, actX=act, precision=1)(rt.stream)
        ops.reduce_partials(rt, parts, nsl, N * K, dWs)(rt.stream)
        rt.synchronize()
        wants = widen(G16s.get()).reshape(M, N).astype('f8').T @ q(a)
        np.testing.assert_allclose(dWs.get(), wants, rtol=0, atol=3e-6 * np.sqrt(M) * np.abs(wants).max())
        dW0 = rt.alloc((N, K), zero=False)
        ops.wgrad_stream(rt, G16s if stored else G32s, N, X, K, M, rpw, parts, actX=act)(rt.stream)
        ops.reduce_partials(rt, parts, nsl, N * K, dW0)(rt.stream)
        rt.synchronize()
        assert np.abs(dW0.get() - wants).max() > 10 * 3e-6 * np.sqrt(M) * np.abs(wants).max()          # the float32 kernel does not round
    if variant == 0:
        # filter gradient dW[N][K] = sum_m G[m][N] * act(X)[m][K], reduction over the M pixel rows in 2 K-slices
        G16, G32 = both(rt, rng.normal(size=(M, N)))
        part = rt.alloc((2, N, K), zero=False)
        dW = rt.alloc((N, K), zero=False)
        ops.gemm(rt, G16 if stored else G32, X, None, N, K, M, 0, 0, N, K, K, actB=act, splitk=2, partial=part, precision=1)(rt.stream)
        ops.reduce_partials(rt, part, 2, N * K, dW)(rt.stream)
        rt.synchronize()
        want = widen(G16.get()).reshape(M, N).astype('f8').T @ q(a)
        np.testing.assert_allclose(dW.get(), want, rtol=0, atol=3e-6 * np.sqrt(M) * np.abs(want).max())
