"""Kernel parity: conv3x3 / stem / BatchNorm / loss / ADAM against the NumPy oracle (float64) on the same
seeded inputs.  Tolerances are f32 round-off bounds written next to each check.  Runs on the SIMT emulator
in the CPU tier and through the C ABI of libdpp_hip.so on the MI355X (`-m gpu`)."""
import numpy as np
import pytest

from hipdp import layout, ops
from hipdp.lib import Act
from oracle import layers as L
from tests.backends import BACKENDS, get_runtime


def up(rt, **kw):
    return {k: rt.upload(np.asarray(v, np.float32)) for k, v in kw.items()}


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(3, 8, 8, 16, 16, 64), (2, 16, 16, 32, 32, 128), (5, 4, 4, 64, 64, 64), (2, 6, 10, 16, 32, 128),
                                 (20, 2, 2, 16, 16, 64), (3, 5, 12, 64, 64, 64)])
def test_conv3x3_fwd_dgrad_wgrad(backend, cfg):
    rt = get_runtime(backend)
    N, H, W, Ci, Co, bm = cfg
    rng = np.random.RandomState(11)
    x = rng.normal(size=(N, Ci, H, W))
    Wr = rng.normal(size=(Co, Ci, 3, 3)) * 0.2
    b = rng.normal(size=Co)
    mean, scale, beta = rng.normal(size=Ci) * 0.3, rng.uniform(0.5, 1.5, Ci), rng.normal(size=Ci) * 0.3
    res = rng.normal(size=(N, Co, H, W))
    a = np.maximum((x - mean[None, :, None, None]) * scale[None, :, None, None] + beta[None, :, None, None], 0)
    y_ref = L.conv2d_fwd(a, Wr, b, (1, 1), 'half') + res
    d = up(rt, X=layout.nchw_to_nhwc(x), Wk=layout.conv_w_to_kernel(Wr), b=b, mean=mean, scale=scale, beta=beta,
           res=layout.nchw_to_nhwc(res))
    act = ops.act(Act.BN_RELU, d['mean'], d['scale'], d['beta'], Ci)
    Y = rt.alloc((N, H, W, Co), zero=False)
    ops.conv3x3(rt, d['X'], N, H, W, Ci, d['Wk'], Co, Y, actX=act, bias=d['b'], residual=d['res'], bm=bm)(rt.stream)
    rt.synchronize()
    tol = 3e-6 * np.sqrt(9 * Ci) * np.abs(y_ref).max()
    np.testing.assert_allclose(layout.nhwc_to_nchw(Y.get()), y_ref, rtol=0, atol=tol)
    # data gradient = same kernel on dY with the transposed / mirrored weights
    dy = rng.normal(size=(N, Co, H, W))
    da_ref, dW_ref, _ = L.conv2d_bwd(a, Wr, dy, (1, 1), 'half')
    dYb = rt.upload(layout.nchw_to_nhwc(dy).astype(np.float32))
    Wd = rt.alloc((Ci, 9, Co), zero=False)
    dA = rt.alloc((N, H, W, Ci), zero=False)
    ops.conv3x3_wtrans(rt, d['Wk'], Co, Ci, Wd)(rt.stream)
    ops.conv3x3(rt, dYb, N, H, W, Co, Wd, Ci, dA, bm=bm)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(layout.nhwc_to_nchw(dA.get()), da_ref, rtol=0, atol=3e-6 * np.sqrt(9 * Co) * np.abs(da_ref).max())
    # filter gradient: per-workgroup partials + fixed-order reduce
    nblk = rt.lib.dpp_conv3x3_wgrad_blocks(N, H, W, Ci, Co, bm)
    part = rt.alloc((nblk, Co, 9, Ci), zero=False)
    dWk = rt.alloc((Co, 9, Ci), zero=False)
    ops.conv3x3_wgrad(rt, d['X'], N, H, W, Ci, dYb, Co, part, actX=act, bm=bm)(rt.stream)
    ops.reduce_partials(rt, part, nblk, Co * 9 * Ci, dWk)(rt.stream)
    rt.synchronize()
    dW = layout.conv_w_from_kernel(dWk.get(), (Co, Ci, 3, 3))
    np.testing.assert_allclose(dW, dW_ref, rtol=0, atol=3e-6 * np.sqrt(N * H * W) * np.abs(dW_ref).max())
    # the same gradient on the row-streaming kernel (shapes of the bottleneck convolutions, W % 4 == 0), ragged row ranges
    for rpw in (8, 20, 256):
        nsl = rt.lib.dpp_wgrad3_stream_slices(Co, Ci, N, H, W, rpw)
        if Co != Ci or Co not in (16, 32, 64) or W % 4:
            assert nsl == 0
            continue
        assert nsl > 0
        part = rt.alloc((nsl, Co, 9, Ci), zero=False)
        ops.wgrad3_stream(rt, dYb, Co, d['X'], Ci, N, H, W, rpw, part, actX=act)(rt.stream)
        ops.reduce_partials(rt, part, nsl, Co * 9 * Ci, dWk)(rt.stream)
        rt.synchronize()
        dW = layout.conv_w_from_kernel(dWk.get(), (Co, Ci, 3, 3))
        np.testing.assert_allclose(dW, dW_ref, rtol=0, atol=3e-6 * np.sqrt(N * H * W) * np.abs(dW_ref).max())


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(1, 8, 16, 16), (3, 20, 28, 16), (2, 9, 13, 32), (12, 32, 32, 16), (6, 16, 32, 32), (34, 32, 32, 16), (3, 16, 16, 64), (2, 7, 19, 64)])
def test_conv3x3_filter_gradient_on_transposed_images(backend, cfg):
    """conv3x3_wgrad_t_kernel (round 6: channel-major LDS images, 16-byte operand reads) on the 16- / 32-channel layers: one, three and
    nine taps per workgroup (the grid wgrad_geometry picks from the tile count), ragged maps (zero padding after the activation, tiles
    that hang over the image), workgroups that walk several tiles with the next tile in flight, more workgroups than tiles; float32
    and bf16-stored operands (widened exactly: equal bits).  Against the float64 oracle (T.grad of conv2d,
    /root/reference/src/net/convlayer.py:230-240)."""
    rt = get_runtime(backend)
    N, H, W, Cc = cfg
    rng = np.random.RandomState(19)
    x = rng.normal(size=(N, Cc, H, W)) + 0.2
    mean, scale, beta = rng.normal(size=Cc) * 0.3, rng.uniform(0.5, 1.5, Cc), rng.normal(size=Cc) * 0.3
    a = np.maximum((x - mean[None, :, None, None]) * scale[None, :, None, None] + beta[None, :, None, None], 0)
    dy = rng.normal(size=(N, Cc, H, W))
    d = up(rt, X=layout.nchw_to_nhwc(x), dY=layout.nchw_to_nhwc(dy), mean=mean, scale=scale, beta=beta)
    for use_act in (True, False):
        _, dW_ref, _ = L.conv2d_bwd(a if use_act else x, np.zeros((Cc, Cc, 3, 3)), dy, (1, 1), 'half')
        act = ops.act(Act.BN_RELU, d['mean'], d['scale'], d['beta'], Cc) if use_act else None
        for bm in (64, 128):
            nblk = rt.lib.dpp_conv3x3_wgrad_blocks(N, H, W, Cc, Cc, bm)
            part = rt.alloc((nblk, Cc, 9, Cc), zero=False)
            part.set(np.full((nblk, Cc, 9, Cc), np.nan, np.float32))          # every slice element must be written
            dWk = rt.alloc((Cc, 9, Cc), zero=False)
            ops.conv3x3_wgrad(rt, d['X'], N, H, W, Cc, d['dY'], Cc, part, actX=act, bm=bm)(rt.stream)
            ops.reduce_partials(rt, part, nblk, Cc * 9 * Cc, dWk)(rt.stream)
            rt.synchronize()
            dW = layout.conv_w_from_kernel(dWk.get(), (Cc, Cc, 3, 3))
            np.testing.assert_allclose(dW, dW_ref, rtol=0, atol=3e-6 * np.sqrt(N * H * W) * np.abs(dW_ref).max())
            again = rt.alloc((nblk, Cc, 9, Cc), zero=False)
            ops.conv3x3_wgrad(rt, d['X'], N, H, W, Cc, d['dY'], Cc, again, actX=act, bm=bm)(rt.stream)
            rt.synchronize()
            assert np.array_equal(again.get(), part.get())                      # fixed summation order
    # bf16 MFMA operands (BASELINE config 5, dpp_conv3x3_wgrad_bf16): both operands rounded to bfloat16 -- act(X) AFTER the prologue --,
    # float32 accumulation: against the float64 oracle on the rounded operands; a bf16-stored dY gives the same bits as its widened twin
    assert rt.lib.dpp_conv3x3_wgrad_bf16_ok(N, H, W, Cc, Cc) == 1
    rne = lambda v: ((np.ascontiguousarray(v, np.float32).view(np.uint32).astype(np.uint64) + 0x7FFF +          # noqa: E731
                      ((np.ascontiguousarray(v, np.float32).view(np.uint32).astype(np.uint64) >> 16) & 1)) >> 16).astype(np.uint16)
    wid = lambda b: (b.astype(np.uint32) << 16).view(np.float32)                                                   # noqa: E731
    xf = layout.nchw_to_nhwc(x).astype(np.float32)
    af = np.maximum((xf - mean.astype(np.float32)) * scale.astype(np.float32) + beta.astype(np.float32), np.float32(0))      # the kernel's f32 prologue
    a_q = layout.nhwc_to_nchw(wid(rne(af)).astype(np.float64))
    dy_bits = rne(layout.nchw_to_nhwc(dy))
    dy_q = layout.nhwc_to_nchw(wid(dy_bits).astype(np.float64))
    _, dW_ref, _ = L.conv2d_bwd(a_q, np.zeros((Cc, Cc, 3, 3)), dy_q, (1, 1), 'half')
    act = ops.act(Act.BN_RELU, d['mean'], d['scale'], d['beta'], Cc)
    nblk = rt.lib.dpp_conv3x3_wgrad_blocks(N, H, W, Cc, Cc, 128)
    got = []
    for dYbuf in (rt.upload(dy_bits), rt.upload(wid(dy_bits))):
        part = rt.alloc((nblk, Cc, 9, Cc), zero=False)
        part.set(np.full((nblk, Cc, 9, Cc), np.nan, np.float32))
        dWk = rt.alloc((Cc, 9, Cc), zero=False)
        ops.conv3x3_wgrad(rt, d['X'], N, H, W, Cc, dYbuf, Cc, part, actX=act, bm=128, precision=1)(rt.stream)
        ops.reduce_partials(rt, part, nblk, Cc * 9 * Cc, dWk)(rt.stream)
        rt.synchronize()
        got.append(part.get())
        dW = layout.conv_w_from_kernel(dWk.get(), (Cc, Cc, 3, 3))
        # an activation within float32 round-off of a bf16 rounding boundary may round the other way than the float32 prologue here
        # (FMA contraction on the device): allow a few bf16 steps of one product among the N*H*W of a sum
        np.testing.assert_allclose(dW, dW_ref, rtol=0, atol=(3e-6 * np.sqrt(N * H * W) + 2e-4) * np.abs(dW_ref).max())
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(3, 16, 32, 16, 0, 0), (2, 9, 20, 16, 0, 1), (3, 16, 16, 32, 0, 0), (2, 24, 16, 32, 1, 1), (3, 16, 32, 16, 1, 1),
                                 (7, 8, 8, 64, 0, 0), (8, 8, 8, 64, 1, 1)],
                         ids=lambda c: 'x'.join(str(v) for v in c))
def test_tile_walking_conv3x3_equals_one_workgroup_per_tile(backend, cfg, monkeypatch):
    """conv3x3_p_kernel (round 6: a workgroup stages the nine weight slices once and walks tiles, the next halo in flight) against
    conv3x3_kernel on the same problem: the same images, tap loop and epilogue, so the output, the fused BatchNorm statistics and the
    BatchNorm-backward sums are BIT-identical -- forward with prologue + statistics, data gradient with the BatchNorm-backward epilogue,
    whole and ragged maps, float32 and bf16 operands, float32 and bf16-stored tensors, more tiles than workgroups."""
    from tests.test_bf16_store import both, _bn_coeffs
    rt = get_runtime(backend)
    N, H, W, Cc, prec, st16 = cfg
    monkeypatch.setenv('DPP_C3_P_64', '2')            # (the 64-channel form in float32 as well: off by default)
    rng = np.random.RandomState(41)
    x16, x32 = both(rt, rng.normal(size=(N, H, W, Cc)) + 0.2)
    X = x16 if st16 else x32
    Wk = rt.upload((rng.normal(size=(Cc, 9, Cc)) * 0.2).astype(np.float32))
    bias = rt.upload(rng.normal(size=Cc).astype(np.float32))
    mean, scale, beta = (rt.upload(v.astype(np.float32)) for v in (rng.normal(size=Cc) * 0.3, rng.uniform(0.5, 1.5, Cc), rng.normal(size=Cc) * 0.3))
    act = ops.act(Act.BN_RELU, mean, scale, beta, Cc)
    ntiles = rt.lib.dpp_conv3x3_tiling(N, H, W, 128, None, None, None)
    assert ntiles > 3
    bx16, bx32 = both(rt, rng.normal(size=(N, H, W, Cc)))
    bn = _bn_coeffs(rt, rng, Cc)
    dy16, dy32 = both(rt, rng.normal(size=(N, H, W, Cc)))
    got = {}
    for mode in ('0', '3'):
        monkeypatch.setenv('DPP_C3_PERSIST', mode)
        Y = rt.alloc((N, H, W, Cc), np.uint16 if st16 else np.float32, zero=False)
        stats = rt.alloc((ntiles, 2, Cc), zero=False)
        ops.conv3x3(rt, X, N, H, W, Cc, Wk, Cc, Y, actX=act, bias=bias, bm=128, epi=ops.epilogue(stats=stats), precision=prec)(rt.stream)
        dX = rt.alloc((N, H, W, Cc), np.uint16 if st16 else np.float32, zero=False)
        part = rt.alloc((ntiles, 2, Cc), zero=False)
        ops.conv3x3(rt, dy16 if st16 else dy32, N, H, W, Cc, Wk, Cc, dX, bm=128, precision=prec,
                    epi=ops.epilogue(bn=bn, bn_x=bx16 if st16 else bx32, bn_relu=True, bn_partial=part))(rt.stream)
        rt.synchronize()
        got[mode] = (Y.get(), stats.get(), dX.get(), part.get())
    for k, (a, b) in enumerate(zip(got['0'], got['3'])):
        if k in (0, 2) or Cc == 16 or Cc == 64:         # (64 channels: 16-column workgroups in both kernels, two whole 8 x 8 images per tile)
            assert np.array_equal(a, b)
        else:
            # 32 channels: the walk keeps the 32 columns of a tile together, the one-tile kernel splits a launch this small into 16-column
            # workgroups -- the same values summed by a different tree of threads: the per-tile sums agree to float32 round-off
            np.testing.assert_allclose(a, b, rtol=3e-6, atol=3e-6 * np.abs(a).max())
    assert np.isfinite(got['3'][1]).all() and np.abs(got['3'][0].astype(np.float64)).max() > 0


@pytest.mark.parametrize('backend', BACKENDS)
def test_batchnorm_finalize_with_an_outlier_first_block(backend):
    """the one-pass finalize takes the block means about a PIVOT (block 0's mean) and subtracts S1^2 / M -- what cancels is the
    spread of the block means about that pivot.  Block 0 a constant far-plane background block (mean 1.0, no variance) in front of blocks of
    hand pixels around -0.3 +- 0.2, and an extreme case (block 0 at +300, the rest ~N(0, 0.01)): mean and inv_std against float64."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(5)
    M, Cc, rpb = 64 * 40, 16, 64
    for first, loc, sd in ((1.0, -0.3, 0.2), (300.0, 0.0, 0.01)):
        x = (rng.normal(size=(M, Cc)) * sd + loc).astype(np.float32)
        x[:rpb] = np.float32(first)
        gamma = np.ones(Cc, np.float32)
        nb = M // rpb
        d = up(rt, x=x, gamma=gamma)
        part = rt.alloc((nb, 2, Cc), zero=False)
        mean, istd, scale = (rt.alloc(Cc, zero=False) for _ in range(3))
        ops.bn_stats_partial(rt, d['x'], M, Cc, rpb, part)(rt.stream)
        ops.bn_finalize(rt, part, nb, M, rpb, Cc, d['gamma'], 1e-4, mean, istd, scale, None, None, 0.0)(rt.stream)
        rt.synchronize()
        x64 = x.astype('f8')
        np.testing.assert_allclose(mean.get(), x64.mean(0), rtol=3e-7, atol=1e-7)
        np.testing.assert_allclose(istd.get(), 1.0 / np.sqrt(x64.var(0) + 1e-4), rtol=3e-6)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(2, 32, 32, 32), (3, 16, 48, 32), (1, 36, 20, 16), (1, 32, 16, 30), (2, 16, 16, 24)])
def test_stem_fwd_and_wgrad(backend, cfg):
    rt = get_runtime(backend)
    N, H, W, Co = cfg
    rng = np.random.RandomState(12)
    x = rng.uniform(-1, 1, size=(N, 1, H, W))
    x[:, :, :, W // 2:] = 1.0                        # constant far-plane background: whole pooling windows tie
    Wr = rng.normal(size=(Co, 1, 5, 5)) * 0.3
    b = rng.normal(size=Co)
    y_ref, cache = L.convpool_fwd(x, Wr, b, (1, 1), 'half', (2, 2), False)
    d = up(rt, X=x[:, 0], Wk=layout.conv_w_to_kernel(Wr).reshape(Co, 25), b=b)
    Y = rt.alloc((N, H // 2, W // 2, Co), zero=False)
    arg = rt.alloc((N, H // 2, W // 2, Co), np.uint8, zero=False)
    full = H % 16 == 0 and W % 16 == 0                 # the fused BatchNorm statistics need full 16x16 conv tiles
    nblk_s = N * (H // 16) * (W // 16)
    stats = rt.alloc((nblk_s, 2, Co), zero=False) if full else None
    ops.stem_fwd(rt, d['X'], N, H, W, d['Wk'], d['b'], Co, Y, arg, stats)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(layout.nhwc_to_nchw(Y.get()), y_ref, rtol=0, atol=3e-6 * 5 * np.abs(y_ref).max())
    if full:
        M = N * (H // 2) * (W // 2)
        mean, istd, scale = (rt.alloc(Co, zero=False) for _ in range(3))
        ops.bn_finalize(rt, stats, nblk_s, M, 64, Co, rt.upload(np.ones(Co, np.float32)), 1e-4, mean, istd, scale)(rt.stream)
        rt.synchronize()
        yv = Y.get().astype('f8').reshape(M, Co)
        np.testing.assert_allclose(mean.get(), yv.mean(0), rtol=0, atol=1e-6 * np.abs(yv).max())
        np.testing.assert_allclose(istd.get(), 1.0 / np.sqrt(yv.var(0) + 1e-4), rtol=1e-5)
    else:
        assert rt.lib.dpp_stem_fwd(d['X'].ptr, N, H, W, d['Wk'].ptr, d['b'].ptr, Co, Y.ptr, arg.ptr, rt.alloc((4, 2, Co)).ptr, 0, rt.stream) != 0
    # the tie mask agrees with the oracle wherever the oracle's window is either clearly decided or exactly tied
    cshape, ties_ref, _ = cache
    c = L.conv2d_fwd(x, Wr, None, (1, 1), 'half')
    cv = c.reshape(N, Co, H // 2, 2, W // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(N, Co, H // 2, W // 2, 4)
    gap = cv.max(axis=4, keepdims=True) - cv
    clear = ((gap == 0) | (gap > 1e-4)).all(axis=4)
    got_bits = layout.nhwc_to_nchw(arg.get())
    got = ((got_bits[..., None] >> np.arange(4)) & 1).astype(bool)
    assert clear.mean() > 0.9 and ties_ref.sum(axis=4).max() == 4
    assert (got[clear] == ties_ref[clear]).all()
    # filter gradient with the device's own tie mask
    dy = rng.normal(size=y_ref.shape)
    dyb = rt.upload(layout.nchw_to_nhwc(dy).astype(np.float32))
    tpb = 2
    nblk = rt.lib.dpp_stem_wgrad_blocks(N, H, W, tpb)
    part = rt.alloc((nblk, Co, 25), zero=False)
    dWk = rt.alloc((Co, 25), zero=False)
    ops.stem_wgrad(rt, d['X'], N, H, W, dyb, arg, Co, part, tpb)(rt.stream)
    ops.reduce_partials(rt, part, nblk, Co * 25, dWk)(rt.stream)
    rt.synchronize()
    _, dW_ref, _ = L.convpool_bwd(x, Wr, dy, (cshape, got, y_ref), (1, 1), 'half', (2, 2), False, need_dx=False)
    dW = layout.conv_w_from_kernel(dWk.get().reshape(Co, 25, 1), (Co, 1, 5, 5))
    np.testing.assert_allclose(dW, dW_ref, rtol=0, atol=3e-6 * np.sqrt(N * H * W / 4) * np.abs(dW_ref).max())


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(700, 16, 128), (333, 64, 64), (130, 256, 32), (1000, 32, 1000), (1024, 16, 32), (256, 64, 32)])
def test_batchnorm_forward_stats_and_backward(backend, cfg):
    rt = get_runtime(backend)
    M, Cc, rpb = cfg
    rng = np.random.RandomState(13)
    # large mean / std ratio on purpose: the shifted two-level statistics must survive it
    x = (rng.normal(size=(M, Cc)) * rng.uniform(0.05, 2.0, Cc) + rng.normal(size=Cc) * 20.0).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, Cc).astype(np.float32), (rng.normal(size=Cc) * 0.3).astype(np.float32)
    rm, ris = rng.normal(size=Cc).astype(np.float32), rng.uniform(0.5, 2, Cc).astype(np.float32)
    x4 = x.astype('f8').T.reshape(1, Cc, M, 1)
    y_ref, mean_ref, istd_ref = L.bn_fwd_train(x4, gamma.astype('f8'), beta.astype('f8'))
    d = up(rt, x=x, gamma=gamma, beta=beta, rm=rm, ris=ris)
    nb = -(-M // rpb)
    part = rt.alloc((nb, 2, Cc), zero=False)
    mean, istd, scale = (rt.alloc(Cc, zero=False) for _ in range(3))
    ops.bn_stats_partial(rt, d['x'], M, Cc, rpb, part)(rt.stream)
    ops.bn_finalize(rt, part, nb, M, rpb, Cc, d['gamma'], 1e-4, mean, istd, scale, d['rm'], d['ris'], 0.1)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(mean.get(), mean_ref, rtol=2e-7, atol=1e-7)          # f32 rounding of an f64 result
    np.testing.assert_allclose(istd.get(), istd_ref, rtol=2e-6)
    np.testing.assert_allclose(scale.get(), gamma * istd_ref, rtol=2e-6)
    rm2, ris2 = L.bn_running_update(rm, ris, mean_ref.astype(np.float32), istd_ref.astype(np.float32))
    np.testing.assert_allclose(d['rm'].get(), rm2, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(d['ris'].get(), ris2, rtol=2e-6)
    # backward with ReLU mask
    dA = rng.normal(size=(M, Cc)).astype(np.float32)
    dAb = rt.upload(dA)
    G = rt.alloc((M, Cc), zero=False)
    part2 = rt.alloc((nb, 2, Cc), zero=False)
    dbeta, dgamma, c1, c2 = (rt.alloc(Cc, zero=False) for _ in range(4))
    add = rng.normal(size=(M, Cc)).astype(np.float32)
    addb = rt.upload(add)
    dX = rt.alloc((M, Cc), zero=False)
    ops.bn_bwd_reduce(rt, dAb, d['x'], M, Cc, mean, istd, scale, d['beta'], 1, G, rpb, part2)(rt.stream)
    ops.bn_bwd_finalize(rt, part2, nb, M, Cc, dbeta, dgamma, c1, c2)(rt.stream)
    csp = rt.alloc((nb, Cc), zero=False)
    csum = rt.alloc(Cc, zero=False)
    ops.bn_bwd_apply(rt, G, d['x'], M, Cc, mean, istd, scale, c1, c2, dX, add=addb, rpb=rpb, colsum=csp)(rt.stream)
    jobs = ops.ReduceJobs(rt)
    jobs.add(csp, nb, Cc, csum)
    jobs.add(part2, nb, 2 * Cc, rt.alloc(2 * Cc))       # a second, unrelated job
    jobs.launch()(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(csum.get(), dX.get().astype('f8').sum(0), rtol=0, atol=3e-6 * np.sqrt(M) * np.abs(dX.get()).max())
    v = y_ref[0, :, :, 0].T                           # bn output (M, C)
    # the mask is decided on the device's own f32 bn value; exclude elements within f32 noise of zero
    safe = np.abs(v) > 1e-4
    g_ref = dA.astype('f8') * (v >= 0)
    assert (G.get()[safe] == g_ref[safe].astype(np.float32)).all()
    gdev = G.get().astype('f8')
    dx_ref, dgamma_ref, dbeta_ref = L.bn_bwd_train(x4, gamma.astype('f8'), mean_ref, istd_ref, gdev.T.reshape(1, Cc, M, 1))
    np.testing.assert_allclose(dbeta.get(), dbeta_ref, rtol=0, atol=2e-6 * np.sqrt(M) * np.abs(gdev).max())
    np.testing.assert_allclose(dgamma.get(), dgamma_ref, rtol=0, atol=2e-5 * np.sqrt(M) * np.abs(gdev).max())
    np.testing.assert_allclose(dX.get(), dx_ref[0, :, :, 0].T + add, rtol=0, atol=2e-5 * np.abs(dx_ref).max())
    # deterministic-mode coefficients
    ops.bn_eval_coeffs(rt, d['gamma'], d['rm'], d['ris'], Cc, mean, istd, scale)(rt.stream)
    rt.synchronize()
    np.testing.assert_array_equal(mean.get(), d['rm'].get())
    np.testing.assert_allclose(scale.get(), gamma * d['ris'].get(), rtol=1e-7)


@pytest.mark.parametrize('backend', BACKENDS)
def test_loss_colsum_adam_elementwise(backend):
    rt = get_runtime(backend)
    rng = np.random.RandomState(14)
    B, D = 128, 30
    out, y = rng.normal(size=(B, D)).astype(np.float32), rng.normal(size=(B, D)).astype(np.float32)
    ob, yb = rt.upload(out), rt.upload(y)
    cost, err, dout = rt.alloc(1), rt.alloc(2), rt.alloc((B, D), zero=False)
    ops.loss_sse(rt, ob, yb, B, D, B, cost, dout)(rt.stream)
    ops.error_l2(rt, ob, yb, B, D, err)(rt.stream)
    rt.synchronize()
    c_ref, d_ref = L.loss_embedding(out.astype('f8'), y.astype('f8'))
    np.testing.assert_allclose(cost.get()[0], c_ref, rtol=1e-6)
    np.testing.assert_allclose(dout.get(), d_ref, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(err.get()[0], L.error_embedding(out.astype('f8'), y.astype('f8')), rtol=1e-6)
    np.testing.assert_allclose(err.get()[1], np.sqrt(((out.astype('f8') - y) ** 2).sum(1)).max(), rtol=1e-6)
    # column sums (bias gradients)
    for (M, Cc, rpb) in ((1000, 64, 128), (77, 30, 16), (40, 1024, 8)):
        X = rng.normal(size=(M, Cc)).astype(np.float32)
        nb = -(-M // rpb)
        part, s = rt.alloc((nb, Cc), zero=False), rt.alloc(Cc, zero=False)
        ops.colsum_partial(rt, rt.upload(X), M, Cc, rpb, part)(rt.stream)
        ops.reduce_partials(rt, part, nb, Cc, s)(rt.stream)
        rt.synchronize()
        np.testing.assert_allclose(s.get(), X.astype('f8').sum(0), rtol=0, atol=2e-6 * np.sqrt(M) * 4)
    # ADAM: three steps against the oracle's float32 restatement
    n = 1003
    w = rng.normal(size=n).astype(np.float32)
    wb, m, v = rt.upload(w), rt.alloc(n), rt.alloc(n)
    P, Mo, Vo, t = [w.copy()], [np.zeros(n, np.float32)], [np.zeros(n, np.float32)], 1.0
    lr = np.float32(1e-3)
    hyper = rt.upload(np.array([lr, 1.0, 0.9, 0.999, 1e-8, 1 - 1e-8, 0, 0], np.float32))
    for step in range(3):
        g = (rng.normal(size=n) * 10 ** rng.uniform(-6, 0, n)).astype(np.float32)
        ops.adam(rt, wb, rt.upload(g), m, v, n, hyper)(rt.stream)
        ops.adam_tick(rt, hyper)(rt.stream)
        rt.synchronize()
        assert hyper.get()[1] == step + 2
        m_prev = Mo[0].copy()
        t = L.adam_step(P, [g], Mo, Vo, t, lr)
        np.testing.assert_allclose(wb.get(), P[0], rtol=2e-6, atol=1e-9)
        # b1*m + (1-b1)*g may cancel: the round-off bound is relative to the terms, not to the result
        assert (np.abs(m.get() - Mo[0]) <= 2e-7 * (np.abs(m_prev) + np.abs(g)) + 1e-30).all()
        np.testing.assert_allclose(v.get(), Vo[0], rtol=2e-6, atol=1e-20)
    # RMSProp (optimizer.py:92-116) through the same kernel (hyper slot 6 selects the rule; slots 3 / 4 = decay / epsilon)
    w = rng.normal(size=n).astype(np.float32)
    wb, msg = rt.upload(w), rt.alloc(n)
    hyper = rt.upload(np.array([1e-2, 1.0, 0, 0.9, 1.0 / 100., 0, 1, 0], np.float32))
    ref_w, ref_ms = w.copy(), np.zeros(n, np.float32)
    for step in range(3):
        g = (rng.normal(size=n) * 10 ** rng.uniform(-4, 0, n)).astype(np.float32)
        ops.adam(rt, wb, rt.upload(g), m, msg, n, hyper)(rt.stream)
        rt.synchronize()
        ref_ms = (np.float32(0.9) * ref_ms + np.float32(1 - np.float32(0.9)) * (g * g)).astype(np.float32)
        ref_w = (ref_w + (-np.float32(1e-2) * g) / np.maximum(np.sqrt(ref_ms), np.float32(1.0 / 100.))).astype(np.float32)
        np.testing.assert_allclose(msg.get(), ref_ms, rtol=2e-6, atol=1e-20)
        np.testing.assert_allclose(wb.get(), ref_w, rtol=2e-6, atol=1e-7)
    # dropout forward (deterministic + mask) and relu backward
    pre = rng.normal(size=500).astype(np.float32)
    mask = (rng.uniform(size=500) < 0.7).astype(np.float32)
    pb, mb, o1, o2, g1 = rt.upload(pre), rt.upload(mask), rt.alloc(500), rt.alloc(500), rt.alloc(500)
    ops.scale(rt, pb, o1, 500, a=np.float32(0.7), relu=True)(rt.stream)
    ops.scale(rt, pb, o2, 500, relu=True, mask=mb)(rt.stream)
    ops.relu_bwd(rt, pb, pb, g1, 500, mask=mb)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(o1.get(), np.float32(0.7) * np.maximum(pre, 0), rtol=1e-7)
    np.testing.assert_array_equal(o2.get(), mask * np.maximum(pre, 0))
    np.testing.assert_array_equal(g1.get(), pre * mask * (pre >= 0))


class _BN(object):
    pass


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('kind', ['gemm', 'conv3x3'])
def test_fused_epilogues_stats_and_bn_backward(backend, kind):
    """Fused epilogues: (a) per-block BatchNorm statistics of the tensor a conv writes == a separate statistics pass;
    (b) the data-gradient epilogue's ReLU mask + (sum G, sum G*xhat) == dpp_bn_bwd_reduce."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(31)
    if kind == 'gemm':
        N, H, W, Ci, Co, bm = 3, 10, 10, 32, 48, 64          # 300 rows: 5 row blocks, the last one partial
    else:
        N, H, W, Ci, Co, bm = 3, 8, 8, 16, 32, 64
    M = N * H * W
    x = (rng.normal(size=(M, Ci)) + 3.0).astype(np.float32)
    res = rng.normal(size=(M, Co)).astype(np.float32)
    bias = rng.normal(size=Co).astype(np.float32)
    xb, resb, bb = rt.upload(x), rt.upload(res), rt.upload(bias)
    Y = rt.alloc((M, Co), zero=False)
    if kind == 'gemm':
        Wk = (rng.normal(size=(Co, Ci)) * 0.2).astype(np.float32)
        nblk = -(-M // bm)
        stats = rt.alloc((nblk, 2, Co), zero=False)
        ops.gemm(rt, xb, rt.upload(Wk), Y, M, Co, Ci, 1, 1, Ci, Ci, Co, bias=bb, residual=resb, tile=(bm, 16, 4),
                 epi=ops.epilogue(stats=stats))(rt.stream)
        y_ref = x.astype('f8') @ Wk.astype('f8').T + bias + res
        rpb = bm
    else:
        Wr = (rng.normal(size=(Co, Ci, 3, 3)) * 0.2).astype(np.float32)
        nblk = rt.lib.dpp_conv3x3_tiling(N, H, W, bm, None, None, None)
        stats = rt.alloc((nblk, 2, Co), zero=False)
        ops.conv3x3(rt, xb, N, H, W, Ci, rt.upload(layout.conv_w_to_kernel(Wr)), Co, Y, bias=bb, residual=resb, bm=bm,
                    epi=ops.epilogue(stats=stats))(rt.stream)
        xn = layout.nhwc_to_nchw(x.reshape(N, H, W, Ci).astype('f8'))
        y_ref = layout.nchw_to_nhwc(L.conv2d_fwd(xn, Wr.astype('f8'), bias.astype('f8'), (1, 1), 'half')).reshape(M, Co) + res
        rpb = bm
    gamma = rng.uniform(0.5, 1.5, Co).astype(np.float32)
    mean, istd, scale = (rt.alloc(Co, zero=False) for _ in range(3))
    ops.bn_finalize(rt, stats, nblk, M, rpb, Co, rt.upload(gamma), 1e-4, mean, istd, scale)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(Y.get(), y_ref, rtol=0, atol=3e-5 * np.abs(y_ref).max())
    np.testing.assert_allclose(mean.get(), y_ref.mean(0), rtol=0, atol=3e-6 * np.abs(y_ref).max())
    np.testing.assert_allclose(istd.get(), 1 / np.sqrt(y_ref.var(0) + np.float32(1e-4)), rtol=3e-5)

    # (b) data gradient with the BatchNorm-backward fusion: out channels = Ci of the forward conv
    dy = rng.normal(size=(M, Co)).astype(np.float32)
    dyb = rt.upload(dy)
    bn = _BN()
    mu, sg, be = x.mean(0).astype(np.float32), rng.uniform(0.5, 1.5, Ci).astype(np.float32), (rng.normal(size=Ci) * 0.5).astype(np.float32)
    isd = (1 / np.sqrt(x.var(0) + 1e-4)).astype(np.float32)
    bn.mean, bn.inv_std, bn.scale, bn.beta_buf = rt.upload(mu), rt.upload(isd), rt.upload(sg * isd), rt.upload(be)
    G = rt.alloc((M, Ci), zero=False)
    if kind == 'gemm':
        nb2 = -(-M // bm)
        part = rt.alloc((nb2, 2, Ci), zero=False)
        ops.gemm(rt, dyb, rt.upload(Wk), G, M, Ci, Co, 1, 0, Co, Ci, Ci, tile=(bm, 32, 4),
                 epi=ops.epilogue(bn=bn, bn_x=xb, bn_relu=True, bn_partial=part))(rt.stream)
        da_ref = dy.astype('f8') @ Wk.astype('f8')
    else:
        nb2 = nblk
        part = rt.alloc((nb2, 2, Ci), zero=False)
        Wd = rt.alloc(Ci * 9 * Co, zero=False)
        ops.conv3x3_wtrans(rt, rt.upload(layout.conv_w_to_kernel(Wr)), Co, Ci, Wd)(rt.stream)
        ops.conv3x3(rt, dyb, N, H, W, Co, Wd, Ci, G, bm=bm, epi=ops.epilogue(bn=bn, bn_x=xb, bn_relu=True, bn_partial=part))(rt.stream)
        dyn = layout.nhwc_to_nchw(dy.reshape(N, H, W, Co).astype('f8'))
        da_ref = layout.nchw_to_nhwc(L.conv2d_bwd(xn, Wr.astype('f8'), dyn, (1, 1), 'half')[0]).reshape(M, Ci)
    dbeta, dgamma, c1, c2 = (rt.alloc(Ci, zero=False) for _ in range(4))
    ops.bn_bwd_finalize(rt, part, nb2, M, Ci, dbeta, dgamma, c1, c2)(rt.stream)
    rt.synchronize()
    v = (x.astype('f8') - mu) * (sg * isd) + be
    safe = np.abs(v) > 1e-4
    g_ref = da_ref * (v >= 0)
    got = G.get()
    np.testing.assert_allclose(got[safe], g_ref[safe], rtol=0, atol=3e-5 * np.abs(da_ref).max())
    xhat = (x.astype('f8') - mu) * isd
    np.testing.assert_allclose(dbeta.get(), got.astype('f8').sum(0), rtol=0, atol=3e-5 * np.sqrt(M) * np.abs(got).max())
    np.testing.assert_allclose(dgamma.get(), (got.astype('f8') * xhat).sum(0), rtol=0, atol=3e-4 * np.sqrt(M) * np.abs(got).max())


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [
    # N, H, W, Ci, Co, k, border, pool, relu_in
    (2, 44, 44, 1, 8, 5, 'valid', 4, False),      # PoseRegNet conv-pool 1 (poseregnet.py:62-66), reduced map
    (3, 31, 31, 8, 8, 5, 'valid', 2, True),       # conv-pool 2: odd conv map 27 -> pool ignores the border
    (3, 13, 13, 8, 8, 3, 'valid', 1, True),       # conv-pool 3: no pooling (poolType -1)
    (2, 20, 18, 3, 12, 3, 'half', 3, True),       # odd sizes: 'half' padding, pool 3, channel counts off the wave grid
    (2, 40, 36, 8, 8, 5, 'valid', 2, True),       # several 32 x 32 input tiles with ragged edges (register-patch kernels)
    (2, 14, 14, 8, 8, 5, 'valid', 1, True),       # ScaleNet's third tower: 5x5 without pooling on a small map
    (2, 30, 22, 12, 8, 5, 'half', 2, True),       # 'half' padding and two groups of input channels on the register-patch kernels
])
def test_convpool_fwd_wgrad_dgrad(backend, cfg):
    """Generic ConvPoolLayer kernels against the oracle, incl. Theano's gradient-to-every-tied-maximum rule on a constant
    background region."""
    rt = get_runtime(backend)
    N, H, W, Ci, Co, k, border, pool, relu_in = cfg
    pad = k // 2 if border == 'half' else 0
    rng = np.random.RandomState(31)
    pre = rng.uniform(-1, 1, size=(N, Ci, H, W))
    pre[:, :, :, W // 2:] = 0.75                       # constant region: conv output ties inside whole pooling windows
    x = np.maximum(pre, 0) if relu_in else pre         # the previous layer's ReLU is this layer's operand prologue
    Wr = rng.normal(size=(Co, Ci, k, k)) * 0.3
    b = rng.normal(size=Co)
    y_ref, cache = L.convpool_fwd(x, Wr, b, (1, 1), border, (pool, pool), False)
    cshape, ties_ref, _ = cache
    Hp, Wp = y_ref.shape[2:]
    d = up(rt, X=layout.nchw_to_nhwc(pre), Wk=layout.conv_w_to_kernel(Wr), b=b)
    act = ops.act(mode=1) if relu_in else None
    Y = rt.alloc((N, Hp, Wp, Co), zero=False)
    ties = rt.alloc((N, Hp, Wp, Co), np.uint16, zero=False) if pool > 1 else None
    ops.convpool_fwd(rt, d['X'], N, H, W, Ci, d['Wk'], k, k, pad, Co, pool, d['b'], Y, ties, actX=act)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(layout.nhwc_to_nchw(Y.get()), y_ref, rtol=0, atol=3e-6 * np.sqrt(k * k * Ci) * np.abs(y_ref).max())
    if pool > 1:
        c = L.conv2d_fwd(x, Wr, None, (1, 1), border)
        cv = c[:, :, :Hp * pool, :Wp * pool].reshape(N, Co, Hp, pool, Wp, pool).transpose(0, 1, 2, 4, 3, 5).reshape(N, Co, Hp, Wp, pool * pool)
        gap = cv.max(axis=4, keepdims=True) - cv
        clear = ((gap == 0) | (gap > 1e-4)).all(axis=4)
        got = ((layout.nhwc_to_nchw(ties.get())[..., None] >> np.arange(pool * pool)) & 1).astype(bool)
        assert clear.mean() > 0.9 and ties_ref.sum(axis=4).max() == pool * pool
        assert (got[clear] == ties_ref[clear]).all()
    else:
        got = None
    # gradients with the device's own tie masks
    dy = rng.normal(size=y_ref.shape)
    dyb = rt.upload(layout.nchw_to_nhwc(dy).astype(np.float32))
    nblk = rt.lib.dpp_convpool_wgrad_blocks(N, Hp, Wp)
    nW = Co * k * k * Ci
    part = rt.alloc((nblk, nW), zero=False)
    dWk = rt.alloc(nW, zero=False)
    dX = rt.alloc((N, H, W, Ci), zero=False)
    ops.convpool_wgrad(rt, d['X'], N, H, W, Ci, dyb, ties, k, k, pad, Co, pool, part, actX=act)(rt.stream)
    ops.reduce_partials(rt, part, nblk, nW, dWk)(rt.stream)
    ops.convpool_dgrad(rt, dyb, ties, N, H, W, Ci, d['Wk'], k, k, pad, Co, pool, dX)(rt.stream)
    rt.synchronize()
    dx_ref, dW_ref, _ = L.convpool_bwd(x, Wr, dy, (cshape, got, y_ref), (1, 1), border, (pool, pool), False)
    dW = layout.conv_w_from_kernel(dWk.get().reshape(Co, k * k, Ci), (Co, Ci, k, k))
    np.testing.assert_allclose(dW, dW_ref, rtol=0, atol=3e-6 * np.sqrt(N * Hp * Wp) * np.abs(dW_ref).max())
    np.testing.assert_allclose(layout.nhwc_to_nchw(dX.get()), dx_ref, rtol=0, atol=3e-6 * np.sqrt(k * k * Co) * np.abs(dx_ref).max())


@pytest.mark.parametrize('backend', BACKENDS)
def test_bernoulli_mask_streams(backend):
    """Dropout masks (dropoutlayer.py:98-103): Bernoulli(keep), reproducible per (seed, counter), advanced by the
    device-resident step counter so that a recorded launch draws a new mask every step."""
    rt = get_runtime(backend)
    n, keep = 1 << 16, 0.7
    m = [rt.alloc(n, zero=False) for _ in range(4)]
    ctr = rt.alloc(1, np.int64)
    ops.bernoulli_mask(rt, m[0], n, keep, 1234, 5)(rt.stream)
    ops.bernoulli_mask(rt, m[1], n, keep, 1234, 5, ctr)(rt.stream)          # counter_dev = 0: same stream
    ops.counter_add(rt, ctr, 1)(rt.stream)
    ops.bernoulli_mask(rt, m[2], n, keep, 1234, 5, ctr)(rt.stream)          # = host counter 6
    ops.bernoulli_mask(rt, m[3], n, keep, 1234, 6)(rt.stream)
    rt.synchronize()
    a, b, c, d = (x.get() for x in m)
    assert set(np.unique(a)) == {0.0, 1.0}
    assert abs(a.mean() - keep) < 4 * np.sqrt(keep * (1 - keep) / n)
    assert np.array_equal(a, b) and np.array_equal(c, d) and (a != c).mean() > 0.3


@pytest.mark.parametrize('backend', BACKENDS)
def test_reduce_multi_ragged_jobs(backend):
    """One launch sums every job's slices: vector path (n % 4 == 0, aligned), scalar path (ragged n, outputs at odd offsets),
    slice counts around the 4 x 16 unroll."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(5)
    cases = [(1, 16), (5, 30), (37, 100), (20, 1028), (129, 63), (256, 64), (3, 1), (70, 260)]
    flat = rt.alloc(sum(n for _, n in cases) + 3, zero=True)
    jobs, refs, off = ops.ReduceJobs(rt), [], 3          # outputs start 12 bytes into the buffer
    for nz, n in cases:
        P = rng.normal(size=(nz, n)).astype(np.float32)
        jobs.add(rt.upload(P), nz, n, flat.view(off, (n,)))
        refs.append((off, n, P.astype('f8').sum(0)))
        off += n
    jobs.launch()(rt.stream)
    rt.synchronize()
    got = flat.get()
    assert np.all(got[:3] == 0)
    for off, n, ref in refs:
        np.testing.assert_allclose(got[off:off + n], ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('a_kc', [1, 0])
def test_gemm_operand_through_batchnorm_backward(backend, a_kc):
    """dpp_gemm's mode-4 A operand: the gradient through a BatchNorm's batch statistics is formed from (G, x) while the tile is
    staged -- same product as materialising it with dpp_bn_bwd_apply first.  a_kc=1: data gradient (operand [pixels][C],
    reduction over channels); a_kc=0: filter gradient (reduction over pixels), ragged sizes."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(77)
    Mp, Cc, Co = 75, 24, 20                       # pixels, BatchNorm channels, the other GEMM dimension
    G = rng.normal(size=(Mp, Cc)).astype(np.float32)
    X = (rng.normal(size=(Mp, Cc)) * 2 + 5).astype(np.float32)
    mean = X.mean(0).astype(np.float32)
    istd = (1 / np.sqrt(X.var(0) + 1e-4)).astype(np.float32)
    scale = (rng.uniform(0.5, 1.5, Cc) * istd).astype(np.float32)
    c1, c2 = rng.normal(size=Cc).astype(np.float32) * 0.1, rng.normal(size=Cc).astype(np.float32) * 0.1
    bn = _BN()
    bn.mean, bn.inv_std, bn.scale = rt.upload(mean), rt.upload(istd), rt.upload(scale)
    q, p = rt.upload(scale * c1), rt.upload(scale * istd * c2)
    Gb, Xb = rt.upload(G), rt.upload(X)
    dX = scale.astype('f8') * (G.astype('f8') - c1 - (X.astype('f8') - mean) * istd * c2)
    act = ops.act_bn_bwd(bn, q, p, Xb, Cc)
    if a_kc:
        Co = 100                                                     # two column blocks: the operand copy is written once
        Wm = rng.normal(size=(Cc, Co)).astype(np.float32)            # C[m][o] = sum_c dX[m][c] W[c][o]
        out = rt.alloc((Mp, Co), zero=False)
        copy = rt.alloc((Mp, Cc), zero=True)
        act = ops.act_bn_bwd(bn, q, p, Xb, Cc, out=copy)             # ... and leaves the operand it formed in `copy`
        ops.gemm(rt, Gb, rt.upload(Wm), out, Mp, Co, Cc, 1, 0, Cc, Co, Co, actA=act)(rt.stream)
        ref = dX @ Wm.astype('f8')
        rt.synchronize()
        np.testing.assert_allclose(copy.get(), dX, rtol=0, atol=2e-6 * np.abs(dX).max())
    else:
        Y = rng.normal(size=(Mp, Co)).astype(np.float32)             # C[c][o] = sum_m dX[m][c] Y[m][o]
        out = rt.alloc((Cc, Co), zero=False)
        ops.gemm(rt, Gb, rt.upload(Y), out, Cc, Co, Mp, 0, 0, Cc, Co, Co, actA=act)(rt.stream)
        ref = dX.T @ Y.astype('f8')
    rt.synchronize()
    np.testing.assert_allclose(out.get(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(64, 256, 96, 32), (32, 128, 128, 64), (16, 64, 256, 128)])
def test_expand_kernel_operand_through_batchnorm_backward(backend, cfg):
    """The mode-4 operand on dpp_gemm variant 4 (the data gradient of a bottleneck entry): dX = scale*G - p*(x - mean) - q formed in
    registers, the product accumulated onto an earlier share with the BatchNorm-backward epilogue of the block input, the operand left
    in `out` for the filter gradient -- against float64, and bit for bit against the LDS-tiled kernel's operand copy."""
    rt = get_runtime(backend)
    K, N, M, rpw = cfg                       # K = channels of the BatchNorm the operand goes through, N = channels of the conv input
    rng = np.random.RandomState(78)
    G = rng.normal(size=(M, K)).astype(np.float32)
    X = (rng.normal(size=(M, K)) * 2 + 5).astype(np.float32)
    mean = X.mean(0).astype(np.float32)
    istd = (1 / np.sqrt(X.var(0) + 1e-4)).astype(np.float32)
    scale = (rng.uniform(0.5, 1.5, K) * istd).astype(np.float32)
    c1, c2 = rng.normal(size=K).astype(np.float32) * 0.1, rng.normal(size=K).astype(np.float32) * 0.1
    bn = _BN()
    bn.mean, bn.inv_std, bn.scale = rt.upload(mean), rt.upload(istd), rt.upload(scale)
    q, p = rt.upload(scale * c1), rt.upload(scale * istd * c2)
    Gb, Xb = rt.upload(G), rt.upload(X)
    dX = scale.astype('f8') * (G.astype('f8') - c1 - (X.astype('f8') - mean) * istd * c2)
    Wm = (rng.normal(size=(K, N)) * 0.3).astype(np.float32)
    share = rng.normal(size=(M, N)).astype(np.float32)
    bnx = rng.normal(size=(M, N)).astype(np.float32)
    eb = _BN()
    bm_, bs_, bb_, bi_ = (rng.normal(0, 0.3, N).astype('float32'), rng.uniform(0.5, 1.5, N).astype('float32'), rng.normal(0, 0.3, N).astype('float32'),
                          rng.uniform(0.5, 1.5, N).astype('float32'))
    eb.mean, eb.scale, eb.beta_buf, eb.inv_std = rt.upload(bm_), rt.upload(bs_), rt.upload(bb_), rt.upload(bi_)
    got = {}
    for variant, tile in ((4, (rpw, 64, 4)), (0, (32, 64, 1))):
        copy = rt.alloc((M, K), zero=True)
        out = rt.upload(share)
        nb = M // tile[0]
        part = rt.alloc((nb, 2, N), zero=False)
        L = ops.gemm(rt, Gb, rt.upload(Wm), out, M, N, K, 1, 0, K, N, N, actA=ops.act_bn_bwd(bn, q, p, Xb, K, out=copy), residual=out, tile=tile,
                     variant=variant, epi=ops.epilogue(bn=eb, bn_x=rt.upload(bnx), bn_relu=True, bn_partial=part))
        if variant == 4:
            assert ops.gemm_variant_rows(rt, L) == rpw
        L(rt.stream)
        rt.synchronize()
        got[variant] = (out.get(), copy.get(), part.get().reshape(2, N, nb).sum(axis=2))
    keep = ((bnx.astype('f8') - bm_) * bs_ + bb_) >= 0
    ref = np.where(keep, share.astype('f8') + dX @ Wm.astype('f8'), 0.0)
    np.testing.assert_allclose(got[4][1], dX, rtol=0, atol=2e-6 * np.abs(dX).max())
    assert np.array_equal(got[4][1], got[0][1])                        # the same expression, operation for operation
    np.testing.assert_allclose(got[4][0], ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    np.testing.assert_allclose(got[4][2][0], ref.sum(0), rtol=0, atol=2e-5 * np.abs(ref).sum(0).max())
    # without the epilogue and without a copy
    out = rt.alloc((M, N), zero=False)
    ops.gemm(rt, Gb, rt.upload(Wm), out, M, N, K, 1, 0, K, N, N, actA=ops.act_bn_bwd(bn, q, p, Xb, K), tile=(rpw, 64, 4), variant=4)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(out.get(), dX @ Wm.astype('f8'), rtol=0, atol=2e-5 * np.abs(ref).max())


def _bf16(a):
    """Round-to-nearest-even to bfloat16 (returned as float64): what the bf16 kernels do to an operand when they stage it."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(2, 8, 8, 64, 64, 64), (3, 16, 16, 16, 16, 64), (2, 12, 20, 32, 32, 128), (1, 8, 8, 64, 32, 64)])
def test_conv3x3_bf16_operands(backend, cfg):
    """dpp_conv3x3_bf16 (BASELINE config 5): forward with the BN+ReLU prologue, bias and residual, and the data gradient, against
    the float64 convolution of the bf16-ROUNDED activated input and weights -- the rounding is the whole difference to the f32
    kernel, accumulation stays f32, so the f32 tolerance applies; the 16-channel case zero-fills half of each 32-deep MFMA step."""
    rt = get_runtime(backend)
    N, H, W, Ci, Co, bm = cfg
    rng = np.random.RandomState(13)
    x = rng.normal(size=(N, Ci, H, W))
    Wr = rng.normal(size=(Co, Ci, 3, 3)) * 0.2
    b = rng.normal(size=Co)
    mean, scale, beta = rng.normal(size=Ci) * 0.3, rng.uniform(0.5, 1.5, Ci), rng.normal(size=Ci) * 0.3
    res = rng.normal(size=(N, Co, H, W))
    d = up(rt, X=layout.nchw_to_nhwc(x), Wk=layout.conv_w_to_kernel(Wr), b=b, mean=mean, scale=scale, beta=beta, res=layout.nchw_to_nhwc(res))
    f = np.float32
    a32 = np.maximum((x.astype(f) - mean.astype(f)[None, :, None, None]) * scale.astype(f)[None, :, None, None] + beta.astype(f)[None, :, None, None], 0)
    y_ref = L.conv2d_fwd(_bf16(a32), _bf16(Wr), b.astype(f).astype('f8'), (1, 1), 'half') + res.astype(f)
    act = ops.act(Act.BN_RELU, d['mean'], d['scale'], d['beta'], Ci)
    Y = rt.alloc((N, H, W, Co), zero=False)
    ops.conv3x3(rt, d['X'], N, H, W, Ci, d['Wk'], Co, Y, actX=act, bias=d['b'], residual=d['res'], bm=bm, precision=1)(rt.stream)
    rt.synchronize()
    got = layout.nhwc_to_nchw(Y.get())
    np.testing.assert_allclose(got, y_ref, rtol=0, atol=3e-6 * np.sqrt(9 * Ci) * np.abs(y_ref).max())
    # and it IS a different result from the exact f32 kernel (the rounding is visible, ~2^-9 relative)
    y32 = L.conv2d_fwd(a32.astype('f8'), Wr.astype(f).astype('f8'), b.astype(f).astype('f8'), (1, 1), 'half') + res.astype(f)
    rel = np.abs(got - y32).max() / np.abs(y32).max()
    assert 1e-5 < rel < 2e-2, rel
    dy = rng.normal(size=(N, Co, H, W))
    da_ref, _, _ = L.conv2d_bwd(np.zeros((N, Ci, H, W)), _bf16(Wr), _bf16(dy), (1, 1), 'half')
    dYb = rt.upload(layout.nchw_to_nhwc(dy).astype(np.float32))
    Wd = rt.alloc((Ci, 9, Co), zero=False)
    dA = rt.alloc((N, H, W, Ci), zero=False)
    ops.conv3x3_wtrans(rt, d['Wk'], Co, Ci, Wd)(rt.stream)
    ops.conv3x3(rt, dYb, N, H, W, Co, Wd, Ci, dA, bm=bm, precision=1)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(layout.nhwc_to_nchw(dA.get()), da_ref, rtol=0, atol=3e-6 * np.sqrt(9 * Co) * np.abs(da_ref).max())


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(2, 8, 16, 16), (1, 6, 32, 16), (8, 32, 32, 16), (4, 3, 16, 16), (4, 16, 16, 32), (3, 8, 16, 16)])
def test_conv3x3_stream_forward_statistics_and_fused_data_gradient(backend, cfg):
    """dpp_conv3x3_stream (the barrier-free kernel of the narrow 3x3 layers): forward with the BatchNorm + ReLU prologue and the
    statistics of the written tensor, and the data gradient with the BatchNorm-backward mask and sums, against float64."""
    rt = get_runtime(backend)
    N, H, W, Cc = cfg
    M = N * H * W
    rows = rt.lib.dpp_conv3x3_stream_rows(N, H, W, Cc)
    rng = np.random.RandomState(71)
    X = rt.upload(rng.normal(size=(N, H, W, Cc)).astype(np.float32))
    Y = rt.alloc((N, H, W, Cc), zero=False)
    Wr = (rng.normal(size=(Cc, Cc, 3, 3)) * 0.2).astype(np.float32)
    Wk = rt.upload(layout.conv_w_to_kernel(Wr))
    if M % 64:
        assert rows == 0
        assert rt.lib.dpp_conv3x3_stream(X.ptr, N, H, W, Cc, None, Wk.ptr, None, Y.ptr, None, rt.stream) != 0
        return
    assert rows in (64, 128) and M % rows == 0
    x = X.get().astype('f8')
    bias = rng.normal(size=Cc).astype(np.float32)
    mean, scale, beta = (rng.normal(size=Cc) * 0.3).astype(np.float32), rng.uniform(0.5, 1.5, Cc).astype(np.float32), (rng.normal(size=Cc) * 0.3).astype(np.float32)
    act = ops.act(Act.BN_RELU, rt.upload(mean), rt.upload(scale), rt.upload(beta), Cc)
    nblk = M // rows
    stats = rt.alloc((nblk, 2, Cc), zero=False)
    ops.conv3x3_stream(rt, X, N, H, W, Cc, Wk, Y, actX=act, bias=rt.upload(bias), epi=ops.epilogue(stats=stats))(rt.stream)
    a = np.maximum((x - mean) * scale + beta, 0)
    y_ref = layout.nchw_to_nhwc(L.conv2d_fwd(layout.nhwc_to_nchw(a), Wr.astype('f8'), bias.astype('f8'), (1, 1), 'half'))
    gamma = rng.uniform(0.5, 1.5, Cc).astype(np.float32)
    mo, io, so = (rt.alloc(Cc, zero=False) for _ in range(3))
    ops.bn_finalize(rt, stats, nblk, M, rows, Cc, rt.upload(gamma), 1e-4, mo, io, so)(rt.stream)
    rt.synchronize()
    np.testing.assert_allclose(Y.get(), y_ref, rtol=0, atol=3e-6 * np.sqrt(9 * Cc) * np.abs(y_ref).max())
    yv = Y.get().astype('f8').reshape(M, Cc)
    np.testing.assert_allclose(mo.get(), yv.mean(0), rtol=0, atol=1e-6 * np.abs(yv).max())
    np.testing.assert_allclose(io.get(), 1 / np.sqrt(yv.var(0) + np.float32(1e-4)), rtol=1e-5)
    # plain call (no prologue, no bias, no epilogue) == the LDS-tiled kernel's contract
    ops.conv3x3_stream(rt, X, N, H, W, Cc, Wk, Y)(rt.stream)
    rt.synchronize()
    y0 = layout.nchw_to_nhwc(L.conv2d_fwd(layout.nhwc_to_nchw(x), Wr.astype('f8'), None, (1, 1), 'half'))
    np.testing.assert_allclose(Y.get(), y0, rtol=0, atol=3e-6 * np.sqrt(9 * Cc) * np.abs(y0).max())

    # data gradient: the same kernel on dY with the mirrored weights, BatchNorm-backward epilogue
    dy = rng.normal(size=(N, H, W, Cc)).astype(np.float32)
    xs = (x + 1.0).astype(np.float32).reshape(M, Cc)
    bn = _BN()
    mu, sg, be = xs.mean(0).astype(np.float32), rng.uniform(0.5, 1.5, Cc).astype(np.float32), (rng.normal(size=Cc) * 0.5).astype(np.float32)
    isd = (1 / np.sqrt(xs.var(0) + 1e-4)).astype(np.float32)
    bn.mean, bn.inv_std, bn.scale, bn.beta_buf = rt.upload(mu), rt.upload(isd), rt.upload(sg * isd), rt.upload(be)
    Wd = rt.alloc(Cc * 9 * Cc, zero=False)
    G = rt.alloc((M, Cc), zero=False)
    part = rt.alloc((nblk, 2, Cc), zero=False)
    ops.conv3x3_wtrans(rt, Wk, Cc, Cc, Wd)(rt.stream)
    ops.conv3x3_stream(rt, rt.upload(dy), N, H, W, Cc, Wd, G, epi=ops.epilogue(bn=bn, bn_x=rt.upload(xs), bn_relu=True, bn_partial=part))(rt.stream)
    dbeta, dgamma, c1, c2 = (rt.alloc(Cc, zero=False) for _ in range(4))
    ops.bn_bwd_finalize(rt, part, nblk, M, Cc, dbeta, dgamma, c1, c2)(rt.stream)
    rt.synchronize()
    da_ref = layout.nchw_to_nhwc(L.conv2d_bwd(layout.nhwc_to_nchw(x), Wr.astype('f8'), layout.nhwc_to_nchw(dy.astype('f8')), (1, 1), 'half')[0]).reshape(M, Cc)
    v = (xs.astype('f8') - mu) * (sg * isd) + be
    safe = np.abs(v) > 1e-4
    got = G.get()
    np.testing.assert_allclose(got[safe], (da_ref * (v >= 0))[safe], rtol=0, atol=3e-6 * np.sqrt(9 * Cc) * np.abs(da_ref).max())
    xhat = (xs.astype('f8') - mu) * isd
    np.testing.assert_allclose(dbeta.get(), got.astype('f8').sum(0), rtol=0, atol=3e-6 * np.sqrt(M) * np.abs(got).max())
    np.testing.assert_allclose(dgamma.get(), (got.astype('f8') * xhat).sum(0), rtol=0, atol=3e-5 * np.sqrt(M) * np.abs(got).max())


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(512, 64, 8, 32, True), (300, 32, 5, 64, False), (2048, 128, 32, 96, True), (96, 256, 3, 32, True), (64, 48, 2, 32, False)])
def test_bn_bwd_finalize_apply_equals_the_two_launches(backend, cfg):
    """dpp_bn_bwd_finalize_apply (finalize of the per-block sums + apply pass in one launch, small maps) against dpp_bn_bwd_finalize +
    dpp_bn_bwd_apply on the same partial sums: dX, dbeta, dgamma and the column sums of dX."""
    rt = get_runtime(backend)
    M, Cc, nb, rpb, with_add = cfg
    rng = np.random.RandomState(83)
    assert bool(rt.lib.dpp_bn_bwd_finalize_apply_ok(M, Cc, nb)) == (Cc % 32 == 0)
    G, X = (rt.upload(rng.normal(size=(M, Cc)).astype(np.float32)) for _ in range(2))
    add = rt.upload(rng.normal(size=(M, Cc)).astype(np.float32)) if with_add else None
    mean, istd, scale = (rt.upload(rng.uniform(0.5, 1.5, Cc).astype(np.float32)) for _ in range(3))
    part = rt.upload(rng.normal(size=(2, Cc, nb)).astype(np.float32) * 10)
    dX1, dX2 = rt.alloc((M, Cc), zero=False), rt.alloc((M, Cc), zero=False)
    db1, dg1, db2, dg2, c1, c2 = (rt.alloc(Cc, zero=False) for _ in range(6))
    nb1, nb2 = -(-M // 32), -(-M // rpb)
    cs1, cs2 = rt.alloc((nb1, Cc), zero=False), rt.alloc((nb2, Cc), zero=False)
    if Cc % 32:
        assert rt.lib.dpp_bn_bwd_finalize_apply(G.ptr, X.ptr, M, Cc, mean.ptr, istd.ptr, scale.ptr, part.ptr, nb, None, dX2.ptr, rpb, cs2.ptr,
                                                db2.ptr, dg2.ptr, 0, rt.stream) != 0
        return
    ops.bn_bwd_finalize(rt, part, nb, M, Cc, db1, dg1, c1, c2)(rt.stream)
    ops.bn_bwd_apply(rt, G, X, M, Cc, mean, istd, scale, c1, c2, dX1, add=add, rpb=32, colsum=cs1)(rt.stream)
    ops.bn_bwd_finalize_apply(rt, G, X, M, Cc, mean, istd, scale, part, nb, dX2, db2, dg2, add=add, rpb=rpb, colsum=cs2)(rt.stream)
    rt.synchronize()
    np.testing.assert_array_equal(db1.get(), db2.get())              # f64 sums of <= 256 floats, one rounding: order-independent in practice
    np.testing.assert_array_equal(dg1.get(), dg2.get())
    np.testing.assert_array_equal(dX1.get(), dX2.get())
    s1, s2 = cs1.get().astype('f8').sum(0), cs2.get().astype('f8').sum(0)
    np.testing.assert_allclose(s2, s1, rtol=0, atol=1e-5 * np.abs(dX1.get()).sum(0).max())
    np.testing.assert_allclose(s2, dX2.get().astype('f8').sum(0), rtol=0, atol=1e-5 * np.abs(dX1.get()).sum(0).max())
