"""dpp_gemm against NumPy float64 on the same seeded inputs (tolerance: f32 round-off of a K-term dot
product, 1e-5 relative to sum|a||b|), every operand layout / row map / prologue / epilogue variant."""
import numpy as np
import pytest

from hipdp import ops
from hipdp.lib import Act, RowMap
from tests.backends import BACKENDS, get_runtime


def _tol(K):
    return 2e-6 * max(8, K) ** 0.5


def _check(out, ref, K, scale=1.0):
    np.testing.assert_allclose(out, ref, rtol=0, atol=_tol(K) * scale * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('tile', [(0, 0, 0), (128, 64, 4), (64, 16, 4), (128, 32, 4), (16, 64, 1), (32, 64, 1)])
def test_gemm_kc_kc(backend, tile):
    rt = get_runtime(backend)
    rng = np.random.RandomState(1)
    M, N, K = 200, 48, 40
    A = rng.normal(size=(M, K)).astype('float32')
    B = rng.normal(size=(N, K)).astype('float32')
    bias = rng.normal(size=N).astype('float32')
    dA, dB, db, dC = rt.upload(A), rt.upload(B), rt.upload(bias), rt.alloc((M, N), zero=False)
    ops.gemm(rt, dA, dB, dC, M, N, K, 1, 1, K, K, N, bias=db, tile=tile)(rt.stream)
    rt.synchronize()
    _check(dC.get(), A.astype('f8') @ B.astype('f8').T + bias, K, 4)


@pytest.mark.parametrize('backend', BACKENDS)
def test_gemm_unaligned_scalar_paths(backend):
    """FC 1024->30 and 30->42 shapes: N, K not multiples of 4 -> scalar loads, zero fill."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(2)
    for (M, N, K) in ((37, 30, 50), (20, 42, 30), (5, 7, 3)):
        A = rng.normal(size=(M, K)).astype('float32')
        W = rng.normal(size=(K, N)).astype('float32')
        dA, dW, dC = rt.upload(A), rt.upload(W), rt.alloc((M, N), zero=False)
        ops.gemm(rt, dA, dW, dC, M, N, K, 1, 0, K, N, N)(rt.stream)           # x . W   (B is [K][N])
        rt.synchronize()
        _check(dC.get(), A.astype('f8') @ W.astype('f8'), K, 4)
        dY = rng.normal(size=(M, N)).astype('float32')
        dYb, dX = rt.upload(dY), rt.alloc((M, K), zero=False)
        ops.gemm(rt, dYb, dW, dX, M, K, N, 1, 1, N, N, K)(rt.stream)          # dy . W^T (B is [n=K][k=N])
        rt.synchronize()
        _check(dX.get(), dY.astype('f8') @ W.astype('f8').T, N, 4)
        dWg = rt.alloc((K, N), zero=False)
        ops.gemm(rt, dA, dYb, dWg, K, N, M, 0, 0, K, N, N)(rt.stream)         # x^T . dy (both [kred][mn])
        rt.synchronize()
        _check(dWg.get(), A.astype('f8').T @ dY.astype('f8'), M, 4)


@pytest.mark.parametrize('backend', BACKENDS)
def test_conv1x1_stride2_fwd_with_bn_relu_bias_residual(backend):
    """1x1/s2 ConvLayer on relu(bn(x)): A rows gathered with the stride-2 row map."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(3)
    Nb, Hi, Wi, Ci, Co, s = 3, 8, 6, 32, 16, 2
    Ho, Wo = Hi // s, Wi // s
    X = rng.normal(size=(Nb, Hi, Wi, Ci)).astype('float32')
    Wk = rng.normal(size=(Co, Ci)).astype('float32')
    mean, scale, beta = (rng.normal(size=Ci).astype('float32') for _ in range(3))
    bias = rng.normal(size=Co).astype('float32')
    res = rng.normal(size=(Nb, Ho, Wo, Co)).astype('float32')
    d = {k: rt.upload(v) for k, v in dict(X=X, Wk=Wk, mean=mean, scale=scale, beta=beta, bias=bias, res=res).items()}
    Y = rt.alloc((Nb, Ho, Wo, Co), zero=False)
    M = Nb * Ho * Wo
    ops.gemm(rt, d['X'], d['Wk'], Y, M, Co, Ci, 1, 1, Ci, Ci, Co, mapA=RowMap.strided(s, Ho, Wo, Hi, Wi),
             actA=ops.act(Act.BN_RELU, d['mean'], d['scale'], d['beta'], Ci), bias=d['bias'], residual=d['res'])(rt.stream)
    rt.synchronize()
    A = np.maximum((X.astype('f8') - mean) * scale + beta, 0)[:, ::s, ::s, :]
    ref = A.reshape(M, Ci) @ Wk.astype('f8').T + bias + res.reshape(M, Co)
    _check(Y.get().reshape(M, Co), ref, Ci, 8)


@pytest.mark.parametrize('backend', BACKENDS)
def test_conv1x1_stride2_dgrad_scatter_accumulate(backend):
    """Data gradient of two 1x1/s2 convs sharing their input: scatter to even pixels, second call adds."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(4)
    Nb, Hi, Wi, Ci, s = 2, 8, 8, 32, 2
    Ho, Wo = Hi // s, Wi // s
    M = Nb * Ho * Wo
    dY1 = rng.normal(size=(M, 16)).astype('float32')
    dY2 = rng.normal(size=(M, 64)).astype('float32')
    W1 = rng.normal(size=(16, Ci)).astype('float32')
    W2 = rng.normal(size=(64, Ci)).astype('float32')
    dH = rt.alloc((Nb, Hi, Wi, Ci), zero=True)
    mp = RowMap.strided(s, Ho, Wo, Hi, Wi)
    b = {k: rt.upload(v) for k, v in dict(dY1=dY1, dY2=dY2, W1=W1, W2=W2).items()}
    ops.gemm(rt, b['dY2'], b['W2'], dH, M, Ci, 64, 1, 0, 64, Ci, Ci, mapC=mp)(rt.stream)
    ops.gemm(rt, b['dY1'], b['W1'], dH, M, Ci, 16, 1, 0, 16, Ci, Ci, mapC=mp, residual=dH)(rt.stream)
    rt.synchronize()
    ref = np.zeros((Nb, Hi, Wi, Ci))
    ref[:, ::s, ::s, :] = (dY1.astype('f8') @ W1 + dY2.astype('f8') @ W2).reshape(Nb, Ho, Wo, Ci)
    _check(dH.get(), ref, 80, 8)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('tile', [(0, 0, 0), (16, 64, 1), (64, 64, 4)])
def test_conv1x1_wgrad_splitk_with_prologue(backend, tile):
    """Filter gradient dW[o][c] = sum_m dY[m][o] * relu(bn(X))[map(m)][c]: both operands [kred][mn],
    prologue on B, stride-2 row map on the reduction index, split-K + deterministic reduce."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(5)
    Nb, Hi, Wi, Ci, Co, s = 4, 8, 8, 64, 16, 2
    Ho, Wo = Hi // s, Wi // s
    M = Nb * Ho * Wo
    X = rng.normal(size=(Nb, Hi, Wi, Ci)).astype('float32')
    dY = rng.normal(size=(M, Co)).astype('float32')
    mean, scale, beta = (rng.normal(size=Ci).astype('float32') for _ in range(3))
    b = {k: rt.upload(v) for k, v in dict(X=X, dY=dY, mean=mean, scale=scale, beta=beta).items()}
    splitk = 3
    part = rt.alloc((splitk, Co, Ci), zero=False)
    dW = rt.alloc((Co, Ci), zero=False)
    ops.gemm(rt, b['dY'], b['X'], None, Co, Ci, M, 0, 0, Co, Ci, mapB=RowMap.strided(s, Ho, Wo, Hi, Wi),
             actB=ops.act(Act.BN_RELU, b['mean'], b['scale'], b['beta'], Ci), splitk=splitk, partial=part, tile=tile)(rt.stream)
    ops.reduce_partials(rt, part, splitk, Co * Ci, dW)(rt.stream)
    rt.synchronize()
    A = np.maximum((X.astype('f8') - mean) * scale + beta, 0)[:, ::s, ::s, :].reshape(M, Ci)
    _check(dW.get(), dY.astype('f8').T @ A, M, 8)


@pytest.mark.parametrize('backend', BACKENDS)
def test_fc_prologue_channel_modulo(backend):
    """FC1 on the flattened NHWC map: BN+ReLU prologue with channel = k % C."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(6)
    M, Cc, HW, N = 6, 16, 4, 24
    K = Cc * HW
    X = rng.normal(size=(M, K)).astype('float32')
    W = rng.normal(size=(K, N)).astype('float32')
    mean, scale, beta = (rng.normal(size=Cc).astype('float32') for _ in range(3))
    b = {k: rt.upload(v) for k, v in dict(X=X, W=W, mean=mean, scale=scale, beta=beta).items()}
    Y = rt.alloc((M, N), zero=False)
    a = ops.act(Act.BN_RELU, b['mean'], b['scale'], b['beta'], Cc)
    ops.gemm(rt, b['X'], b['W'], Y, M, N, K, 1, 0, K, N, N, actA=a)(rt.stream)
    rt.synchronize()
    A = np.maximum((X.astype('f8').reshape(M, HW, Cc) - mean) * scale + beta, 0).reshape(M, K)
    _check(Y.get(), A @ W.astype('f8'), K, 8)
    # weight gradient: A operand is [kred = sample][i = k], prologue channel = i % C
    dY = rng.normal(size=(M, N)).astype('float32')
    dYb, dW = rt.upload(dY), rt.alloc((K, N), zero=False)
    ops.gemm(rt, b['X'], dYb, dW, K, N, M, 0, 0, K, N, N, actA=a)(rt.stream)
    rt.synchronize()
    _check(dW.get(), A.T @ dY.astype('f8'), M, 8)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(64, 16, 32, 16), (64, 32, 256, 64), (128, 32, 64, 128), (128, 64, 16, 64), (64, 16, 40, 24)])
def test_rowstream_variant_fwd_and_dgrad(backend, cfg):
    """The barrier-free row-streaming kernel (variant 1): 1x1 conv forward with BN+ReLU prologue, bias, residual and fused
    statistics, and the data gradient (B in [k][n] layout) with stride-2 scatter."""
    rt = get_runtime(backend)
    bm, bn, K, N = cfg
    rng = np.random.RandomState(8)
    Nb, Hi, Wi, s = 3, 10, 6, 2
    M = Nb * Hi * Wi
    X = rng.normal(size=(M, K)).astype('float32')
    Wk = (rng.normal(size=(N, K)) * 0.3).astype('float32')
    cm = K if K % 4 == 0 else 1
    mean, scale, beta = (rng.normal(size=K).astype('float32') for _ in range(3))
    bias = rng.normal(size=N).astype('float32')
    res = rng.normal(size=(M, N)).astype('float32')
    b = {k: rt.upload(v) for k, v in dict(X=X, Wk=Wk, mean=mean, scale=scale, beta=beta, bias=bias, res=res).items()}
    Y = rt.alloc((M, N), zero=False)
    nblk = -(-M // bm)
    stats = rt.alloc((nblk, 2, N), zero=False)
    ops.gemm(rt, b['X'], b['Wk'], Y, M, N, K, 1, 1, K, K, N, actA=ops.act(Act.BN_RELU, b['mean'], b['scale'], b['beta'], K),
             bias=b['bias'], residual=b['res'], tile=(bm, bn, 4), variant=1, epi=ops.epilogue(stats=stats))(rt.stream)
    gamma, mo, io, so = rt.upload(np.ones(N, 'float32')), rt.alloc(N), rt.alloc(N), rt.alloc(N)
    ops.bn_finalize(rt, stats, nblk, M, bm, N, gamma, 1e-4, mo, io, so)(rt.stream)
    rt.synchronize()
    A = np.maximum((X.astype('f8') - mean) * scale + beta, 0)
    ref = A @ Wk.astype('f8').T + bias + res
    _check(Y.get(), ref, K, 8)
    np.testing.assert_allclose(mo.get(), ref.mean(0), rtol=0, atol=3e-6 * np.abs(ref).max())
    np.testing.assert_allclose(io.get(), 1 / np.sqrt(ref.var(0) + np.float32(1e-4)), rtol=3e-5)
    # data gradient: dA[m][c] = sum_o dY[m][o] Wk[o][c], scattered to the stride-2 positions of a zeroed map
    Ho, Wo = Hi // s, Wi // s
    Mo = Nb * Ho * Wo
    dY = rng.normal(size=(Mo, N)).astype('float32')
    dH = rt.alloc((M, K), zero=True)
    mp = RowMap.strided(s, Ho, Wo, Hi, Wi)
    ops.gemm(rt, rt.upload(dY), b['Wk'], dH, Mo, K, N, 1, 0, N, K, K, mapC=mp, tile=(bm, 16 if K < 32 else 32, 4), variant=1)(rt.stream)
    rt.synchronize()
    refd = np.zeros((Nb, Hi, Wi, K))
    refd[:, ::s, ::s, :] = (dY.astype('f8') @ Wk.astype('f8')).reshape(Nb, Ho, Wo, K)
    _check(dH.get().reshape(Nb, Hi, Wi, K), refd, N, 8)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(256, 64, 96), (256, 128, 96), (128, 32, 96), (128, 96, 96), (256, 64, 512), (128, 32, 1024),
                                 (256, 128, 4160), (128, 32, 32960)])       # (the last two: more row tiles than workgroups -- 130 on 128, 1 030 on 1 024: the walk)
def test_ksplit_variant_fwd_and_dgrad(backend, cfg):
    """The K-split kernel (variant 2: 32 rows x all columns x the whole K per workgroup, the waves split K, partial tiles summed in
    the epilogue's LDS images): 1x1 conv forward (filters K-contiguous) with BN+ReLU prologue, bias, residual and fused statistics,
    and the data gradient of a channel-expanding conv (B in [k][n] layout) accumulating onto an earlier share."""
    rt = get_runtime(backend)
    K, N, M = cfg
    bn = 64 if K == 256 else 32
    rng = np.random.RandomState(18)
    X = rng.normal(size=(M, K)).astype('float32')
    Wk = (rng.normal(size=(N, K)) * 0.3).astype('float32')
    mean, scale, beta = (rng.normal(size=K).astype('float32') for _ in range(3))
    bias = rng.normal(size=N).astype('float32')
    res = rng.normal(size=(M, N)).astype('float32')
    b = {k: rt.upload(v) for k, v in dict(X=X, Wk=Wk, mean=mean, scale=scale, beta=beta, bias=bias, res=res).items()}
    Y = rt.alloc((M, N), zero=False)
    nblk = M // 32
    stats = rt.alloc((nblk, 2, N), zero=False)
    ops.gemm(rt, b['X'], b['Wk'], Y, M, N, K, 1, 1, K, K, N, actA=ops.act(Act.BN_RELU, b['mean'], b['scale'], b['beta'], K),
             bias=b['bias'], residual=b['res'], tile=(32, bn, 4), variant=2, epi=ops.epilogue(stats=stats))(rt.stream)
    gamma, mo, io, so = rt.upload(np.ones(N, 'float32')), rt.alloc(N), rt.alloc(N), rt.alloc(N)
    ops.bn_finalize(rt, stats, nblk, M, 32, N, gamma, 1e-4, mo, io, so)(rt.stream)
    rt.synchronize()
    A = np.maximum((X.astype('f8') - mean) * scale + beta, 0)
    ref = A @ Wk.astype('f8').T + bias + res
    _check(Y.get(), ref, K, 8)
    np.testing.assert_allclose(mo.get(), ref.mean(0), rtol=0, atol=3e-6 * np.abs(ref).max())
    np.testing.assert_allclose(io.get(), 1 / np.sqrt(ref.var(0) + np.float32(1e-4)), rtol=3e-5)
    # data gradient of a conv with K output channels and N input channels: dA[m][c] = share[m][c] + sum_o dY[m][o] W2[o][c]
    W2 = (rng.normal(size=(K, N)) * 0.3).astype('float32')
    dY = rng.normal(size=(M, K)).astype('float32')
    share = rng.normal(size=(M, N)).astype('float32')
    dH = rt.upload(share)
    ops.gemm(rt, rt.upload(dY), rt.upload(W2), dH, M, N, K, 1, 0, K, N, N, residual=dH, tile=(32, bn, 4), variant=2)(rt.stream)
    rt.synchronize()
    _check(dH.get(), share + dY.astype('f8') @ W2.astype('f8'), K, 8)
    # what the kernel does not take is refused, not mis-computed
    with pytest.raises(Exception):
        ops.gemm(rt, b['X'], b['Wk'], Y, M - 8, N, K, 1, 1, K, K, N, tile=(32, bn, 4), variant=2)(rt.stream)
    # ... and the host can ask first (dpp_gemm_variant_rows): rows per workgroup when the kernel takes it, 0 for ragged rows or an
    # operand that is not 16-byte aligned, so that the engine describes those with the generic tile
    assert ops.gemm_variant_rows(rt, ops.gemm(rt, b['X'], b['Wk'], Y, M, N, K, 1, 1, K, K, N, tile=(32, bn, 4), variant=2)) == 32
    assert ops.gemm_variant_rows(rt, ops.gemm(rt, b['X'], b['Wk'], Y, M - 8, N, K, 1, 1, K, K, N, tile=(32, bn, 4), variant=2)) == 0
    Xo = rt.alloc(M * K + 4, zero=False).view(1, (M, K))                                     # 4 bytes off a 16-byte boundary
    assert ops.gemm_variant_rows(rt, ops.gemm(rt, Xo, b['Wk'], Y, M, N, K, 1, 1, K, K, N, tile=(32, bn, 4), variant=2)) == 0
    assert ops.gemm_variant_rows(rt, ops.gemm(rt, b['X'], b['Wk'], Y, M, N, K, 1, 1, K, K, N, tile=(64, bn, 4), variant=0)) == 0


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('shape', [(64, 16, 128), (16, 64, 64)])
def test_stream16_variant_against_the_tiled_kernel(backend, shape):
    """The barrier-free row-streaming kernel for K = 64 -> 16 columns (variant 3) against float64 and against the LDS-tiled kernel on
    the same inputs: forward with BN+ReLU prologue, bias, residual and fused statistics; data gradient (B in [k][n] layout) with the fused
    BatchNorm-backward epilogue (ReLU mask, per-block (sum G, sum G*xhat)) accumulating onto an earlier share; the same for the
    K = 16 -> 64 columns shape (four column tiles per wave, one 16-row tile)."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(28)
    K, N, rows = shape                     # rows per workgroup = rows per BatchNorm partial block
    M = 384
    X = rng.normal(size=(M, K)).astype('float32')
    Wk = (rng.normal(size=(N, K)) * 0.3).astype('float32')
    mean, scale, beta = (rng.normal(size=K).astype('float32') for _ in range(3))
    bias = rng.normal(size=N).astype('float32')
    b = {k: rt.upload(v) for k, v in dict(X=X, Wk=Wk, mean=mean, scale=scale, beta=beta, bias=bias).items()}
    Y = rt.alloc((M, N), zero=False)
    res = rng.normal(size=(M, N)).astype('float32')
    Y.set(res)
    stats = rt.alloc((M // rows, 2, N), zero=False)
    ops.gemm(rt, b['X'], b['Wk'], Y, M, N, K, 1, 1, K, K, N, actA=ops.act(Act.BN_RELU, b['mean'], b['scale'], b['beta'], K),
             bias=b['bias'], residual=Y, tile=(rows, N, 4), variant=3, epi=ops.epilogue(stats=stats))(rt.stream)
    gamma, mo, io, so = rt.upload(np.ones(N, 'float32')), rt.alloc(N), rt.alloc(N), rt.alloc(N)
    ops.bn_finalize(rt, stats, M // rows, M, rows, N, gamma, 1e-4, mo, io, so)(rt.stream)
    rt.synchronize()
    ref = np.maximum((X.astype('f8') - mean) * scale + beta, 0) @ Wk.astype('f8').T + bias + res
    _check(Y.get(), ref, K, 8)
    np.testing.assert_allclose(mo.get(), ref.mean(0), rtol=0, atol=3e-6 * np.abs(ref).max())
    np.testing.assert_allclose(io.get(), 1 / np.sqrt(ref.var(0) + np.float32(1e-4)), rtol=3e-5)
    # data gradient with the fused BatchNorm-backward epilogue, on both kernels
    W2 = (rng.normal(size=(K, N)) * 0.3).astype('float32')
    dY = rng.normal(size=(M, K)).astype('float32')
    bnx = rng.normal(size=(M, N)).astype('float32')
    share = rng.normal(size=(M, N)).astype('float32')

    class BN(object):
        pass
    bnl = BN()
    bm, bs, bb, bi = (rng.normal(0, 0.3, N).astype('float32'), rng.uniform(0.5, 1.5, N).astype('float32'), rng.normal(0, 0.3, N).astype('float32'),
                      rng.uniform(0.5, 1.5, N).astype('float32'))
    bnl.mean, bnl.scale, bnl.beta_buf, bnl.inv_std = rt.upload(bm), rt.upload(bs), rt.upload(bb), rt.upload(bi)
    out = {}
    for variant, tile in ((3, (rows, N, 4)), (0, (64, 16, 4))):
        dH = rt.upload(share)
        nb = M // tile[0]
        part = rt.alloc((nb, 2, N), zero=False)
        ops.gemm(rt, rt.upload(dY), rt.upload(W2), dH, M, N, K, 1, 0, K, N, N, residual=dH, tile=tile, variant=variant,
                 epi=ops.epilogue(bn=bnl, bn_x=rt.upload(bnx), bn_relu=True, bn_partial=part))(rt.stream)
        rt.synchronize()
        p = part.get().reshape(2, N, nb)                      # [s][c][b], block index fastest
        out[variant] = (dH.get(), p.sum(axis=2))
    g = share.astype('f8') + dY.astype('f8') @ W2.astype('f8')
    keep = ((bnx.astype('f8') - bm) * bs + bb) >= 0
    g = np.where(keep, g, 0.0)
    _check(out[3][0], g, K, 8)
    xhat = (bnx.astype('f8') - bm) * bi
    np.testing.assert_allclose(out[3][1][0], g.sum(0), rtol=0, atol=2e-5 * np.abs(g).sum(0).max())
    np.testing.assert_allclose(out[3][1][1], (g * xhat).sum(0), rtol=0, atol=2e-5 * np.abs(g * xhat).sum(0).max())
    np.testing.assert_allclose(out[3][0], out[0][0], rtol=0, atol=1e-5 * np.abs(g).max())
    np.testing.assert_allclose(out[3][1], out[0][1], rtol=0, atol=2e-5 * np.abs(g).sum(0).max())


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(64, 256, 96, 32), (64, 256, 192, 96), (32, 128, 128, 64), (16, 64, 256, 128), (16, 64, 96, 32), (64, 64, 64, 32),
                                 (32, 192, 64, 64)])
def test_expand_variant_fwd_and_dgrad(backend, cfg):
    """The wave-autonomous kernel for the channel-expanding 1x1 convolutions (variant 4: a wave owns rpw rows x 64 columns x the whole
    K, columns dealt interleaved to the accumulator tiles so that the MFMA D layout yields 16-byte accesses; no LDS, no barrier):
    forward with BN+ReLU prologue, bias, residual (aliasing the output) and fused statistics -- merged over the wave's 32-row iterations
    by Chan's update -- and the data gradient (B in [k][n] layout) with the fused BatchNorm-backward epilogue accumulating onto an
    earlier share, both against float64 and against the LDS-tiled kernel."""
    rt = get_runtime(backend)
    K, N, M, rpw = cfg
    rng = np.random.RandomState(44)
    X = rng.normal(size=(M, K)).astype('float32')
    Wk = (rng.normal(size=(N, K)) * 0.3).astype('float32')
    mean, scale, beta = (rng.normal(size=K).astype('float32') for _ in range(3))
    bias = rng.normal(size=N).astype('float32')
    res = (rng.normal(size=(M, N)) + np.linspace(-3, 3, N)).astype('float32')          # column means well away from 0: the M2 merge matters
    b = {k: rt.upload(v) for k, v in dict(X=X, Wk=Wk, mean=mean, scale=scale, beta=beta, bias=bias).items()}
    ref = np.maximum((X.astype('f8') - mean) * scale + beta, 0) @ Wk.astype('f8').T + bias + res
    nblk = M // rpw
    for act_mode in (Act.BN_RELU, None):
        Y = rt.upload(res)
        stats = rt.alloc((nblk, 2, N), zero=False)
        a = ops.act(act_mode, b['mean'], b['scale'], b['beta'], K) if act_mode is not None else None
        L = ops.gemm(rt, b['X'], b['Wk'], Y, M, N, K, 1, 1, K, K, N, actA=a, bias=b['bias'], residual=Y, tile=(rpw, 64, 4), variant=4,
                     epi=ops.epilogue(stats=stats))
        assert ops.gemm_variant_rows(rt, L) == rpw
        L(rt.stream)
        gamma, mo, io, so = rt.upload(np.ones(N, 'float32')), rt.alloc(N), rt.alloc(N), rt.alloc(N)
        ops.bn_finalize(rt, stats, nblk, M, rpw, N, gamma, 1e-4, mo, io, so)(rt.stream)
        rt.synchronize()
        want = ref if act_mode is not None else X.astype('f8') @ Wk.astype('f8').T + bias + res
        _check(Y.get(), want, K, 8)
        np.testing.assert_allclose(mo.get(), want.mean(0), rtol=0, atol=3e-6 * np.abs(want).max())
        np.testing.assert_allclose(io.get(), 1 / np.sqrt(want.var(0) + np.float32(1e-4)), rtol=3e-5)
    # plain (no bias / residual / statistics), rows chosen by the library
    Y = rt.alloc((M, N), zero=False)
    L = ops.gemm(rt, b['X'], b['Wk'], Y, M, N, K, 1, 1, K, K, N, tile=(0, 64, 4), variant=4)
    assert ops.gemm_variant_rows(rt, L) == 32
    L(rt.stream)
    rt.synchronize()
    _check(Y.get(), X.astype('f8') @ Wk.astype('f8').T, K, 8)
    # data gradient with the fused BatchNorm-backward epilogue, on this kernel and on the LDS-tiled one
    W2 = (rng.normal(size=(K, N)) * 0.3).astype('float32')
    dY = rng.normal(size=(M, K)).astype('float32')
    bnx = rng.normal(size=(M, N)).astype('float32')
    share = rng.normal(size=(M, N)).astype('float32')

    class BN(object):
        pass
    bnl = BN()
    bm, bs, bb, bi = (rng.normal(0, 0.3, N).astype('float32'), rng.uniform(0.5, 1.5, N).astype('float32'), rng.normal(0, 0.3, N).astype('float32'),
                      rng.uniform(0.5, 1.5, N).astype('float32'))
    bnl.mean, bnl.scale, bnl.beta_buf, bnl.inv_std = rt.upload(bm), rt.upload(bs), rt.upload(bb), rt.upload(bi)
    out = {}
    for variant, tile, acc in ((4, (rpw, 64, 4), True), (0, (32, 64, 1), True), (4, (rpw, 64, 4), False)):
        dH = rt.upload(share)
        nb = M // tile[0]
        part = rt.alloc((nb, 2, N), zero=False)
        ops.gemm(rt, rt.upload(dY), rt.upload(W2), dH, M, N, K, 1, 0, K, N, N, residual=dH if acc else None, tile=tile, variant=variant,
                 epi=ops.epilogue(bn=bnl, bn_x=rt.upload(bnx), bn_relu=True, bn_partial=part))(rt.stream)
        rt.synchronize()
        out[(variant, acc)] = (dH.get(), part.get().reshape(2, N, nb).sum(axis=2))           # [s][c][b], block index fastest
    keep = ((bnx.astype('f8') - bm) * bs + bb) >= 0
    xhat = (bnx.astype('f8') - bm) * bi
    for acc in (True, False):
        g = np.where(keep, (share.astype('f8') if acc else 0.0) + dY.astype('f8') @ W2.astype('f8'), 0.0)
        _check(out[(4, acc)][0], g, K, 8)
        np.testing.assert_allclose(out[(4, acc)][1][0], g.sum(0), rtol=0, atol=2e-5 * np.abs(g).sum(0).max())
        np.testing.assert_allclose(out[(4, acc)][1][1], (g * xhat).sum(0), rtol=0, atol=2e-5 * np.abs(g * xhat).sum(0).max())
    np.testing.assert_allclose(out[(4, True)][0], out[(0, True)][0], rtol=0, atol=1e-5 * np.abs(out[(0, True)][0]).max())
    # what the kernel does not take is refused, and the host can ask first
    with pytest.raises(Exception):
        ops.gemm(rt, b['X'], b['Wk'], Y, M - 8, N, K, 1, 1, K, K, N, tile=(32, 64, 4), variant=4)(rt.stream)
    assert ops.gemm_variant_rows(rt, ops.gemm(rt, b['X'], b['Wk'], Y, M - 8, N, K, 1, 1, K, K, N, tile=(32, 64, 4), variant=4)) == 0
    assert ops.gemm_variant_rows(rt, ops.gemm(rt, b['X'], b['Wk'], Y, M, N, K, 1, 1, K, K, N, tile=(48, 64, 4), variant=4)) == 0
    Xo = rt.alloc(M * K + 4, zero=False).view(1, (M, K))                                     # 4 bytes off a 16-byte boundary
    assert ops.gemm_variant_rows(rt, ops.gemm(rt, Xo, b['Wk'], Y, M, N, K, 1, 1, K, K, N, tile=(32, 64, 4), variant=4)) == 0


@pytest.mark.parametrize('backend', BACKENDS)
def test_every_gemm_instantiation_of_a_small_resnet(backend):
    """Every distinct dpp_gemm problem the train plans of a small ResNet launch (tile / split-K heuristics, strided row maps,
    BN+ReLU prologues, bias / residual / statistics epilogues), stand-alone against float64 (tests/gemm_cases.py); the
    benchmarked batch of 128 runs the same check on the GPU (tests/test_full_size.py)."""
    from hipdp import engine
    from net.resnet import ResNet, ResNetParams
    from tests import gemm_cases
    rt = get_runtime(backend)
    net = ResNet(np.random.RandomState(23455), cfgParams=ResNetParams(type=0, nChan=1, wIn=32, hIn=32, batchSize=4, numJoints=1, nDims=30))
    eng = engine.CompiledNet(net, train=True, runtime=rt, loss=dict(kind='embedding'))
    checked, skipped = gemm_cases.check_all(rt, eng)
    assert len(checked) >= 15, (len(checked), len(skipped))


def _bf16_round(a):
    """Round-to-nearest-even to bfloat16, returned as float32 (what the kernel does when it writes an operand to LDS)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('precision', [0, 1])
@pytest.mark.parametrize('layout', ['fwd', 'dgrad', 'wgrad', 'tn'])
def test_fc_gemm_layouts_and_precisions(backend, precision, layout):
    """dpp_fc_gemm (the FC1 weight-streaming kernel): the three layouts FC1 runs in (forward: A K-contiguous, W [K][N];
    data gradient: both K-contiguous; weight gradient: both MN-contiguous, transposed on the way into LDS) plus the fourth
    combination, ragged M / N / K, split-K partials, BN+ReLU prologue on A, bias and residual -- f32 against float64, bf16
    against float64 on bf16-rounded operands (f32 accumulation: twice the tolerance)."""
    rt = get_runtime(backend)
    rng = np.random.RandomState(17 + precision)
    a_kc, b_kc = {'fwd': (1, 0), 'dgrad': (1, 1), 'wgrad': (0, 0), 'tn': (0, 1)}[layout]
    shapes = [(40, 72, 200, 1, 0, 'act+bias+res'), (128, 64, 192, 3, 32, ''), (200, 132, 96, 1, 32, 'relu'), (16, 8, 64, 2, 0, 'act')]
    if True:
        # whole 128-row tiles, K slices of whole 32-deep chunks: the three-stage kernel (fc_stream_kernel, f32 and bf16), 64- and 128-wide
        # column tiles (the latter needs >= 256 workgroups), one and several chunks per slice, every epilogue / prologue feature
        shapes += [(128, 192, 96, 1, 0, 'act+bias+res'), (256, 64, 256, 2, 0, 'relu'), (256, 512, 1024, 32, 0, 'act'), (128, 128, 64, 1, 0, 'bias')]
    for (M, N, K, splitk, kchunk, feats) in shapes:
        A = rng.normal(0, 1, (M, K) if a_kc else (K, M)).astype(np.float32)
        Bm = rng.normal(0, 1, (N, K) if b_kc else (K, N)).astype(np.float32)
        cmod = (K if a_kc else M)
        actA, Aact = None, A.astype(np.float64)
        if 'act' in feats and cmod % 4 == 0:
            mean, scale, beta = (rng.normal(0, 0.3, cmod).astype(np.float32), rng.uniform(0.5, 1.5, cmod).astype(np.float32),
                                 rng.normal(0, 0.3, cmod).astype(np.float32))
            actA = ops.act(Act.BN_RELU, rt.upload(mean), rt.upload(scale), rt.upload(beta), cmod)
            Aact = np.maximum(((A - mean) * scale + beta).astype(np.float32), 0).astype(np.float64)
        elif 'relu' in feats:
            actA = ops.act(Act.RELU, None, None, None, cmod)
            Aact = np.maximum(A, 0).astype(np.float64)
        Bq = Bm.astype(np.float64)
        if precision == 1:
            Aact, Bq = _bf16_round(Aact.astype(np.float32)).astype(np.float64), _bf16_round(Bm).astype(np.float64)
        ref = (Aact if a_kc else Aact.T) @ (Bq.T if b_kc else Bq)
        bias = rng.normal(0, 1, N).astype(np.float32) if 'bias' in feats and splitk == 1 else None
        res = rng.normal(0, 1, (M, N)).astype(np.float32) if 'res' in feats and splitk == 1 else None
        Cb = rt.alloc((M, N), zero=False)
        if res is not None:
            Cb.set(res)
        part = rt.alloc(splitk * M * N, zero=False) if splitk > 1 else None
        ops.fc_gemm(rt, rt.upload(A), rt.upload(Bm), Cb if splitk == 1 else None, M, N, K, a_kc, b_kc, K if a_kc else M, K if b_kc else N, N,
                    actA=actA, bias=rt.upload(bias) if bias is not None else None, residual=Cb if res is not None else None, splitk=splitk,
                    partial=part, precision=precision, kchunk=kchunk)(rt.stream)
        rt.synchronize()
        got = part.get().reshape(splitk, M, N).astype(np.float64).sum(axis=0) if splitk > 1 else Cb.get().astype(np.float64)
        want = ref + (bias if bias is not None else 0) + (res if res is not None else 0)
        tol = 6e-7 * (np.sqrt(K) + 4) * max(1.0, float(np.abs(Aact).max()) * float(np.abs(Bq).max()))
        if precision == 1:
            tol *= 2        # v_mfma_f32_16x16x32_bf16 adds the 32 products of a k-step in its own order before the f32 accumulate
        assert np.abs(got - want).max() < tol, (layout, precision, (M, N, K, splitk), np.abs(got - want).max(), tol)



@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('shape', [(16, 64), (64, 16), (16, 32), (64, 32), (32, 64), (32, 128), (128, 32), (128, 64), (64, 128), (64, 256),
                                   (256, 64), (256, 128)])
def test_wgrad_stream_matches_float64(backend, shape):
    """dpp_wgrad_stream (csrc/wgrad.hip): the filter gradient of every 1x1 layer shape of the ResNet as per-slice partials -- with the
    BatchNorm + ReLU prologue on X, a ragged last slice, and the stride-2 row map of the projection blocks -- against float64."""
    rt = get_runtime(backend)
    Co, Ci = shape
    rng = np.random.RandomState(Co * 1000 + Ci)
    for strided in (False, True):
        if strided:
            N, Ho, Wo, s = 3, 5, 7, 2
            Hi, Wi = Ho * s, Wo * s
            M, rows_x = N * Ho * Wo, N * Hi * Wi
            mp = RowMap.strided(s, Ho, Wo, Hi, Wi)
            n, q = np.divmod(np.arange(M), Ho * Wo)
            y, x = np.divmod(q, Wo)
            xrow = n * Hi * Wi + (y * s) * Wi + x * s
        else:
            M = rows_x = 150                               # not a multiple of the slice: the last one is ragged
            mp, xrow = None, np.arange(M)
        dY = rng.normal(size=(M, Co)).astype(np.float32)
        X = rng.normal(size=(rows_x, Ci)).astype(np.float32)
        mean, scale, beta = (rng.normal(size=Ci).astype(np.float32), rng.uniform(0.5, 1.5, Ci).astype(np.float32),
                             rng.normal(0, 0.3, Ci).astype(np.float32))
        for mode in (0, 3):
            rpw = 32
            nsl = rt.lib.dpp_wgrad_stream_slices(Co, Ci, M, rpw)
            assert nsl >= 1
            part = rt.upload(np.full((nsl, Co, Ci), np.nan, np.float32))
            bufs = [rt.upload(a) for a in (mean, scale, beta)]
            act = ops.act(mode, *bufs, cmod=Ci) if mode else None
            ops.wgrad_stream(rt, rt.upload(dY), Co, rt.upload(X), Ci, M, rpw, part, mapX=mp, actX=act)(rt.stream)
            rt.synchronize()
            got = part.get().astype(np.float64).sum(axis=0)
            Xa = X[xrow].astype(np.float64)
            if mode:
                Xa = np.maximum((Xa - mean) * scale + beta, 0.0)
            ref = dY.astype(np.float64).T @ Xa
            _check(got, ref, M, 4)
    assert rt.lib.dpp_wgrad_stream_slices(48, 64, 100, 32) == 0          # other shapes stay on dpp_gemm


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('cfg', [(128, 256, 128, True), (6, 128, 256, True), (9, 384, 128, False), (1, 128, 128, True), (37, 128, 128, True)])
def test_fc_wgrad_stream(backend, cfg):
    """dpp_fc_wgrad_stream: dW [K][N] = act(X)^T . dY with the BatchNorm + ReLU prologue of the flattened map (channel = column % cmod)
    against float64 -- whole steps of four rows, a ragged last step, and no prologue; shapes it does not take are refused."""
    rt = get_runtime(backend)
    Nb, K, N, with_act = cfg
    rng = np.random.RandomState(77)
    X = rng.normal(size=(Nb, K)).astype('float32')
    dY = rng.normal(size=(Nb, N)).astype('float32')
    cmod = 32
    mean, scale, beta = (rng.normal(size=cmod) * 0.3).astype('float32'), rng.uniform(0.5, 1.5, cmod).astype('float32'), (rng.normal(size=cmod) * 0.3).astype('float32')
    A = X.astype('f8')
    act = None
    keep = None
    if with_act:
        ch = np.arange(K) % cmod
        A = np.maximum((A - mean[ch]) * scale[ch] + beta[ch], 0)
        keep = [rt.upload(v) for v in (mean, scale, beta)]
        act = ops.act(Act.BN_RELU, keep[0], keep[1], keep[2], cmod)
    dW = rt.alloc((K, N), zero=False)
    assert rt.lib.dpp_fc_wgrad_stream_ok(Nb, K, N) == 1
    ops.fc_wgrad_stream(rt, rt.upload(X), rt.upload(dY), dW, Nb, K, N, actX=act)(rt.stream)
    rt.synchronize()
    ref = A.T @ dY.astype('f8')
    np.testing.assert_allclose(dW.get(), ref, rtol=0, atol=3e-6 * np.sqrt(Nb) * np.abs(ref).max())
    assert rt.lib.dpp_fc_wgrad_stream_ok(Nb, K + 64, N) == 0 and rt.lib.dpp_fc_wgrad_stream_ok(Nb, K, 96) == 0
    with pytest.raises(Exception):
        ops.fc_wgrad_stream(rt, rt.upload(X), rt.upload(dY), dW, Nb, K, 96, actX=act)(rt.stream)
