"""
oracle/layers.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

NumPy restatement of the reference's layer arithmetic, forward and backward.
All functions are dtype-generic: feed float64 for the arbiter, float32 to mimic
`floatX=float32`.  Tensors are NCHW like the reference.

Third-party semantics assumed (Theano 0.9, un-vendored, parity unpinned):
  * conv2d is a TRUE convolution (filter_flip=True), border_mode 'half' pads k//2,
    `subsample` keeps every s-th output of the stride-1 result;
  * pool_2d(ignore_border=True, mode='max') uses non-overlapping windows, floor(H/ds);
    its gradient goes to EVERY window element equal to the maximum (Theano's CPU MaxPoolGrad: `if (a == maximum) gx += gz`;
    the reference itself trained through cuDNN 5 pooling, whose tie rule is not Theano-CPU's -- the oracle and the kernels
    follow the Theano-CPU graph, which is the one north_star times and compares against);
  * T.var is the biased variance; T.maximum(x, 0) passes the gradient where x >= 0
    (Theano's Maximum.grad uses eq(out, x)).
"""
import numpy as np


# --------------------------------------------------------------------------- bf16 operands (BASELINE config 5)
def bf16_round(x):
    """Round to bfloat16 (8 bits of significand, round to nearest even) and return the value in x's dtype: what the bf16 kernels
    do to an MFMA operand when they stage it (csrc: dpp_bf16_rne).  float64 input is first rounded to float32, the type the device
    holds the operand in."""
    x = np.asarray(x)
    f = np.ascontiguousarray(x, np.float32)
    u = f.view(np.uint32)
    r = ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) & np.uint32(0xFFFF0000)).view(np.float32)
    r = np.where(np.isfinite(f), r, f)
    return r.astype(x.dtype).reshape(x.shape)


# --------------------------------------------------------------------------- conv
def _out_hw(H, W, kh, kw, stride, border_mode):
    """Shape rule of ConvLayerParams.update, /root/reference/src/net/convlayer.py:131-163."""
    if border_mode in ('half', 'same'):
        oh, ow = H, W
    elif border_mode == 'valid':
        oh, ow = H - kh + 1, W - kw + 1
    else:
        raise ValueError("Unknown border mode")
    return int(np.ceil(oh / float(stride[0]))), int(np.ceil(ow / float(stride[1])))


def _pad_amount(kh, kw, border_mode):
    if border_mode in ('half', 'same'):
        return kh // 2, kw // 2
    return 0, 0


def _im2col(xp, kh, kw, oh, ow, stride):
    """cols[n, i, j, c, a, b] = xp[n, c, i*s0 + a, j*s1 + b] (a strided view, no copy)."""
    N, C, Hp, Wp = xp.shape
    s = xp.strides
    shape = (N, oh, ow, C, kh, kw)
    strides = (s[0], s[2] * stride[0], s[3] * stride[1], s[1], s[2], s[3])
    return np.lib.stride_tricks.as_strided(xp, shape=shape, strides=strides, writeable=False)


def conv2d_fwd(x, W, b=None, stride=(1, 1), border_mode='valid'):
    """
    y = conv2d(x, W, subsample=stride, border_mode) + b[None,:,None,None]
    /root/reference/src/net/convlayer.py:230-240 (true convolution: kernel flipped).
    x: (N,C,H,W)  W: (F,C,kh,kw)  ->  (N,F,oh,ow)
    """
    N, C, H, Wd = x.shape
    F, C2, kh, kw = W.shape
    assert C == C2
    ph, pw = _pad_amount(kh, kw, border_mode)
    oh, ow = _out_hw(H, Wd, kh, kw, stride, border_mode)
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    cols = _im2col(xp, kh, kw, oh, ow, stride)
    Wf = W[:, :, ::-1, ::-1].reshape(F, C * kh * kw)          # flip = true convolution
    y = cols.reshape(N * oh * ow, C * kh * kw) @ Wf.T
    y = y.reshape(N, oh, ow, F).transpose(0, 3, 1, 2)
    if b is not None:
        y = y + b[None, :, None, None]
    return np.ascontiguousarray(y)


def conv2d_bwd(x, W, dy, stride=(1, 1), border_mode='valid', need_dx=True):
    """Exact gradients of conv2d_fwd w.r.t. x, W, b (what T.grad produces,
    /root/reference/src/trainer/poseregnettrainer.py:110-111)."""
    N, C, H, Wd = x.shape
    F, _, kh, kw = W.shape
    ph, pw = _pad_amount(kh, kw, border_mode)
    oh, ow = dy.shape[2], dy.shape[3]
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    cols = _im2col(xp, kh, kw, oh, ow, stride).reshape(N * oh * ow, C * kh * kw)
    dy2 = dy.transpose(0, 2, 3, 1).reshape(N * oh * ow, F)
    dWf = (dy2.T @ cols).reshape(F, C, kh, kw)
    dW = np.ascontiguousarray(dWf[:, :, ::-1, ::-1])
    db = dy2.sum(axis=0)
    dx = None
    if need_dx:
        Wf = W[:, :, ::-1, ::-1].reshape(F, C * kh * kw)
        dcols = (dy2 @ Wf).reshape(N, oh, ow, C, kh, kw)
        dxp = np.zeros_like(xp)
        for a in range(kh):
            for bb in range(kw):
                dxp[:, :, a:a + oh * stride[0]:stride[0], bb:bb + ow * stride[1]:stride[1]] += \
                    dcols[:, :, :, :, a, bb].transpose(0, 3, 1, 2)
        dx = np.ascontiguousarray(dxp[:, :, ph:ph + H, pw:pw + Wd])
    return dx, dW, db


# --------------------------------------------------------------------------- pool
def maxpool_fwd(x, ds):
    """pool_2d(ds, ignore_border=True, mode='max'), /root/reference/src/net/convpoollayer.py:261.
    Returns (y, ties): ties[..., j] is True where window element j (row-major) EQUALS the window maximum.  Theano's
    gradient op (theano/tensor/signal/pool.py, MaxPoolGrad: `if (a == maximum) gx += gz`) gives the gradient to every
    tied element, which matters here: the constant far-plane background of a depth crop makes whole windows tie."""
    N, C, H, W = x.shape
    ph, pw = ds
    oh, ow = H // ph, W // pw
    xv = x[:, :, :oh * ph, :ow * pw].reshape(N, C, oh, ph, ow, pw).transpose(0, 1, 2, 4, 3, 5)
    xv = xv.reshape(N, C, oh, ow, ph * pw)
    y = xv.max(axis=4)
    ties = xv == y[..., None]
    return np.ascontiguousarray(y), ties


def maxpool_bwd(dy, ties, ds, in_hw):
    N, C, oh, ow = dy.shape
    ph, pw = ds
    H, W = in_hw
    dxv = ties * dy[..., None]
    dxv = dxv.reshape(N, C, oh, ow, ph, pw).transpose(0, 1, 2, 4, 3, 5).reshape(N, C, oh * ph, ow * pw)
    dx = np.zeros((N, C, H, W), dtype=dy.dtype)
    dx[:, :, :oh * ph, :ow * pw] = dxv
    return dx


def convpool_fwd(x, W, b, stride, border_mode, poolsize, relu):
    """ConvPoolLayer: conv -> max-pool -> bias AFTER pooling -> activation,
    /root/reference/src/net/convpoollayer.py:251-282."""
    c = conv2d_fwd(x, W, None, stride, border_mode)
    if tuple(poolsize) == (1, 1):           # poolType = -1, convpoollayer.py:180-181
        p, arg = c, None
    else:
        p, arg = maxpool_fwd(c, poolsize)
    pre = p + b[None, :, None, None]
    y = np.maximum(pre, 0) if relu else pre
    return y, (c.shape, arg, pre)


def convpool_bwd(x, W, dy, cache, stride, border_mode, poolsize, relu, need_dx=True):
    cshape, arg, pre = cache
    g = dy * (pre >= 0) if relu else dy
    db = g.sum(axis=(0, 2, 3))
    dc = g if arg is None else maxpool_bwd(g, arg, poolsize, cshape[2:])
    dx, dW, _ = conv2d_bwd(x, W, dc, stride, border_mode, need_dx=need_dx)
    return dx, dW, db


# --------------------------------------------------------------------------- batch norm
BN_EPS = 1e-4      # /root/reference/src/net/batchnormlayer.py:42
BN_ALPHA = 0.1     # /root/reference/src/net/batchnormlayer.py:42


def bn_fwd_train(x, gamma, beta, eps=BN_EPS):
    """Training mode (flag_on = 1), /root/reference/src/net/batchnormlayer.py:154-155,192.
    mean / BIASED var over axes (0,2,3); inv_std = 1/sqrt(var+eps);
    y = (x - mean) * (gamma * inv_std) + beta.  Returns y, mean, inv_std."""
    axes = (0, 2, 3) if x.ndim == 4 else (0,)
    mean = x.mean(axis=axes)
    var = x.var(axis=axes)
    inv_std = 1.0 / np.sqrt(var + x.dtype.type(np.float32(eps)))   # python floats become floatX constants
    sh = (1, -1, 1, 1) if x.ndim == 4 else (1, -1)
    y = (x - mean.reshape(sh)) * (gamma * inv_std).reshape(sh) + beta.reshape(sh)
    return y, mean, inv_std


def bn_fwd_eval(x, gamma, beta, run_mean, run_inv_std):
    """Deterministic mode (flag_on = 0) uses the stored mean / inv_std, batchnormlayer.py:158-159."""
    sh = (1, -1, 1, 1) if x.ndim == 4 else (1, -1)
    return (x - run_mean.reshape(sh)) * (gamma * run_inv_std).reshape(sh) + beta.reshape(sh)


def bn_running_update(run_mean, run_inv_std, mean, inv_std, alpha=BN_ALPHA):
    """EMA of the mean and of the INVERSE STD (not the variance), batchnormlayer.py:164-172."""
    dt = run_mean.dtype.type
    a = dt(np.float32(alpha))
    oma = dt(np.float32(1.0) - np.float32(alpha))          # (1. - alpha) folded in floatX
    return oma * run_mean + a * mean, oma * run_inv_std + a * inv_std


def bn_bwd_train(x, gamma, mean, inv_std, dy):
    """Gradient through the batch statistics (T.grad of batchnormlayer.py:154-192)."""
    axes = (0, 2, 3) if x.ndim == 4 else (0,)
    sh = (1, -1, 1, 1) if x.ndim == 4 else (1, -1)
    n = x.size // x.shape[1]
    xhat = (x - mean.reshape(sh)) * inv_std.reshape(sh)
    dbeta = dy.sum(axis=axes)
    dgamma = (dy * xhat).sum(axis=axes)
    dx = (gamma * inv_std).reshape(sh) * (dy - dbeta.reshape(sh) / n - xhat * dgamma.reshape(sh) / n)
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------- relu / fc / dropout
def relu_fwd(x):
    """ReLU = T.maximum(x, 0), /root/reference/src/util/theano_helpers.py:61-69."""
    return np.maximum(x, 0)


def relu_bwd(x, dy):
    """Theano Maximum.grad: gradient flows where out == x, i.e. x >= 0."""
    return dy * (x >= 0)


def fc_fwd(x, W, b):
    """HiddenLayer: x . W + b with W of shape (n_in, n_out), /root/reference/src/net/hiddenlayer.py:136-139."""
    return x @ W + b


def fc_bwd(x, W, dy):
    return dy @ W.T, x.T @ dy, dy.sum(axis=0)


DROPOUT_P = 0.3    # /root/reference/src/net/dropoutlayer.py:40


def dropout_fwd_eval(x, p=DROPOUT_P):
    """Deterministic mode: prob_keep * x (NON-inverted dropout), dropoutlayer.py:104."""
    return x.dtype.type(np.float32(1.0 - p)) * x


def dropout_fwd_train(x, mask):
    """Training mode: mask * x with mask ~ Bernoulli(prob_keep), dropoutlayer.py:98-104."""
    return mask * x


# --------------------------------------------------------------------------- loss
def loss_embedding(out, y):
    """numJoints == 1 case, /root/reference/src/trainer/poseregnettrainer.py:94-99:
    cost = mean_n( sum_d (out - y)^2 ).  Returns (cost, dcost/dout)."""
    B = out.shape[0]
    d = out - y
    cost = (d * d).sum(axis=1).mean()
    return cost, (2.0 / B) * d


def loss_joints(out, y, numJoints, nDims):
    """General case, poseregnettrainer.py:97: mean_n mean_j sum_d (.)^2."""
    B = out.shape[0]
    d = out.reshape(B, numJoints, nDims) - y.reshape(B, numJoints, nDims)
    cost = (d * d).sum(axis=2).mean(axis=1).mean()
    return cost, ((2.0 / (B * numJoints)) * d).reshape(out.shape)


def error_embedding(out, y):
    """Monitor, poseregnettrainer.py:117: mean_n sqrt(sum_d (out-y)^2)."""
    return np.sqrt(((out - y) ** 2).sum(axis=1)).mean()


# --------------------------------------------------------------------------- ADAM
def adam_step(params, grads, m, v, t, lr, beta1=0.9, beta2=0.999, epsilon=1e-8, gamma=1 - 1e-8):
    """
    The reference's ADAM with decayed beta1, /root/reference/src/trainer/optimizer.py:58-90.
    t starts at 1.0; all updates use pre-step values; everything in the params' dtype
    (float32 in the reference).  Python-float constants are first rounded to float32,
    as Theano's 'custom' cast policy does for floatX=float32 (so gamma = 1-1e-8 is
    exactly 1.0 and the beta1 decay is a no-op in the reference).  Lists are updated
    in place; returns t + 1.
    """
    beta1, beta2, epsilon, gamma = (float(np.float32(z)) for z in (beta1, beta2, epsilon, gamma))
    for i in range(len(params)):
        dt = params[i].dtype.type
        beta1_t = dt(beta1) * dt(gamma) ** (dt(t) - dt(1.0))
        g = grads[i]
        m_new = beta1_t * m[i] + (dt(1.0) - beta1_t) * g
        v_new = dt(beta2) * v[i] + (dt(1.0) - dt(beta2)) * (g * g)
        m_unb = m_new / (dt(1.0) - dt(beta1) ** dt(t))
        v_unb = v_new / (dt(1.0) - dt(beta2) ** dt(t))
        params[i] = params[i] - (dt(lr) * m_unb) / (np.sqrt(v_unb) + dt(epsilon))
        m[i] = m_new
        v[i] = v_new
    return t + 1.0


def lr_of_ep(learning_rate, ep):
    """NetTrainerParams.lr_of_ep, /root/reference/src/trainer/nettrainer.py:54."""
    if ep <= 1:
        return np.float32(learning_rate / 10.)
    if 1 < ep <= 2:
        return np.float32(learning_rate / 3.)
    return np.float32(learning_rate * np.exp(-0.04 * ep))


# --------------------------------------------------------------------------- metric
def mean_joint_error(gt, pred):
    """HandposeEvaluation.getMeanError, /root/reference/src/util/handpose_evaluation.py:92-97."""
    return np.nanmean(np.nanmean(np.sqrt(np.square(gt - pred).sum(axis=2)), axis=1))


def max_joint_error(gt, pred):
    """getMaxError, handpose_evaluation.py:122-128."""
    return np.nanmax(np.sqrt(np.square(gt - pred).sum(axis=2)))


def joint_mean_error(gt, pred, j):
    """getJointMeanError, handpose_evaluation.py:138-145."""
    return np.nanmean(np.sqrt(np.square(gt[:, j, :] - pred[:, j, :]).sum(axis=1)))
