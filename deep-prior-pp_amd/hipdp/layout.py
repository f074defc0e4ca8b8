"""
Host-side layout conversions between the reference's parameter layouts (what checkpoints and
`get_value()/set_value()` expose) and the kernel layouts of libdpp_hip.so.  Pure index permutations.

  conv W      reference (F, C, kh, kw), Theano true convolution   <->  Wk[F][kh*kw][C], flipped: the kernels
              correlate, so Wk[o][a*kw + b][c] = W[o][c][kh-1-a][kw-1-b]   (convlayer.py:230-235, filter_flip)
  FC after a conv map: the reference flattens NCHW (row index c*H*W + y*W + x, resnet.py:141); the kernels keep
              activations NHWC, so the rows of that W are permuted to (y*W + x)*C + c.
"""
import numpy as np


def conv_w_to_kernel(W):
    F, Cc, kh, kw = W.shape
    return np.ascontiguousarray(W[:, :, ::-1, ::-1].transpose(0, 2, 3, 1).reshape(F, kh * kw, Cc))


def conv_w_from_kernel(Wk, shape):
    F, Cc, kh, kw = shape
    return np.ascontiguousarray(np.asarray(Wk).reshape(F, kh, kw, Cc).transpose(0, 3, 1, 2)[:, :, ::-1, ::-1])


def fc_rows_nchw_to_nhwc(W, Cc, H, Wd):
    n_out = W.shape[1]
    return np.ascontiguousarray(W.reshape(Cc, H, Wd, n_out).transpose(1, 2, 0, 3).reshape(Cc * H * Wd, n_out))


def fc_rows_nhwc_to_nchw(W, Cc, H, Wd):
    n_out = W.shape[1]
    return np.ascontiguousarray(np.asarray(W).reshape(H, Wd, Cc, n_out).transpose(2, 0, 1, 3).reshape(Cc * H * Wd, n_out))


def nchw_to_nhwc(x):
    return np.ascontiguousarray(np.asarray(x).transpose(0, 2, 3, 1))


def nhwc_to_nchw(x):
    return np.ascontiguousarray(np.asarray(x).transpose(0, 3, 1, 2))


def fc_rows(W, info, fn):
    """Row permutation between the reference's NCHW flatten order and the NHWC order of the activations: one (C, H, W) block, or
    several side by side when the FC input is a concatenation of flattened maps (ScaleNet)."""
    if isinstance(info[0], int):
        return fn(W, *info)
    out, o = [], 0
    for (Cc, H, Wd) in info:
        n = Cc * H * Wd
        out.append(fn(W[o:o + n], Cc, H, Wd) if H * Wd > 1 else W[o:o + n])
        o += n
    assert o == W.shape[0]
    return np.concatenate(out, axis=0)


def from_kernel(kind, info, shape, flat):
    """A parameter's value in the reference's layout from its kernel-layout slice of the flat device buffer (kind: 'conv_w' | 'fc_w' |
    'vec'; info: the (C, H, W) block(s) of an FC behind a conv map, else None)."""
    if kind == 'conv_w':
        return conv_w_from_kernel(flat, shape)
    if kind == 'fc_w' and info is not None:
        return fc_rows(flat.reshape(shape), info, fc_rows_nhwc_to_nchw)
    return flat.reshape(shape).copy()
