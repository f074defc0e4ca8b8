"""
Prepared kernel launches ("ops") over the C ABI.  An op is built once (descriptor structs filled, device
pointers resolved) and then replayed every step with `op(stream)`; a `Plan` is an ordered list of ops.
Building is host logic only; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from .lib import Act, GemmDesc, RowMap, check


class Launch(object):
    __slots__ = ('fn', 'args', 'keep', 'name')

    def __init__(self, fn, args, keep, name):
        self.fn, self.args, self.keep, self.name = fn, args, keep, name

    def __call__(self, stream):
        st = self.fn(*self.args, stream)
        if st != 0:
            check(st, self.name)


class Plan(object):
    """An ordered list of launches replayed on one stream."""

    def __init__(self, name=''):
        self.name = name
        self.ops = []

    def add(self, op):
        if op is not None:
            self.ops.append(op)
        return op

    def extend(self, ops):
        for o in ops:
            self.add(o)

    def run(self, stream):
        for op in self.ops:
            op(stream)

    def __len__(self):
        return len(self.ops)


def _p(buf):
    return None if buf is None else buf.ptr


def act(mode=0, mean=None, scale=None, beta=None, cmod=1):
    return Act(_p(mean), _p(scale), _p(beta), int(mode), int(cmod))


def gemm(rt, A, B, Cbuf, M, N, K, a_kc, b_kc, lda, ldb, ldc=0, mapA=None, mapB=None, mapC=None, actA=None, actB=None,
         bias=None, residual=None, splitk=1, partial=None, tile=(0, 0, 0), name='gemm'):
    """C = A_op . B_op, see dpp_gemm in include/dpp_hip.h."""
    d = GemmDesc()
    d.A, d.lda, d.a_kc = A.ptr, lda, int(a_kc)
    d.mapA = mapA or RowMap.identity()
    d.actA = actA or Act.none()
    d.B, d.ldb, d.b_kc = B.ptr, ldb, int(b_kc)
    d.mapB = mapB or RowMap.identity()
    d.actB = actB or Act.none()
    d.C, d.ldc = _p(Cbuf), ldc
    d.mapC = mapC or RowMap.identity()
    d.bias, d.residual = _p(bias), _p(residual)
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.splitk, d.partial = int(splitk), _p(partial)
    d.bm, d.bn, d.wm = tile
    return Launch(rt.lib.dpp_gemm, (C.byref(d),), (d, A, B, Cbuf, bias, residual, partial, actA, actB), name)


def reduce_partials(rt, partial, nz, n, out, bias=None, nbias=1, name='reduce_partials'):
    return Launch(rt.lib.dpp_reduce_partials, (partial.ptr, int(nz), int(n), _p(bias), int(nbias), out.ptr),
                  (partial, out, bias), name)
