"""dpp_gemm with the mode-4 operand (BatchNorm gradient formed from (G, x) while staging) against the plain operand plus the
dpp_bn_bwd_apply launch it replaces, on the 1x1 data-/filter-gradient shapes of the ResNet."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/deep-prior-pp_amd')
import numpy as np, torch
from hipdp import ops
from hipdp.runtime import TorchHipRuntime
rt = TorchHipRuntime()


class BN(object):
    pass


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for label, M, C, Co in (('stage1 conv a', 131072, 16, 64), ('stage2 conv a', 32768, 32, 128), ('stage3 conv a', 8192, 64, 256)):
    G, X, dX = rt.alloc((M, C), zero=False), rt.alloc((M, C), zero=False), rt.alloc((M, C), zero=False)
    rt.tensor(G).normal_(); rt.tensor(X).normal_()
    bn = BN()
    bn.mean, bn.inv_std, bn.scale = rt.alloc(C), rt.alloc(C), rt.alloc(C)
    rt.tensor(bn.inv_std).fill_(1.0); rt.tensor(bn.scale).fill_(1.0)
    q, p, c1, c2 = rt.alloc(C), rt.alloc(C), rt.alloc(C), rt.alloc(C)
    act = ops.act_bn_bwd(bn, q, p, X, C)
    W = rt.alloc((C, Co), zero=False); rt.tensor(W).normal_()
    out = rt.alloc((M, Co), zero=False)
    Xin = rt.alloc((M, Co), zero=False); rt.tensor(Xin).normal_()
    gW = rt.alloc((C, Co), zero=False)
    nb = -(-M // 128)
    apply_ = ops.bn_bwd_apply(rt, G, X, M, C, bn.mean, bn.inv_std, bn.scale, c1, c2, dX, rpb=128, colsum=rt.alloc((nb, C), zero=False))
    for name, a in (('plain', None), ('lazy', act)):
        dg = ops.gemm(rt, G if a is not None else dX, W, out, M, Co, C, 1, 0, C, Co, Co, actA=a, tile=(64, 64, 4))
        splitk = 256 if M >= 32768 else 64
        part = rt.alloc(splitk * C * Co, zero=False)
        wg = ops.gemm(rt, G if a is not None else dX, Xin, None, C, Co, M, 0, 0, C, Co, Co, actA=a, splitk=splitk, partial=part,
                      tile=(16, 64, 1) if C <= 16 else ((32, 64, 1) if C <= 32 else (64, 64, 4)))
        print('%-14s %-5s dgrad %7.2f us   wgrad %7.2f us   (bn_bwd_apply alone %6.2f us)' %
              (label, name, timeit(lambda: dg(rt.stream)), timeit(lambda: wg(rt.stream)), timeit(lambda: apply_(rt.stream))))
