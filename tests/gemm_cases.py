"""Test helper: every distinct GEMM problem a compiled net launches -- (M, N, K), operand layouts and row maps, tile, split-K,
kernel variant, prologue / epilogue flags, exactly as hipdp.engine's gemm_plan / wgrad_plan emitted them -- re-run stand-alone
through dpp_gemm on random operands and compared with a float64 matmul."""
import numpy as np

from hipdp import ops
from hipdp.lib import RowMap


def gemm_problems(eng):
    seen = {}
    lib = eng.rt.lib
    for _, l in eng.all_launches():
        if l.fn is not lib.dpp_gemm and l.fn is not lib.dpp_fc_gemm:
            continue
        d = l.keep[0]
        # the weight-streaming kernels (dpp_fc_gemm: same descriptor) are keyed with tile (-1, precision, kchunk)
        tile = (d.bm, d.bn, d.wm) if l.fn is lib.dpp_gemm else (-1, l.args[1], l.args[2])
        key = (d.M, d.N, d.K, d.a_kc, d.b_kc, d.lda, d.ldb, d.ldc, d.mapA.s, d.mapB.s, d.mapC.s) + tile + (d.splitk, d.variant,
               d.actA.mode, d.actB.mode, bool(d.bias), bool(d.residual), bool(d.epi.stats), bool(d.epi.bn_x))
        seen.setdefault(key, (l.name, d))
    return seen


def _maprows(mp, n):
    r = np.arange(n)
    if mp.s == 1:
        return r
    nn, q = r // mp.HoWo, r % mp.HoWo
    return nn * mp.HiWi + (q // mp.Wo) * mp.s * mp.Wi + (q % mp.Wo) * mp.s


def _copy_map(mp):
    return RowMap(mp.s, mp.Wo, mp.HoWo, mp.Wi, mp.HiWi)


def _random_act(rng, a):
    """A random prologue with the mode / channel modulus of the recorded one."""
    if a.mode == 0:
        return None
    c = dict(mode=a.mode, cmod=a.cmod, mean=None, scale=None, beta=None)
    if a.mode & 2:
        c['mean'] = rng.normal(0, 0.3, a.cmod).astype(np.float32)
        c['scale'] = rng.uniform(0.5, 1.5, a.cmod).astype(np.float32)
        c['beta'] = rng.normal(0, 0.3, a.cmod).astype(np.float32)
    return c


def _apply_act(X, c):
    if c is None:
        return X.astype(np.float64)
    ch = np.arange(X.shape[1]) % c['cmod']
    v = X.astype(np.float64)
    if c['mode'] & 2:
        v = (X - c['mean'][ch]).astype(np.float64) * c['scale'][ch] + c['beta'][ch]
    if c['mode'] & 1:
        v = np.maximum(v, 0)
    return v


def check_gemm_problem(rt, rng, name, d0, fc=None):
    """Returns None if the problem was checked, or a reason string if it is covered elsewhere.  fc = (precision, kchunk) when the
    launch was a dpp_fc_gemm."""
    if fc is not None and fc[0] != 0:
        return 'bf16 operands: tests/test_gemm.py, tests/test_configs.py'
    if d0.actA.mode == 4:
        return 'mode-4 prologue (lazy BatchNorm backward): tests/test_engine.py'
    if d0.epi.bn_x:
        return 'BatchNorm-backward epilogue: gradient parity tests'
    M, N, K = d0.M, d0.N, d0.K
    rowsA = _maprows(d0.mapA, M if d0.a_kc else K)
    rowsB = np.arange(N) if d0.b_kc else _maprows(d0.mapB, K)
    A = rng.normal(0, 1, (int(rowsA.max()) + 1, d0.lda)).astype(np.float32)
    Bm = rng.normal(0, 1, (int(rowsB.max()) + 1, d0.ldb)).astype(np.float32)
    ca, cb = _random_act(rng, d0.actA), _random_act(rng, d0.actB)
    Aop = _apply_act(A, ca)[rowsA]
    Aop = Aop[:, :K] if d0.a_kc else Aop[:, :M].T
    Bop = _apply_act(Bm, cb)[rowsB]
    Bop = Bop[:, :K].T if d0.b_kc else Bop[:, :N]
    ref = Aop @ Bop
    bias = rng.normal(0, 1, N).astype(np.float32) if d0.bias else None
    crow = _maprows(d0.mapC, M)
    ldc = d0.ldc or N
    nC = int(crow.max()) + 1
    res = rng.normal(0, 1, (nC, ldc)).astype(np.float32) if d0.residual else None
    dA, dB = rt.upload(A), rt.upload(Bm)
    Cb = rt.alloc((nC, ldc), zero=True)
    if res is not None:
        Cb.set(res)                                            # the plans accumulate in place (residual aliases C)
    part = rt.alloc(d0.splitk * M * N, zero=False) if d0.splitk > 1 else None

    def mkact(c):
        if c is None:
            return None
        up = lambda v: rt.upload(v) if v is not None else None          # noqa: E731
        return ops.act(c['mode'], up(c['mean']), up(c['scale']), up(c['beta']), c['cmod'])

    nblk = -(-M // d0.bm) if fc is None else 1
    stats = rt.alloc((2, N, nblk), zero=False) if d0.epi.stats else None
    if fc is not None:
        op = ops.fc_gemm(rt, dA, dB, Cb if d0.splitk == 1 else None, M, N, K, d0.a_kc, d0.b_kc, d0.lda, d0.ldb, ldc, actA=mkact(ca),
                         actB=mkact(cb), bias=rt.upload(bias) if bias is not None else None, residual=Cb if res is not None else None,
                         splitk=d0.splitk, partial=part, precision=fc[0], kchunk=fc[1], name=name)
    else:
        op = ops.gemm(rt, dA, dB, Cb if d0.splitk == 1 else None, M, N, K, d0.a_kc, d0.b_kc, d0.lda, d0.ldb, ldc,
                  mapA=_copy_map(d0.mapA), mapB=_copy_map(d0.mapB), mapC=_copy_map(d0.mapC), actA=mkact(ca), actB=mkact(cb),
                  bias=rt.upload(bias) if bias is not None else None, residual=Cb if res is not None else None, splitk=d0.splitk,
                      partial=part, tile=(d0.bm, d0.bn, d0.wm), epi=ops.epilogue(stats=stats) if stats is not None else None,
                      variant=d0.variant, name=name)
    op(rt.stream)
    rt.synchronize()
    if d0.splitk > 1:
        got, want = part.get().reshape(d0.splitk, M, N).astype(np.float64).sum(axis=0), ref
    else:
        got = Cb.get()[crow][:, :N].astype(np.float64)
        want = ref + (bias if bias is not None else 0) + (res[crow][:, :N] if res is not None else 0)
    # f32 accumulation of K products of O(1) operands: error ~ eps * sqrt(K) * |operand scale|^2, a few times that allowed
    tol = 6e-7 * (np.sqrt(K) + 4) * max(1.0, float(np.abs(Aop).max()) * float(np.abs(Bop).max()))
    err = float(np.abs(got - want).max())
    assert err < tol, (name, (M, N, K), (d0.bm, d0.bn, d0.wm, d0.splitk), err, tol)
    if stats is not None and M % d0.bm == 0:
        st = stats.get()
        blk = got.reshape(nblk, d0.bm, N)
        mean = blk.mean(axis=1)
        np.testing.assert_allclose(st[0].T, mean, rtol=0, atol=1e-5 * max(1.0, np.abs(got).max()), err_msg=name)
        m2 = ((blk - mean[:, None, :]) ** 2).sum(axis=1)
        np.testing.assert_allclose(st[1].T, m2, rtol=5e-5, atol=1e-4 * max(1.0, m2.max()), err_msg=name)
    return None


def check_all(rt, eng, seed=3):
    rng = np.random.RandomState(seed)
    probs = gemm_problems(eng)
    checked, skipped = [], []
    for key, (name, d0) in sorted(probs.items(), key=lambda kv: kv[1][0]):
        why = check_gemm_problem(rt, rng, name, d0, fc=(key[12], key[13]) if key[11] == -1 else None)
        (checked if why is None else skipped).append((name, key))
    return checked, skipped
