"""
Prepared kernel launches ("ops") over the C ABI.  An op is built once (descriptor structs filled, device
pointers resolved) and then replayed every step with `op(stream)`; a `Plan` is an ordered list of ops.
Building is host logic only; nothing here computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

from .lib import ST_A, ST_B, ST_BNX, ST_C, Act, BnEval, DppError, Epilogue, GemmDesc, ResblockDesc, RowMap, check


class Launch(object):
    """One prepared kernel launch.  `meta` = dict(kernel=<family>, flops=<algorithmic flops>, bytes=<algorithmic HBM
    bytes>) is what bench.py's roofline accounting reads."""
    __slots__ = ('fn', 'args', 'keep', 'name', 'meta', 'kernels')

    def __init__(self, fn, args, keep, name, meta=None, kernels=1):
        self.fn, self.args, self.keep, self.name, self.meta = fn, args, keep, name, meta
        self.kernels = kernels         # kernels the entry point issues (the plan recorder checks what it captured against this)

    def __call__(self, stream):
        st = self.fn(*self.args, stream)
        if st != 0:
            check(st, self.name)


class Fork(object):
    """Marker: from here on, launches added with side=True run on the side stream, which first waits for everything
    issued so far on the main stream."""
    name = 'fork'
    meta = None


class Join(object):
    """Marker: the main stream waits for everything issued so far on the side stream."""
    name = 'join'
    meta = None


# How a Plan issues its launches when it runs on a runtime:
#   native (default)  recorded once into a dpp_plan (include/dpp_hip.h) and re-issued from C++ on the two HIP streams;
#   graph             the same recording replayed as an explicit hipGraph (lanes = parallel branches);
#   graph1            one-lane hipGraph (everything chained in recorded order);
#   python            every launch is a ctypes call from the interpreter (what round 1 measured: host-bound).
LAUNCH_MODE = os.environ.get('DPP_LAUNCH_MODE', 'native')


class NativePlan(object):
    """A run of launches / fork / join markers recorded into a dpp_plan.  Recording calls every prepared op once with the
    library in recording mode: the C side keeps the resolved kernel, grid and a private copy of the arguments."""

    def __init__(self, rt, items, mode='native'):
        self.rt, self.lib, self.items, self.mode = rt, rt.lib, items, mode       # items keep the device buffers alive
        h = C.c_void_p()
        check(self.lib.dpp_plan_create(C.byref(h)), 'dpp_plan_create')
        self.handle = h
        lib = self.lib
        check(lib.dpp_plan_record_begin(h), 'dpp_plan_record_begin')
        try:
            lane = 0
            for op, side in items:
                if isinstance(op, Fork):
                    check(lib.dpp_plan_fork(h), 'dpp_plan_fork')
                elif isinstance(op, Join):
                    check(lib.dpp_plan_join(h), 'dpp_plan_join')
                else:
                    if int(bool(side)) != lane:
                        lane = int(bool(side))
                        check(lib.dpp_plan_record_lane(h, lane), 'dpp_plan_record_lane')
                    op(None)
        finally:
            check(lib.dpp_plan_record_end(h), 'dpp_plan_record_end')
        n = C.c_int()
        check(lib.dpp_plan_count(h, C.byref(n), None, None), 'dpp_plan_count')
        want = sum(op.kernels for op, _ in items if isinstance(op, Launch))
        if n.value != want:
            raise DppError("plan recording captured %d launches, expected %d" % (n.value, want))
        self.graph_ready = False

    def run(self, rt):
        two = getattr(rt, 'has_side_stream', False)
        if self.mode in ('graph', 'graph1') and not getattr(rt, 'is_emulator', False):
            if not self.graph_ready:
                check(self.lib.dpp_plan_graph_build(self.handle, 1 if (self.mode == 'graph' and two) else 0), 'dpp_plan_graph_build')
                self.graph_ready = True
            check(self.lib.dpp_plan_graph_launch(self.handle, rt.stream), 'dpp_plan_graph_launch')
            return
        st = self.lib.dpp_plan_run(self.handle, rt.stream, rt.side_stream if two else None)
        if st != 0:
            check(st, 'dpp_plan_run')

    def __del__(self):
        try:
            if self.handle:
                self.lib.dpp_plan_destroy(self.handle)
                self.handle = None
        except Exception:          # noqa: BLE001  (interpreter shutdown)
            pass


class _OpList(list):
    """The (op, side) list of a Plan.  Every mutation bumps `version`: a compiled plan stays valid exactly as long as the list it was
    recorded from is untouched -- the engine splices plans in place (Plan.ops[0:0] = ..., pop / insert) -- without rebuilding a
    400-tuple identity key on every step."""
    version = 0

    def _bump(name):
        base = getattr(list, name)

        def method(self, *a, **k):
            self.version += 1
            return base(self, *a, **k)
        method.__name__ = name
        return method

    for _n in ('append', 'extend', 'insert', 'pop', 'remove', 'clear', 'sort', 'reverse', '__setitem__', '__delitem__', '__iadd__', '__imul__'):
        locals()[_n] = _bump(_n)
    del _n, _bump


class Plan(object):
    """An ordered list of launches.  Every launch belongs to the main stream or (side=True) to an auxiliary stream that
    the runtime provides; Fork / Join markers order the two (event wait).  Run on a runtime, consecutive launches are
    compiled into NativePlans (one C call each); steps that are not kernel launches (collectives) stay Python calls
    between them."""

    def __init__(self, name=''):
        self.name = name
        self._ops = _OpList()  # (op, side)
        self.uses_side = False
        self._compiled = None

    @property
    def ops(self):
        return self._ops

    @ops.setter
    def ops(self, value):          # assigning a new list replaces the recording's source as well
        new = _OpList(value)
        new.version = self._ops.version + 1
        self._ops = new

    @staticmethod
    def concat(name, plans):
        out = Plan(name)
        for p in plans:
            out.ops.extend(p.ops)
            out.uses_side = out.uses_side or p.uses_side
        return out

    def add(self, op, side=False):
        if op is not None:
            self.ops.append((op, side))
            self.uses_side = self.uses_side or side
        return op

    def fork(self):
        self.ops.append((Fork(), False))

    def join(self):
        self.ops.append((Join(), False))

    def launches(self):
        return [o for (o, _) in self.ops if isinstance(o, Launch)]

    def steps(self):
        return [o for (o, _) in self.ops if not isinstance(o, (Fork, Join))]

    def _compile(self, rt):
        # keyed on the identity of the ops: a plan whose ops are replaced or reordered (the engine splices plans) must be re-recorded,
        # a recording freezes each launch's grid and arguments
        key = (id(rt), LAUNCH_MODE, id(self._ops), self._ops.version)
        if self._compiled is not None and self._compiled[0] == key:
            return self._compiled[1]
        segs, cur = [], []
        for op, side in self.ops:
            if isinstance(op, (Launch, Fork, Join)):
                cur.append((op, side))
            else:
                if cur:
                    segs.append(NativePlan(rt, cur, LAUNCH_MODE))
                    cur = []
                segs.append(op)
        if cur:
            segs.append(NativePlan(rt, cur, LAUNCH_MODE))
        self._compiled = (key, segs)
        return segs

    def segment_counts(self):
        """(native launch segments, host-issued steps) of this plan: a step that is not a kernel launch -- an RCCL collective -- ends the
        native segment in front of it and is issued from Python between two dpp_plan_run calls."""
        native, host, open_ = 0, 0, False
        for op, _ in self.ops:
            if isinstance(op, (Launch, Fork, Join)):
                if not open_:
                    native, open_ = native + 1, True
            else:
                host, open_ = host + 1, False
        return native, host

    def run(self, rt_or_stream):
        """rt_or_stream: a runtime (multi-stream aware) or a raw stream handle (single stream, side ops inline)."""
        rt = rt_or_stream if hasattr(rt_or_stream, 'stream') else None
        if rt is not None and LAUNCH_MODE != 'python':
            for seg in self._compile(rt):
                if isinstance(seg, NativePlan):
                    seg.run(rt)
                else:
                    seg(rt.stream)
            return
        if rt is None or not self.uses_side or not getattr(rt, 'has_side_stream', False):
            st = rt.stream if rt is not None else rt_or_stream
            for op, _ in self.ops:
                if not isinstance(op, (Fork, Join)):
                    op(st)
            return
        main, side = rt.stream, rt.side_stream
        for op, on_side in self.ops:
            if isinstance(op, Fork):
                rt.side_wait_main()
            elif isinstance(op, Join):
                rt.main_wait_side()
            else:
                op(side if on_side else main)

    def __len__(self):
        return len(self.launches())


def _p(buf):
    return None if buf is None else buf.ptr


BF16 = np.dtype(np.uint16)          # storage dtype of a bf16 activation tensor (runtime buffers have no bfloat16: the bits travel as uint16)


def _is16(buf):
    """Does this buffer hold a bf16-stored activation tensor (DPP_ST_* of include/dpp_hip.h)?"""
    return buf is not None and getattr(buf, 'dtype', None) == BF16


def _store(a=None, b=None, c=None, bnx=None):
    """DPP_ST_* mask of a call from the dtypes of the buffers in its A / B / C (+ residual) / epilogue.bn_x roles."""
    return (ST_A if _is16(a) else 0) | (ST_B if _is16(b) else 0) | (ST_C if _is16(c) else 0) | (ST_BNX if _is16(bnx) else 0)


def _esz(buf):
    return 2.0 if _is16(buf) else 4.0


def act(mode=0, mean=None, scale=None, beta=None, cmod=1, x2=None, aux=None, out=None):
    a = Act(_p(mean), _p(scale), _p(beta), int(mode), int(cmod), _p(x2), _p(aux), _p(out))
    a._keep = (mean, scale, beta, x2, aux, out)
    return a


def act_bn_bwd(bn, q, p, x, C, out=None):
    """Operand prologue of dpp_gemm's A in mode 4: the operand is the masked gradient G of BatchNorm `bn`, and the value used
    is dX = scale*G - p*(x - mean) - q (q = scale*c1, p = scale*inv_std*c2 from bn_bwd_finalize): bn_bwd_apply on the fly."""
    return act(Act.BN_BWD, mean=bn.mean, scale=bn.scale, beta=q, cmod=C, x2=x, aux=p, out=out)


def epilogue(stats=None, bn=None, bn_x=None, bn_relu=True, bn_partial=None):
    """dpp_epilogue: fused BatchNorm statistics (`stats`) and/or BatchNorm-backward mask + sums (`bn` = an object with
    mean / inv_std / scale / beta_buf buffers, `bn_x` the BatchNorm input, `bn_partial` the per-block sums)."""
    e = Epilogue()
    e.stats = _p(stats)
    if bn is not None:
        e.bn_x, e.bn_mean, e.bn_inv_std, e.bn_scale, e.bn_beta = bn_x.ptr, bn.mean.ptr, bn.inv_std.ptr, bn.scale.ptr, bn.beta_buf.ptr
        e.bn_relu, e.bn_partial = int(bn_relu), bn_partial.ptr
    e._bn_x = bn_x
    e._keep = (stats, bn, bn_x, bn_partial)
    return e


def gemm(rt, A, B, Cbuf, M, N, K, a_kc, b_kc, lda, ldb, ldc=0, mapA=None, mapB=None, mapC=None, actA=None, actB=None,
         bias=None, residual=None, splitk=1, partial=None, tile=(0, 0, 0), epi=None, variant=0, name='gemm', precision=0):
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
    d.variant = int(variant)
    if epi is not None:
        d.epi = epi
    if _is16(residual) != _is16(Cbuf) and residual is not None and Cbuf is not None:
        raise ValueError("dpp_gemm: C and residual must be stored alike")
    d.store = _store(A, B, Cbuf, getattr(epi, '_bn_x', None) if epi is not None else None)
    d.precision = int(precision)
    meta = dict(kernel='gemm_mfma_bf16' if precision else 'gemm_mfma_f32', flops=2.0 * M * N * K,
                bytes=_esz(A) * M * K + _esz(B) * K * N + (4.0 * M * N * max(1, splitk) if splitk > 1 else _esz(Cbuf) * M * N) +
                (_esz(residual) * M * N if residual is not None else 0))
    return Launch(rt.lib.dpp_gemm, (C.byref(d),), (d, A, B, Cbuf, bias, residual, partial, actA, actB, epi), name, meta)


def gemm_variant_rows(rt, launch):
    """dpp_gemm_variant_rows for a launch built by gemm(): rows per workgroup of the kernel its `variant` asks for, or 0 when the C
    side would answer DPP_E_UNSUPPORTED (alignment, prologue mode, epilogue) -- the caller then describes it with variant 0."""
    return int(rt.lib.dpp_gemm_variant_rows(C.byref(launch.keep[0])))


def wgrad_stream(rt, dY, Co, X, Ci, M, rows_per_wave, partial, mapX=None, actX=None, name='wgrad_stream', precision=0):
    """dpp_wgrad_stream: the filter gradient of a 1x1 convolution as per-slice partials [slices][Co][Ci] (precision 1: bf16 MFMA operands,
    dpp_wgrad_stream_bf16, the shapes dpp_wgrad_stream_bf16_ok accepts)."""
    nsl = rt.lib.dpp_wgrad_stream_slices(Co, Ci, M, rows_per_wave)
    meta = dict(kernel='gemm_mfma_bf16' if precision else 'gemm_mfma_f32', flops=2.0 * M * Co * Ci,
                bytes=_esz(dY) * M * Co + 4.0 * nsl * Co * Ci + _esz(X) * M * Ci)
    return Launch(rt.lib.dpp_wgrad_stream_bf16 if precision else rt.lib.dpp_wgrad_stream, (dY.ptr, int(Co), X.ptr, int(Ci), C.byref(mapX) if mapX is not None else None, _actp(actX), int(M),
                                            int(rows_per_wave), partial.ptr, _store(a=dY, b=X)), (dY, X, partial, mapX, actX), name, meta)


def wgrad3_stream(rt, dY, Co, X, Ci, N, H, W, rows_per_wave, partial, actX=None, name='conv3x3_wgrad_stream'):
    """dpp_wgrad3_stream: the filter gradient of a 3x3 convolution as per-slice partials [slices][Co][9][Ci]."""
    nsl = rt.lib.dpp_wgrad3_stream_slices(Co, Ci, N, H, W, rows_per_wave)
    px = float(N) * H * W
    meta = dict(kernel='conv3x3_wgrad_mfma_f32', flops=2.0 * px * 9 * Ci * Co, bytes=4.0 * (px * Co + nsl * 9.0 * Ci * Co) + _esz(X) * px * Ci)
    return Launch(rt.lib.dpp_wgrad3_stream, (dY.ptr, int(Co), X.ptr, int(Ci), int(N), int(H), int(W), _actp(actX), int(rows_per_wave),
                                             partial.ptr, _store(a=dY, b=X)), (dY, X, partial, actX), name, meta)


def fc_gemm(rt, A, B, Cbuf, M, N, K, a_kc, b_kc, lda, ldb, ldc=0, actA=None, actB=None, bias=None, residual=None, splitk=1, partial=None,
            precision=0, kchunk=0, name='fc_gemm'):
    """dpp_fc_gemm: dpp_gemm's contract on the weight-streaming kernel (f32 or bf16 operands), see include/dpp_hip.h."""
    d = GemmDesc()
    d.A, d.lda, d.a_kc = A.ptr, lda, int(a_kc)
    d.mapA, d.mapB, d.mapC = RowMap.identity(), RowMap.identity(), RowMap.identity()
    d.actA = actA or Act.none()
    d.B, d.ldb, d.b_kc = B.ptr, ldb, int(b_kc)
    d.actB = actB or Act.none()
    d.C, d.ldc = _p(Cbuf), ldc
    d.bias, d.residual = _p(bias), _p(residual)
    d.M, d.N, d.K = int(M), int(N), int(K)
    d.splitk, d.partial = int(splitk), _p(partial)
    d.store = _store(A, B, Cbuf)
    if _is16(residual) != _is16(Cbuf) and residual is not None and Cbuf is not None:
        raise ValueError("dpp_fc_gemm: C and residual must be stored alike")
    meta = dict(kernel='fc_gemm_mfma_bf16' if precision else 'gemm_mfma_f32', flops=2.0 * M * N * K,
                bytes=4.0 * (M * K + K * N + M * N * (max(1, splitk) if splitk > 1 else 1) + (M * N if residual is not None else 0)))
    return Launch(rt.lib.dpp_fc_gemm, (C.byref(d), int(precision), int(kchunk)), (d, A, B, Cbuf, bias, residual, partial, actA, actB), name, meta)


def fc_wgrad_stream(rt, X, dY, dW, Nb, K, N, actX=None, name='fc_wgrad_stream'):
    """dpp_fc_wgrad_stream: dW [K][N] = act(X)^T . dY over the Nb rows, each output block owned by one wave (no partials)."""
    meta = dict(kernel='gemm_mfma_f32', flops=2.0 * Nb * K * N, bytes=4.0 * (Nb * K + Nb * N + K * N))
    return Launch(rt.lib.dpp_fc_wgrad_stream, (X.ptr, dY.ptr, dW.ptr, int(Nb), int(K), int(N), _actp(actX), _store(b=X)), (X, dY, dW, actX), name, meta)


def reduce_partials(rt, partial, nz, n, out, bias=None, nbias=1, name='reduce_partials'):
    return Launch(rt.lib.dpp_reduce_partials, (partial.ptr, int(nz), int(n), _p(bias), int(nbias), out.ptr),
                  (partial, out, bias), name, dict(kernel='reduce_partials', flops=float(nz) * n, bytes=4.0 * (nz + 1) * n))


def _actp(a):
    return C.byref(a) if a is not None else None


def conv3x3(rt, X, N, H, W, Ci, Wk, Co, Y, actX=None, bias=None, residual=None, bm=0, epi=None, name='conv3x3', precision=0):
    px = float(N) * H * W
    meta = dict(kernel='conv3x3_mfma_bf16' if precision else 'conv3x3_mfma_f32', flops=2.0 * px * 9 * Ci * Co,
                bytes=px * (_esz(X) * Ci + _esz(Y) * Co + (_esz(residual) * Co if residual is not None else 0)) + 4.0 * 9 * Ci * Co)
    if residual is not None and _is16(residual) != _is16(Y):
        raise ValueError("dpp_conv3x3: Y and residual must be stored alike")
    st = _store(X, None, Y, getattr(epi, '_bn_x', None) if epi is not None else None)
    return Launch(rt.lib.dpp_conv3x3_bf16 if precision else rt.lib.dpp_conv3x3, (X.ptr, N, H, W, Ci, _actp(actX), Wk.ptr, Co, _p(bias), _p(residual), Y.ptr, bm,
                                       C.byref(epi) if epi is not None else None, st),
                  (X, Wk, Y, actX, bias, residual, epi), name, meta)


def conv3x3_stream(rt, X, N, H, W, Cc, Wk, Y, actX=None, bias=None, epi=None, name='conv3x3'):
    """dpp_conv3x3_stream: the 3x3 convolution of a narrow square layer on the barrier-free kernel (float32 tensors only)."""
    _f32_only('dpp_conv3x3_stream', X, Y, getattr(epi, '_bn_x', None) if epi is not None else None)
    px = float(N) * H * W
    meta = dict(kernel='conv3x3_mfma_f32', flops=2.0 * px * 9 * Cc * Cc, bytes=px * 8.0 * Cc + 4.0 * 9 * Cc * Cc)
    return Launch(rt.lib.dpp_conv3x3_stream, (X.ptr, N, H, W, Cc, _actp(actX), Wk.ptr, _p(bias), Y.ptr, C.byref(epi) if epi is not None else None),
                  (X, Wk, Y, actX, bias, epi), name, meta)


def conv3x3_wtrans(rt, Wk, Co, Ci, Wd, name='conv3x3_wtrans'):
    return Launch(rt.lib.dpp_conv3x3_wtrans, (Wk.ptr, Co, Ci, Wd.ptr), (Wk, Wd), name)


def conv3x3_wtrans_multi(rt, jobs, name='conv3x3_wtrans'):
    """jobs: [(Wk buffer, Co, Ci, Wd buffer)]: the mirrored data-gradient weights of every 3x3 layer in ONE launch."""
    import struct
    assert rt.lib.dpp_wtrans_job_bytes() == 32
    raw, block0 = b'', 0
    for (Wk, Co, Ci, Wd) in jobs:
        raw += struct.pack('<QQiiii', Wk.ptr, Wd.ptr, int(Co), int(Ci), block0, 0)
        block0 += -(-(Co * 9 * Ci) // 256)
    table = rt.upload(np.frombuffer(raw, np.uint8).copy())
    return Launch(rt.lib.dpp_conv3x3_wtrans_multi, (table.ptr, len(jobs), block0), (table, list(jobs)), name)


def conv3x3_wgrad(rt, X, N, H, W, Ci, dY, Co, partial, actX=None, bm=64, name='conv3x3_wgrad', precision=0):
    """precision 1: bf16 MFMA operands (dpp_conv3x3_wgrad_bf16; layers dpp_conv3x3_wgrad_bf16_ok accepts)."""
    px = float(N) * H * W
    nblk = rt.lib.dpp_conv3x3_wgrad_blocks(N, H, W, Ci, Co, bm)
    meta = dict(kernel='conv3x3_wgrad_mfma_bf16' if precision else 'conv3x3_wgrad_mfma_f32', flops=2.0 * px * 9 * Ci * Co,
                bytes=_esz(dY) * px * Co + 4.0 * nblk * 9.0 * Ci * Co + _esz(X) * px * Ci)
    fn = rt.lib.dpp_conv3x3_wgrad_bf16 if precision else rt.lib.dpp_conv3x3_wgrad
    return Launch(fn, (X.ptr, N, H, W, Ci, _actp(actX), dY.ptr, Co, partial.ptr, bm, _store(a=X, b=dY)),
                  (X, dY, partial, actX), name, meta)


def stem_fwd(rt, X, N, H, W, Wk, bias, Co, Y, argmax, stats=None, name='stem_fwd'):
    px = float(N) * H * W
    return Launch(rt.lib.dpp_stem_fwd, (X.ptr, N, H, W, Wk.ptr, bias.ptr, Co, Y.ptr, _p(argmax), _p(stats), _store(c=Y)), (X, Wk, bias, Y, argmax, stats), name,
                  dict(kernel='stem_fwd_mfma_f32', flops=2.0 * px * 25 * Co, bytes=4.0 * px + px / 4 * Co * (1.0 + _esz(Y))))


def stem_wgrad(rt, X, N, H, W, dY, argmax, Co, partial, tiles_per_block, name='stem_wgrad'):
    px = float(N) * H * W
    return Launch(rt.lib.dpp_stem_wgrad, (X.ptr, N, H, W, dY.ptr, argmax.ptr, Co, partial.ptr, tiles_per_block),
                  (X, dY, argmax, partial), name, dict(kernel='stem_wgrad', flops=2.0 * px / 4 * Co * 25, bytes=4.0 * px + px / 4 * Co * 5.0))


def convpool_fwd(rt, X, N, H, W, Ci, Wk, kh, kw, pad, Co, pool, bias, Y, ties=None, actX=None, name='convpool_fwd'):
    px = float(N) * (H + 2 * pad - kh + 1) * (W + 2 * pad - kw + 1)
    return Launch(rt.lib.dpp_convpool_fwd, (X.ptr, N, H, W, Ci, _actp(actX), Wk.ptr, kh, kw, pad, Co, pool, bias.ptr, Y.ptr, _p(ties)),
                  (X, Wk, bias, Y, ties, actX), name,
                  dict(kernel='convpool_fwd', flops=2.0 * px * kh * kw * Ci * Co, bytes=4.0 * N * H * W * Ci + 6.0 * px / (pool * pool) * Co))


def convpool_wgrad(rt, X, N, H, W, Ci, dY, ties, kh, kw, pad, Co, pool, partial, actX=None, name='convpool_wgrad'):
    px = float(N) * (H + 2 * pad - kh + 1) * (W + 2 * pad - kw + 1) / (pool * pool)
    return Launch(rt.lib.dpp_convpool_wgrad, (X.ptr, N, H, W, Ci, _actp(actX), dY.ptr, _p(ties), kh, kw, pad, Co, pool, partial.ptr),
                  (X, dY, ties, partial, actX), name,
                  dict(kernel='convpool_wgrad', flops=2.0 * px * kh * kw * Ci * Co, bytes=4.0 * N * H * W * Ci + 6.0 * px * Co))


def convpool_dgrad(rt, dY, ties, N, H, W, Ci, Wk, kh, kw, pad, Co, pool, dX, name='convpool_dgrad'):
    px = float(N) * (H + 2 * pad - kh + 1) * (W + 2 * pad - kw + 1) / (pool * pool)
    return Launch(rt.lib.dpp_convpool_dgrad, (dY.ptr, _p(ties), N, H, W, Ci, Wk.ptr, kh, kw, pad, Co, pool, dX.ptr), (dY, ties, Wk, dX), name,
                  dict(kernel='convpool_dgrad', flops=2.0 * px * kh * kw * Ci * Co, bytes=4.0 * N * H * W * Ci + 6.0 * px * Co))


def bn_stats_partial(rt, X, M, Cc, rpb, partial, name='bn_stats_partial'):
    return Launch(rt.lib.dpp_bn_stats_partial, (X.ptr, M, Cc, rpb, partial.ptr, _store(a=X)), (X, partial), name,
                  dict(kernel='bn_stats_partial', flops=3.0 * M * Cc, bytes=_esz(X) * M * Cc))


def bn_finalize(rt, partial, nb, M, rpb, Cc, gamma, eps, mean, inv_std, scale, run_mean=None, run_inv_std=None, alpha=0.0,
                nseg=1, name='bn_finalize'):
    """partial: nseg segments of [2][Cc][nb] (nb blocks per segment), M rows over all segments."""
    return Launch(rt.lib.dpp_bn_finalize, (partial.ptr, nb, int(nseg), M, rpb, Cc, gamma.ptr, float(eps), mean.ptr, inv_std.ptr, scale.ptr,
                                           _p(run_mean), _p(run_inv_std), float(alpha)),
                  (partial, gamma, mean, inv_std, scale, run_mean, run_inv_std), name,
                  dict(kernel='bn_finalize', flops=0.0, bytes=8.0 * nb * Cc))


def bn_eval_coeffs(rt, gamma, run_mean, run_inv_std, Cc, mean, inv_std, scale, name='bn_eval_coeffs'):
    return Launch(rt.lib.dpp_bn_eval_coeffs, (gamma.ptr, run_mean.ptr, run_inv_std.ptr, Cc, mean.ptr, inv_std.ptr, scale.ptr),
                  (gamma, run_mean, run_inv_std, mean, inv_std, scale), name)


def bn_eval_coeffs_multi(rt, jobs, name='bn_eval_coeffs'):
    """jobs: [(gamma, run_mean, run_inv_std, C, mean, inv_std, scale)] -- dpp_bn_eval_coeffs for every BatchNorm of a net in ONE launch."""
    import struct
    assert rt.lib.dpp_bn_eval_job_bytes() == 56
    raw, block0 = b'', 0
    for (gamma, rm, ris, Cc, mean, inv_std, scale) in jobs:
        raw += struct.pack('<QQQQQQii', gamma.ptr, rm.ptr, ris.ptr, mean.ptr, inv_std.ptr, scale.ptr, int(Cc), block0)
        block0 += -(-int(Cc) // 256)
    table = rt.upload(np.frombuffer(raw, np.uint8).copy())
    return Launch(rt.lib.dpp_bn_eval_coeffs_multi, (table.ptr, len(jobs), block0), (table, list(jobs)), name,
                  dict(kernel='bn_eval_coeffs', flops=0.0, bytes=24.0 * sum(int(j[3]) for j in jobs)))


def bn_eval(mean, inv_std, gamma, beta):
    """dpp_bn_eval: a BatchNorm in deterministic mode as the fused block kernel reads it (stored statistics + affine parameters)."""
    b = BnEval(mean.ptr, inv_std.ptr, gamma.ptr, beta.ptr)
    b._keep = (mean, inv_std, gamma, beta)
    return b


def resblock_eval(rt, X, N, H, W, Cin, stride, Cout, Nb, bn0, bn1, bn2, W1, b1, W2, b2, W3, b3, Y, Wsc=None, bsc=None, name='resblock_eval'):
    """dpp_resblock_eval: one pre-activation bottleneck block of the deterministic forward pass in one launch (csrc/resblock.hip)."""
    d = ResblockDesc()
    d.store = _store(a=X, c=Y)
    Ho, Wo = -(-H // stride), -(-W // stride)
    d.X, d.N, d.H, d.W, d.Cin = X.ptr, int(N), int(H), int(W), int(Cin)
    d.stride, d.Ho, d.Wo, d.Cout, d.Nb = int(stride), Ho, Wo, int(Cout), int(Nb)
    d.bn0, d.bn1, d.bn2 = bn0, bn1, bn2
    d.W1, d.b1, d.W2, d.b2, d.W3, d.b3 = W1.ptr, b1.ptr, W2.ptr, b2.ptr, W3.ptr, b3.ptr
    d.Wsc, d.bsc, d.Y = _p(Wsc), _p(bsc), Y.ptr
    px = float(N) * Ho * Wo
    flops = 2.0 * px * (Cin * Nb + 9.0 * Nb * Nb + Nb * Cout + (Cin * Cout if Wsc is not None else 0))
    byts = _esz(X) * (float(N) * H * W * Cin + (px * Cout if Wsc is None else 0)) + _esz(Y) * px * Cout + 4.0 * (Cin * Nb + 9.0 * Nb * Nb + Nb * Cout)
    return Launch(rt.lib.dpp_resblock_eval, (C.byref(d),), (d, X, Y, bn0, bn1, bn2, W1, b1, W2, b2, W3, b3, Wsc, bsc), name,
                  dict(kernel='resblock_eval_mfma_f32', flops=flops, bytes=byts))


def bn_bwd_reduce(rt, dA, X, M, Cc, mean, inv_std, scale, beta, relu, G, rpb, partial, name='bn_bwd_reduce'):
    return Launch(rt.lib.dpp_bn_bwd_reduce, (dA.ptr, X.ptr, M, Cc, mean.ptr, inv_std.ptr, scale.ptr, beta.ptr, int(relu), G.ptr, rpb,
                                             partial.ptr, _store(a=dA, c=G, bnx=X)), (dA, X, mean, inv_std, scale, beta, G, partial), name,
                  dict(kernel='bn_bwd_reduce', flops=8.0 * M * Cc, bytes=(8.0 + _esz(X)) * M * Cc))


def bn_bwd_finalize(rt, partial, nb, M, Cc, dbeta, dgamma, c1, c2, nseg=1, bn=None, q=None, p=None, name='bn_bwd_finalize'):
    """q, p (with bn): also write the constants of the mode-4 operand prologue (act_bn_bwd)."""
    return Launch(rt.lib.dpp_bn_bwd_finalize, (partial.ptr, nb, int(nseg), M, Cc, dbeta.ptr, dgamma.ptr, c1.ptr, c2.ptr,
                                               _p(bn.inv_std) if q is not None else None, _p(bn.scale) if q is not None else None,
                                               _p(q), _p(p)),
                  (partial, dbeta, dgamma, c1, c2, bn, q, p), name, dict(kernel='bn_bwd_finalize', flops=0.0, bytes=8.0 * nb * Cc))


def bn_bwd_apply(rt, G, X, M, Cc, mean, inv_std, scale, c1, c2, dX, add=None, rpb=None, colsum=None, name='bn_bwd_apply'):
    rpb = rpb or max(32, -(-M // 1024))
    if add is not None and _is16(add) != _is16(dX):
        raise ValueError("dpp_bn_bwd_apply: dX and the gradient added to it must be stored alike")
    return Launch(rt.lib.dpp_bn_bwd_apply, (G.ptr, X.ptr, M, Cc, mean.ptr, inv_std.ptr, scale.ptr, c1.ptr, c2.ptr, _p(add), dX.ptr,
                                            int(rpb), _p(colsum), _store(a=G, c=dX, bnx=X)),
                  (G, X, mean, inv_std, scale, c1, c2, add, dX, colsum), name,
                  dict(kernel='bn_bwd_apply', flops=6.0 * M * Cc, bytes=((12.0 if add is not None else 8.0) + _esz(X)) * M * Cc))


def bn_bwd_finalize_apply(rt, G, X, M, Cc, mean, inv_std, scale, partial, nb, dX, dbeta, dgamma, add=None, rpb=32, colsum=None,
                          name='bn_bwd_finalize_apply'):
    """dpp_bn_bwd_finalize_apply: the finalize of the per-block sums and the apply pass of a BatchNorm backward in one launch."""
    if add is not None and _is16(add) != _is16(dX):
        raise ValueError("dpp_bn_bwd_finalize_apply: dX and the gradient added to it must be stored alike")
    return Launch(rt.lib.dpp_bn_bwd_finalize_apply, (G.ptr, X.ptr, M, Cc, mean.ptr, inv_std.ptr, scale.ptr, partial.ptr, int(nb), _p(add), dX.ptr,
                                                     int(rpb), _p(colsum), dbeta.ptr, dgamma.ptr, _store(a=G, c=dX, bnx=X)),
                  (G, X, mean, inv_std, scale, partial, add, dX, colsum, dbeta, dgamma), name,
                  dict(kernel='bn_bwd_apply', flops=6.0 * M * Cc, bytes=((12.0 if add is not None else 8.0) + _esz(X)) * M * Cc))


def colsum_partial(rt, X, M, Cc, rpb, partial, name='colsum_partial'):
    if _is16(X):
        raise NotImplementedError("dpp_colsum_partial reads float32 gradients")
    return Launch(rt.lib.dpp_colsum_partial, (X.ptr, M, Cc, rpb, partial.ptr), (X, partial), name,
                  dict(kernel='colsum_partial', flops=1.0 * M * Cc, bytes=4.0 * M * Cc))


def loss_sse(rt, out, y, rows, d, denom, cost, dout=None, name='loss_sse'):
    return Launch(rt.lib.dpp_loss_sse, (out.ptr, y.ptr, rows, d, denom, cost.ptr, _p(dout)), (out, y, cost, dout), name)


def reduce_partials_loss(rt, partial, nz, rows, d, out, bias, y, denom, cost, dout=None, name='reduce_partials_loss'):
    """reduce_partials (+ bias) of the net's last HiddenLayer and loss_sse on its output in one launch."""
    n = rows * d
    return Launch(rt.lib.dpp_reduce_partials_loss, (partial.ptr, nz, rows, d, _p(bias), out.ptr, y.ptr, denom, cost.ptr, _p(dout)),
                  (partial, out, bias, y, cost, dout), name, dict(kernel='reduce_partials', flops=float(nz) * n, bytes=4.0 * (nz + 3) * n))


def loss_sse_bcast(rt, out, y, n, cost, dout=None, err=None, name='loss_sse_bcast'):
    return Launch(rt.lib.dpp_loss_sse_bcast, (out.ptr, y.ptr, n, cost.ptr, _p(dout), _p(err)), (out, y, cost, dout, err), name)


def error_l2(rt, out, y, rows, d, err, name='error_l2'):
    return Launch(rt.lib.dpp_error_l2, (out.ptr, y.ptr, rows, d, err.ptr), (out, y, err), name)


def adam(rt, w, g, m, v, n, hyper, name='adam', tick=False):
    """tick: the launch also advances the step count t (dpp_adam_ticked) -- no dpp_adam_tick launch behind it."""
    return Launch(rt.lib.dpp_adam_ticked if tick else rt.lib.dpp_adam, (w.ptr, g.ptr, m.ptr, v.ptr, n, hyper.ptr), (w, g, m, v, hyper), name,
                  dict(kernel='adam', flops=12.0 * n, bytes=28.0 * n))


def _f32_only(what, *bufs):
    if any(_is16(b) for b in bufs):
        raise NotImplementedError("%s reads / writes float32 tensors only (a bf16-stored tensor reached it)" % what)


def axpy(rt, y, x, alpha, n, name='axpy'):
    _f32_only('dpp_axpy', y, x)
    return Launch(rt.lib.dpp_axpy, (y.ptr, x.ptr, float(alpha), n), (y, x), name)


def sumsq(rt, x, n, alpha, out, accumulate, name='sumsq'):
    return Launch(rt.lib.dpp_sumsq, (x.ptr, n, float(alpha), out.ptr, int(accumulate)), (x, out), name)


def segment_table(rt, base, views):
    """(offset, length) pairs of `views` inside the flat buffer `base`, as a device array for sumsq_multi / axpy_multi."""
    seg = []
    for v in views:
        off = (v.ptr - base.ptr) // 4
        assert 0 <= off and off + v.size <= base.size and (v.ptr - base.ptr) % 4 == 0, "view outside the flat buffer"
        seg += [off, v.size]
    return rt.upload(np.asarray(seg, np.int64)), len(seg) // 2


def sumsq_multi(rt, base, seg, nseg, alpha, out, accumulate, name='sumsq_multi'):
    ws = rt.alloc(int(rt.lib.dpp_sumsq_multi_workspace_bytes()) // 8, np.float64, zero=False)
    return Launch(rt.lib.dpp_sumsq_multi, (base.ptr, seg.ptr, int(nseg), float(alpha), ws.ptr, out.ptr, int(accumulate)),
                  (base, seg, ws, out), name, kernels=2)


def axpy_multi(rt, ybase, xbase, seg, nseg, alpha, name='axpy_multi'):
    return Launch(rt.lib.dpp_axpy_multi, (ybase.ptr, xbase.ptr, seg.ptr, int(nseg), float(alpha)), (ybase, xbase, seg), name)


def scale(rt, x, y, n, a=1.0, relu=False, mask=None, name='scale'):
    return Launch(rt.lib.dpp_scale, (x.ptr, _p(mask), float(a), int(relu), y.ptr, n), (x, y, mask), name)


def relu_bwd(rt, dy, pre, g, n, a=1.0, mask=None, name='relu_bwd'):
    _f32_only('dpp_relu_bwd', dy, pre, g)
    return Launch(rt.lib.dpp_relu_bwd, (dy.ptr, pre.ptr, _p(mask), float(a), g.ptr, n), (dy, pre, g, mask), name)


def augment_prepare(rt, img, com3d, cube, Mcrop, gt3d, B, J, dsz, cam, records, out_y, mode=None, off=None, rot=None, sc=None,
                    mode_table=None, n_modes=0, seed=0, counter=0, sigma_com=5., sigma_sc=0.02, rot_range=180.,
                    pca_mean=None, pca_comp=None, E=0, out_mode=None, counter_dev=None, norm_zero_one=False, name='augment_prepare'):
    fx, fy, ux, uy, flip = cam
    return Launch(rt.lib.dpp_augment_prepare,
                  (img.ptr, com3d.ptr, cube.ptr, Mcrop.ptr, gt3d.ptr, B, J, dsz, _p(mode), _p(off), _p(rot), _p(sc), _p(mode_table),
                   n_modes, seed, counter, float(sigma_com), float(sigma_sc), float(rot_range), float(fx), float(fy), float(ux),
                   float(uy), int(flip), int(bool(norm_zero_one)), _p(pca_mean), _p(pca_comp), E, records.ptr, out_y.ptr, _p(out_mode),
                   _p(counter_dev)),
                  (img, com3d, cube, Mcrop, gt3d, mode, off, rot, sc, mode_table, pca_mean, pca_comp, records, out_y, out_mode,
                   counter_dev), name)


def augment_warp(rt, img, records, B, dsz, out, name='augment_warp'):
    return Launch(rt.lib.dpp_augment_warp, (img.ptr, records.ptr, B, dsz, out.ptr), (img, records, out), name,
                  dict(kernel='augment_warp', flops=40.0 * B * dsz * dsz, bytes=8.0 * B * dsz * dsz))


def augment(rt, img, com3d, cube, Mcrop, gt3d, B, J, dsz, cam, out_x, out_y, records=None, mode=None, off=None, rot=None, sc=None,
            mode_table=None, n_modes=0, seed=0, counter=0, sigma_com=5., sigma_sc=0.02, rot_range=180., pca_mean=None, pca_comp=None,
            E=0, out_mode=None, counter_dev=None, ticket=None, sample0=0, global_batch=0, splits=0, norm_zero_one=False, binarize=False,
            name='augment'):
    """prepare + warp as one launch (dpp_augment); with `ticket` it also advances the device draw counter."""
    fx, fy, ux, uy, flip = cam
    return Launch(rt.lib.dpp_augment,
                  (img.ptr, com3d.ptr, cube.ptr, Mcrop.ptr, gt3d.ptr, B, J, dsz, _p(mode), _p(off), _p(rot), _p(sc), _p(mode_table),
                   n_modes, seed, counter, float(sigma_com), float(sigma_sc), float(rot_range), float(fx), float(fy), float(ux),
                   float(uy), int(flip), int(bool(norm_zero_one)) | (2 if binarize else 0), _p(pca_mean), _p(pca_comp), E, _p(records), out_x.ptr, out_y.ptr,
                   _p(out_mode), _p(counter_dev), _p(ticket), int(sample0), int(global_batch), int(splits)),
                  (img, com3d, cube, Mcrop, gt3d, mode, off, rot, sc, mode_table, pca_mean, pca_comp, records, out_x, out_y, out_mode,
                   counter_dev, ticket), name, dict(kernel='augment', flops=40.0 * B * dsz * dsz, bytes=8.0 * B * dsz * dsz))


class AugmentState(object):
    """The device-resident draw counter (+ the ticket word that advances it) of one augmentation stream, shared by the
    launches over all data slices; sample0 / global_batch key the draws by the GLOBAL sample index (data parallelism)."""

    def __init__(self, rt, B, seed=0, sample0=0, global_batch=0):
        self.rt, self.B, self.seed, self.sample0, self.global_batch = rt, int(B), int(seed), int(sample0), int(global_batch or B)
        self.counter = rt.alloc(1, np.int64)
        self.ticket = rt.alloc(1, np.int32)

    def ops(self, img, com3d, cube, Mcrop, gt3d, J, dsz, cam, out_x, out_y, **kw):
        return [augment(self.rt, img, com3d, cube, Mcrop, gt3d, self.B, J, dsz, cam, out_x, out_y, seed=self.seed, counter=0,
                        counter_dev=self.counter, ticket=self.ticket, sample0=self.sample0, global_batch=self.global_batch, **kw)]


def crop_prepare(rt, frames, B, H, W, com, cube, fx, fy, dsz, records, M_out=None, stretch=False, name='crop_prepare'):
    return Launch(rt.lib.dpp_crop_prepare, (frames.ptr, B, H, W, com.ptr, cube.ptr, float(fx), float(fy), dsz, int(bool(stretch)), records.ptr,
                                            _p(M_out)),
                  (frames, com, cube, records, M_out), name, dict(kernel='crop_prepare', flops=2.0 * B * H * W, bytes=4.0 * B * H * W))


def crop_com(rt, frames, records, B, H, W, com_out, name='crop_com'):
    ws = rt.alloc(max(1, int(rt.lib.dpp_crop_com_workspace_bytes(B)) // 8), np.float64, zero=False)     # per-band partial sums
    return Launch(rt.lib.dpp_crop_com, (frames.ptr, records.ptr, B, H, W, ws.ptr, com_out.ptr), (frames, records, ws, com_out), name,
                  kernels=2)


def crop_warp(rt, frames, records, B, H, W, dsz, out, normalize=True, nd_value=0.0, name='crop_warp'):
    return Launch(rt.lib.dpp_crop_warp, (frames.ptr, records.ptr, B, H, W, dsz, int(bool(normalize)), float(nd_value), out.ptr),
                  (frames, records, out), name, dict(kernel='crop_warp', flops=10.0 * B * dsz * dsz, bytes=8.0 * B * dsz * dsz))


def crop_refine(rt, frames, records, B, H, W, com_in, cube, net_out, cam, com_out, gt3d_orig=None, J=0, pca_mean=None, pca_comp=None, E=0,
                com3d_out=None, gt3d_crop=None, out_y=None, name='crop_refine'):
    """dpp_crop_refine: the refined crop centre from a refinement net's output (+ optionally the labels of the re-cropped frame)."""
    fx, fy, ux, uy, flip = cam
    return Launch(rt.lib.dpp_crop_refine,
                  (frames.ptr, records.ptr, B, H, W, com_in.ptr, cube.ptr, net_out.ptr, float(fx), float(fy), float(ux), float(uy), int(flip),
                   _p(gt3d_orig), int(J), _p(pca_mean), _p(pca_comp), int(E), com_out.ptr, _p(com3d_out), _p(gt3d_crop), _p(out_y)),
                  (frames, records, com_in, cube, net_out, gt3d_orig, pca_mean, pca_comp, com_out, com3d_out, gt3d_crop, out_y), name)


def copy2d(rt, src, lds, dst, ldd, rows, cols, relu=False, name='copy2d'):
    return Launch(rt.lib.dpp_copy2d, (src.ptr, lds, dst.ptr, ldd, rows, cols, int(bool(relu))), (src, dst), name)


def rowscale(rt, x, s, scol, factor, out, rows, cols, name='rowscale'):
    """out[r][c] = x[r][c] * (s[r][scol] * factor); s has s.shape[-1] columns."""
    return Launch(rt.lib.dpp_rowscale, (x.ptr, s.ptr, int(s.shape[-1]), int(scol), float(factor), out.ptr, int(rows), int(cols)), (x, s, out), name)


def crop_center(rt, src, B, H, W, dst, h, w, name='crop_center'):
    return Launch(rt.lib.dpp_crop_center, (src.ptr, B, H, W, dst.ptr, h, w), (src, dst), name)


def fill_zero(rt, buf, name='fill_zero'):
    return Launch(rt.lib.dpp_fill_zero, (buf.ptr, buf.nbytes), (buf,), name)


def bernoulli_mask(rt, mask, n, keep, seed, counter, counter_dev=None, name='bernoulli_mask'):
    return Launch(rt.lib.dpp_bernoulli_mask, (mask.ptr, n, float(keep), int(seed), int(counter), _p(counter_dev)), (mask, counter_dev), name)


def adam_tick(rt, state, name='adam_tick'):
    return Launch(rt.lib.dpp_adam_tick, (state.ptr,), (state,), name)


def counter_add(rt, counter, inc=1, name='counter_add'):
    return Launch(rt.lib.dpp_counter_add, (counter.ptr, int(inc)), (counter,), name)


class ReduceJobs(object):
    """Collects (partial, nz, n, out) reductions and emits ONE dpp_reduce_multi launch for all of them."""

    def __init__(self, rt):
        self.rt = rt
        self.jobs = []

    def add(self, partial, nz, n, out):
        self.jobs.append((partial, int(nz), int(n), out))

    def pending_bytes(self):
        return 4 * sum(nz * n for (_, nz, n, _) in self.jobs)

    def flush(self, name='reduce_multi'):
        """The launch for the jobs collected so far; the list starts over (several launches per pass, see engine._emit_backward)."""
        op = self.launch(name)
        self.jobs = []
        return op

    def launch(self, name='reduce_multi'):
        if not self.jobs:
            return None
        import struct
        rt = self.rt
        assert rt.lib.dpp_reduce_job_bytes() == 32
        raw, block0 = b'', 0
        cols = rt.lib.dpp_reduce_multi_block_cols()
        for (partial, nz, n, out) in self.jobs:
            raw += struct.pack('<QQiiii', partial.ptr, out.ptr, nz, n, block0, 0)
            block0 += -(-n // cols)
        table = rt.upload(np.frombuffer(raw, np.uint8).copy())
        flops = float(sum(nz * n for (_, nz, n, _) in self.jobs))
        return Launch(rt.lib.dpp_reduce_multi, (table.ptr, len(self.jobs), block0), (table, list(self.jobs)), name,
                      dict(kernel='reduce_multi', flops=flops, bytes=4.0 * flops))
