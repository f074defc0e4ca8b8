"""
hipdp.backward -- the backward-pass half of hipdp.engine.CompiledNet (split out of hipdp/engine.py in round 6: a mixin, same methods).

What `T.grad(cost, params)` builds for the reference (/root/reference/src/trainer/poseregnettrainer.py:110-111) is emitted here layer by
layer in reverse: data gradients with the BatchNorm-backward mask / sums in their epilogues, bn_bwd_finalize / bn_bwd_apply, filter and bias
gradients as a parallel branch on the second stream, one reduction launch for every partial of the pass.
"""
import numpy as np

from . import ops
from . import heuristics as hz
from .lib import Act, RowMap
from .store import TensorV, View, _layer_kind, _pad4


class BackwardMixin(object):
    def _grad_dtype(self, t, flattened=False):
        """Storage of the gradient of activation tensor t (or of a BatchNorm / ReLU view over it): bf16 in the bf16 mode for the big
        [pixels][channels] maps, float32 where a float32-only kernel touches it -- the stem's output (its gradient feeds stem_wgrad),
        a view the FC head reads (written by dpp_fc_gemm) or one whose gradient is accumulated by two kernels, and maps of at most 256 rows (their gradient doubles as the list of
        bias-gradient partials of dpp_reduce_multi)."""
        if not self.grad16 or not t.is16 or flattened or t.rows <= 256:
            return np.float32
        for io in self.layer_io.values():
            if io.get('out') is t and io.get('stem'):
                return np.float32
        return np.uint16

    def _view_grad(self, view):
        """The buffer holding d(cost)/d(view value); plain views write straight into the base tensor's gradient."""
        if view.plain:
            t = view.base
            if t.grad is None:
                t.grad = self.rt.alloc(t.shape, self._grad_dtype(t), zero=False)
            return t, t.grad
        key = view.key()
        vg = self.view_grads.get(key)
        if vg is None:
            # a view with several consumers (the BatchNorm a projection block feeds to its main path AND its shortcut) collects its
            # gradient in two kernels: the first share would make a bf16 round trip before the second is added -- kept float32
            var = getattr(view, 'var', None)
            shared = var is not None and len(self.consumers.get(id(var), [])) > 1
            vg = TensorV(self.rt.alloc(view.base.shape, self._grad_dtype(view.base, flattened=len(view.shape) == 2 or shared), zero=False),
                         view.base.shape, 'dA')
            vg.grad = vg.buf
            vg.view = view
            self.view_grads[key] = vg
        return vg, vg.grad

    def _grad_of(self, t):
        """Materialised gradient of tensor t (None if nothing flows into it)."""
        rt = self.rt
        if getattr(t, 'lazy', None) is not None:
            self._materialise_lazy(t)
        if t.grad_written:
            for p in t.pending:
                self.bwd.add(ops.axpy(rt, t.grad, p, 1.0, t.grad.size))
            t.pending = []
            return t.grad
        if len(t.pending) == 1:
            g = t.pending[0]
            t.pending = []
            t.grad, t.grad_written = g, True
            return g
        if len(t.pending) > 1:
            t.grad = rt.alloc(t.shape, zero=True)
            t.grad_written = True
            return self._grad_of(t)
        return None

    def _resolve_view(self, vg):
        """Back-propagate an accumulated view gradient through its pending ReLU / BatchNorm into the base tensor."""
        view, rt, st = vg.view, self.rt, self.store
        t = view.base
        if not vg.grad_written:
            return
        n = int(np.prod(t.shape))
        if view.bn is None:
            # ReLU only (hidden-layer activation): g = dA * [pre >= 0], in place
            self.bwd.add(ops.relu_bwd(rt, vg.grad, t.buf, vg.grad, n))
            t.pending.append(vg.grad)
            return
        b = view.bn
        M, C, W = b.M, b.C, b.world
        # (the masked gradient G of this BatchNorm, as its backward kernels read it: kept for the tests' pins)
        self.__dict__.setdefault('bn_view_grad', {})[id(b.layer)] = vg.grad
        c1, c2 = rt.alloc(_pad4(C)), rt.alloc(_pad4(C))
        # If t was produced by a 1x1 convolution and nothing else flows into it, the gradient through the batch statistics
        # dX = scale*(G - c1 - xhat*c2) is never written: that convolution's data- and filter-gradient GEMMs form it from
        # (G, x) while they stage their operand (dpp_act mode 4), which removes a launch from the dependent chain and a pass
        # over the tensor.  The finalize then also writes the two per-channel constants that prologue needs.
        lazy = hz.LAZY_BN_BWD and not t.grad_written and not t.pending and self._produced_by_conv1x1(t)
        if lazy and hz.LAZY_BN_BWD == 3:
            pl = self._producer(t)                       # conv Ci -> C; its data gradient is the GEMM  [M x C] . [C x Ci]
            src_t = self.layer_io[id(pl)]['in_view'].base
            lazy = pl.cfgParams.stride[0] == 1 and src_t not in self.x_ins and hz.expand_plan(M, src_t.C, C, False) is not None
        q, p = (rt.alloc(_pad4(C)), rt.alloc(_pad4(C))) if lazy else (None, None)
        fin = dict(bn=b, q=q, p=p) if lazy else {}
        fused = getattr(vg, 'fused_reduce', None)
        # few blocks of sums: the apply pass reduces them itself (one launch for finalize + apply)
        nbp = fused[1] if fused is not None else b.nb
        one_launch = (not lazy and W <= 1 and 0 < nbp <= hz.BN_BWD_FUSE_MAX_BLOCKS and rt.lib.dpp_bn_bwd_finalize_apply_ok(M, C, nbp))
        if fused is not None:
            # the data-gradient kernel already masked vg.grad and wrote the per-block sums
            part, nbp = fused
            if not one_launch:
                self.bwd.add(ops.bn_bwd_finalize(rt, part, nbp, M, C, st.view(b.layer.beta, 'g'), st.view(b.layer.gamma, 'g'), c1, c2, **fin))
            W = 0
        else:
            part = self.scratch(b.nb * 2 * C * (W + 1)).view(0, (b.nb * 2 * C,))
            self.bwd.add(ops.bn_bwd_reduce(rt, vg.grad, t.buf, M, C, b.mean, b.inv_std, b.scale, b.beta_buf, int(view.relu), vg.grad, b.rpb, part))
        if W > 1:
            allp = self.scratch(b.nb * 2 * C * (W + 1)).view(b.nb * 2 * C, (W * b.nb * 2 * C,))
            self.bwd.add(self.dp.all_gather_op(part, allp, 'bn_bwd_allgather'))
            part = allp
        # with sync-BN dbeta / dgamma are already global sums on every rank: pre-divide so that the gradient all-reduce
        # (a sum over ranks) leaves them unchanged
        if fused is None and not one_launch:
            self.bwd.add(ops.bn_bwd_finalize(rt, part, b.nb, M * W, C, st.view(b.layer.beta, 'g'), st.view(b.layer.gamma, 'g'), c1, c2,
                                             nseg=W, **fin))
        if W > 1:
            for prm in (b.layer.beta, b.layer.gamma):
                gv = st.view(prm, 'g')
                self.bwd.add(ops.scale(rt, gv, gv, C, a=1.0 / W))
        if lazy:
            t.lazy = dict(G=vg.grad, bn=b, c1=c1, c2=c2, q=q, p=p)
            return
        self._emit_bn_bwd_apply(t, vg.grad, b, c1, c2, sums=(part, nbp) if one_launch else None)

    def _producer(self, t):
        for l in self.layers:
            io = self.layer_io.get(id(l))
            if io is not None and io.get('out') is t:
                return l
        return None

    def _produced_by_conv1x1(self, t):
        l = self._producer(t)
        return (l is not None and _layer_kind(l) == 'ConvLayer' and tuple(l.cfgParams.filterDim) == (1, 1) and
                self.layer_io[id(l)].get('residual') is None)

    def _materialise_lazy(self, t):
        """Fallback: write the lazily represented gradient of t after all (a consumer that cannot take the two-tensor operand)."""
        lz = t.lazy
        t.lazy = None
        self._emit_bn_bwd_apply(t, lz['G'], lz['bn'], lz['c1'], lz['c2'])

    def _emit_bn_bwd_apply(self, t, G, b, c1, c2, sums=None):
        """dX = scale * (G - c1 - xhat * c2) (+ the gradient already flowing into t).  sums = (partial, nb): the finalize has not run,
        the pass reduces the per-block sums itself (dpp_bn_bwd_finalize_apply) and writes dbeta / dgamma."""
        rt = self.rt
        M, C = b.M, b.C
        addends = ([t.grad] if t.grad_written else []) + t.pending
        t.pending = []
        add = None
        if addends:
            add = addends[0]
            for extra in addends[1:]:
                raise NotImplementedError("more than one extra gradient path into a BatchNorm input")
        if t.grad is None:
            # (the gradient added to dX -- the identity path of a residual sum -- and dX are stored alike)
            t.grad = rt.alloc(t.shape, add.dtype if add is not None else self._grad_dtype(t), zero=False)
        # t.grad is the dY of the conv(s) that produced t: emit its column sums (their bias gradients) in the same pass
        if sums is not None:
            rpb = max(32, -(-(M * (C // 32)) // hz.BN_BWD_FUSE_TARGET_WGS))
            rpb = -(-rpb // 32) * 32
            nbc = -(-M // rpb)
            cs = rt.alloc((nbc, C), zero=False)
            st = self.store
            self.bwd.add(ops.bn_bwd_finalize_apply(rt, G, t.buf, M, C, b.mean, b.inv_std, b.scale, sums[0], sums[1], t.grad,
                                                   st.view(b.layer.beta, 'g'), st.view(b.layer.gamma, 'g'), add=add, rpb=rpb, colsum=cs))
            self.colsum_of[t.grad.ptr] = (cs, nbc, C)
            t.grad_written = True
            return
        cs = rt.alloc((b.nb, C), zero=False)
        self.bwd.add(ops.bn_bwd_apply(rt, G, t.buf, M, C, b.mean, b.inv_std, b.scale, c1, c2, t.grad, add=add, rpb=b.rpb, colsum=cs))
        self.colsum_of[t.grad.ptr] = (cs, b.nb, C)
        t.grad_written = True

    def _emit_backward(self):
        rt, st = self.rt, self.store
        # views are resolved at the layer that created them, walking the layer list backwards
        created_by = {}
        for var in self.order:
            if var.kind == 'layer' and _layer_kind(var.layer) in ('NonlinearityLayer', 'BatchNormLayer'):
                created_by[id(var.layer)] = self._memo[id(var)]
            if var.kind == 'relu':
                created_by[('relu', id(var.inputs[0].layer))] = self._memo[id(var)]
        order = list(reversed(self.layers))
        for pl_id, ql in self._bwd_after.items():              # a shortcut whose gradient arrives through the LATER-processed exit conv
            pl = [l for l in order if id(l) == pl_id][0]
            order.remove(pl)
            order.insert(order.index(ql) + 1, pl)
        for layer in order:
            kind = _layer_kind(layer)
            if kind in ('NonlinearityLayer', 'BatchNormLayer'):
                v = created_by.get(id(layer))
                if v is not None and v.key() in self.view_grads:
                    vg = self.view_grads.pop(v.key())
                    self._resolve_view(vg)
                continue
            io = self.layer_io[id(layer)]
            v = created_by.get(('relu', id(layer)))          # activation wrapped around this layer's output
            if v is not None and v.key() in self.view_grads:
                self._resolve_view(self.view_grads.pop(v.key()))
            out = io['out']
            lz = getattr(out, 'lazy', None)
            if lz is not None and kind == 'ConvLayer' and tuple(layer.cfgParams.filterDim) == (1, 1) and \
                    (hz.LAZY_BN_BWD == 1 or io['in_view'].base not in self.x_ins):
                out.lazy = None
                b = lz['bn']
                keep = None
                if hz.LAZY_BN_BWD >= 2:
                    if out.grad is None:
                        out.grad = rt.alloc(out.shape, zero=False)
                    keep = out.grad
                self._bwd_conv(layer, io, io['in_view'], lz['G'], dY_act=ops.act_bn_bwd(b, lz['q'], lz['p'], out.buf, b.C, out=keep),
                               dY_keep=keep)
                continue
            dY = self._grad_of(out)
            if dY is None:
                continue
            src = io['in_view']
            if kind == 'ConvLayer':
                self._bwd_conv(layer, io, src, dY)
            elif kind == 'HiddenLayer':
                self._bwd_fc(layer, io, src, dY)
            elif kind == 'ConvPoolLayer':
                if hz.TAIL_REDUCE and self.reduce_jobs.jobs:
                    # the partials collected so far are reduced on the gradient branch BESIDE the stem's filter gradient (the last
                    # launch of the main stream, which otherwise idles at the join while reduce_multi waits behind it)
                    self.bwd.fork()
                    self.bwd.add(self.reduce_jobs.flush('reduce_multi_side'), side=True)
                self._bwd_stem(layer, io, dY)
            elif kind == 'DropoutLayer':
                tgt, dst = self._view_grad(src)
                if tgt.grad_written:
                    raise NotImplementedError("dropout input with several consumers")
                n = int(np.prod(src.shape))
                self.bwd.add(ops.scale(rt, dY, dst, n, relu=False, mask=io['mask']))
                tgt.grad_written = True
            else:
                raise NotImplementedError(kind)
            if hz.EARLY_REDUCE_BYTES > 0 and self.reduce_jobs.pending_bytes() >= hz.EARLY_REDUCE_BYTES:
                self.bwd.fork()
                self.bwd.add(self.reduce_jobs.flush('reduce_multi_early'), side=True)
        self._defer_fc1_wgrad()
        self.bwd.join()
        self.bwd.add(self.reduce_jobs.flush())        # the remaining filter / bias gradient partials of the pass, one launch
        for slot, priv in self.__dict__.get('_shared_grad_adds', []):
            self.bwd.add(ops.axpy(rt, slot, priv, 1.0, slot.size))     # shared parameters: sum of the per-use gradients
        # cost += wd * sum(W^2): gradient 2*wd*W.  Single process: the last step of the backward plan.  Data parallel: the
        # regulariser is NOT a per-shard partial sum, so it is added once, AFTER the gradient all-reduce (added before, the
        # sum over ranks would scale it by the world size -- and the axpy would write the slice whose all-reduce the early bucket
        # still has in flight)
        self._wd_ops, self._wd_of = [], {}
        if self.weight_decay and not self.net.hasDropout():
            seg = self._weight_segments()
            if seg is not None and not hz.EARLY_ADAM:       # (hz.EARLY_ADAM moves FC1's share next to FC1's update: per-layer launches)
                self._wd_ops.append(ops.axpy_multi(rt, st.g, st.w, seg[0], seg[1], 2.0 * self.weight_decay))
            else:
                for W in self._unique_weights():
                    self._wd_ops.append(ops.axpy(rt, st.view(W, 'g'), st.view(W), 2.0 * self.weight_decay, int(np.prod(W.shape))))
                    self._wd_of[id(self._wd_ops[-1])] = W.auto_name
        if self.dp is None:
            for o in self._wd_ops:
                self.bwd.add(o)

    def _defer_fc1_wgrad(self):
        """FC1's filter gradient (4.3 GFLOP, f32-MFMA-bound like FC1's data gradient) is the first big launch of the gradient branch
        and runs BESIDE FC1's data gradient: the two share the matrix cores and the main chain waits longer for its first link.  Nothing
        needs this gradient before the end of the pass, and a few launches later the chain is in the latency-bound stage-4 / 3
        convolutions that leave the matrix cores idle -- so the launch moves hz.FC1_WGRAD_DEFER side launches down the branch (it then
        sits behind a later fork, i.e. waits for more of the chain than it needs).  Not under data parallelism (its all-reduce
        bucket wants the gradient early) or hz.EARLY_ADAM (its update sits right behind it).
        Measured on the MI355X (tools/knob_sweep.sh, 300 steps each): 3.659 / 3.675 / 3.676 / 3.678 / 3.649 / 3.675 ms for 2 / 4 / 8 /
        16 / 30 / 60 launches against 3.680-3.682 without -- inside the run-to-run noise, so it stays off (DPP_FC1_WGRAD_DEFER = 0)."""
        op = self.__dict__.get('_fc1_wgrad_op')
        if op is None or hz.FC1_WGRAD_DEFER <= 0 or self.dp is not None or hz.EARLY_ADAM:
            return
        ops_ = self.bwd.ops
        i = [k for k, (o, _) in enumerate(ops_) if o is op][0]
        entry = ops_.pop(i)
        seen, j = 0, i
        while j < len(ops_) and seen < hz.FC1_WGRAD_DEFER:
            if ops_[j][1] and isinstance(ops_[j][0], ops.Launch):
                seen += 1
            j += 1
        ops_.insert(j, entry)

    def _sole_consumer_bn_view(self, view):
        """A BatchNorm(+ReLU) view read by exactly one conv: that conv's data-gradient epilogue may finish the BatchNorm
        backward reduction itself (no other contribution will be accumulated into the view's gradient)."""
        if view.bn is None or not self.fuse_bn:
            return False
        var = getattr(view, 'var', None)
        return var is not None and self._single_consumer(var) and len(view.shape) == 4

    def _two_conv1x1_consumers(self, view):
        """A BatchNorm(+ReLU) view read by exactly two 1x1 convs with the same stride (the projection blocks)."""
        if view.bn is None or not self.fuse_bn or len(view.shape) != 4:
            return False
        var = getattr(view, 'var', None)
        cons = self.consumers.get(id(var), []) if var is not None else []
        if len(cons) != 2:
            return False
        for c in cons:
            if c.kind != 'layer' or _layer_kind(c.layer) != 'ConvLayer' or tuple(c.layer.cfgParams.filterDim) != (1, 1):
                return False
        return tuple(cons[0].layer.cfgParams.stride) == tuple(cons[1].layer.cfgParams.stride)

    def _bias_grad(self, dY, rows, C, gslot):
        if dY.ptr in self.colsum_of:
            cs, nb, cc = self.colsum_of[dY.ptr]
            assert cc == C
            self.reduce_jobs.add(cs, nb, C, gslot)
            return
        if rows <= 256:
            # few rows (the FC layers: one row per sample): dY itself is the list of partials of the fused reduction launch
            self.reduce_jobs.add(dY, rows, C, gslot)
            return
        rpb = max(32, -(-rows // 256))
        nb = -(-rows // rpb)
        part = self.rt.alloc(nb * C, zero=False)     # private: groups of the gradient branch may run on different side streams
        self.bwd.add(ops.colsum_partial(self.rt, dY, rows, C, rpb, part), side=True)
        self.bwd.add(ops.reduce_partials(self.rt, part, nb, C, gslot), side=True)

    def _bwd_conv(self, layer, io, src, dY, dY_act=None, dY_keep=None):
        """dY_act: dY is the masked BatchNorm gradient G and the true dY is formed by this operand prologue (see _resolve_view);
        dY_keep: the data gradient leaves the dY it forms there, and the filter gradient reads that plain tensor."""
        rt, st = self.rt, self.store
        c = layer.cfgParams
        N, Hi, Wi, Ci = src.base.shape
        _, Co, Ho, Wo = c.outputDim
        k, s = tuple(c.filterDim), c.stride[0]
        M = N * Ho * Wo
        act = self._act(src)
        gW, gb = self._grad_view(layer.W), self._grad_view(layer.b)
        if io.get('residual') is not None:
            io['residual'].pending.append(dY)                 # identity path of the fused residual add
        need_dx = src.base not in self.x_ins

        def emit_param_grads(dy, dy_act):
            # parameter gradients only READ dY / the forward activations, so they run as a parallel branch on the side
            # stream while the main stream continues with the data-gradient chain
            self.bwd.fork()
            if dY_act is None:
                self._bias_grad(dy, M, Co, gb)
            else:
                # sum over pixels of scale*(G - c1 - xhat*c2) with c1 = mean(G), sum(xhat) = 0: the bias of a convolution that
                # feeds a BatchNorm has no gradient (the reference adds up rounding noise); a job without slices writes zeros
                assert k == (1, 1) and io.get('residual') is None
                self.reduce_jobs.add(dy, 0, Co, gb)
            if k != (1, 1):
                return
            mp = RowMap.strided(s, Ho, Wo, Hi, Wi) if s != 1 else None
            # filter gradient dW[o][c] = sum_m dY[m][o] * act(X)[map(m)][c]
            rpw = hz.wgrad_stream_rows(M)
            if dy_act is None and hz.WGRAD_STREAM and rpw > 0 and rt.lib.dpp_wgrad_stream_slices(Co, Ci, M, rpw) > 0:
                # the row-streaming kernel (csrc/wgrad.hip): operands straight from memory into MFMA fragments, one partial slice per
                # (workgroup, row split), all of them summed by the pass's single reduction launch
                nsl = rt.lib.dpp_wgrad_stream_slices(Co, Ci, M, rpw)
                part = rt.alloc(nsl * Co * Ci, zero=False)
                p16 = int(bool(self._gemm_prec(0, M, role='wgrad') and rt.lib.dpp_wgrad_stream_bf16_ok(Co, Ci)))       # bf16 mode: bf16 MFMA operands
                self.bwd.add(ops.wgrad_stream(rt, dy, Co, src.base.buf, Ci, M, rpw, part, mapX=mp, actX=act,
                                              name='wgrad1x1_%d' % layer.layerNum, precision=p16), side=True)
                self.reduce_jobs.add(part, nsl, Co * Ci, gW)
                return
            tile, splitk = hz.wgrad_plan(Co, Ci, M)
            part = rt.alloc(splitk * Co * Ci, zero=False) if splitk > 1 else None     # persistent: reduced at the end of backward
            self.bwd.add(ops.gemm(rt, dy, src.base.buf, None if splitk > 1 else gW, Co, Ci, M, 0, 0, Co, Ci, Ci, mapB=mp, actA=dy_act,
                                  actB=act, splitk=splitk, partial=part, tile=tile, name='wgrad1x1_%d' % layer.layerNum,
                                  precision=self._gemm_prec(0, M, lazy=dy_act is not None, role='wgrad')), side=True)
            if splitk > 1:
                self.reduce_jobs.add(part, splitk, Co * Ci, gW)

        if dY_keep is None:
            emit_param_grads(dY, dY_act)
        if k == (1, 1):
            mp = RowMap.strided(s, Ho, Wo, Hi, Wi) if s != 1 else None
            if need_dx:
                tgt, dst = self._view_grad(src)
                acc = tgt.grad_written
                if s != 1 and not acc:
                    # the strided data gradient leaves the skipped pixels untouched: they are zeroed on the side stream while
                    # the forward pass runs (the buffer is only written in the backward pass), not in the data-gradient chain
                    self.early_side.append(ops.fill_zero(rt, dst))
                tile, _ = hz.gemm_plan(M, Ci, Co, allow_split=False)
                rs = hz.rowstream_plan(M, Ci, Co, False)
                if rs is not None:
                    tile = rs
                ks = hz.ksplit_plan(M, Ci, Co) if (s == 1 and rs is None and dY_act is None) else None
                if ks is not None:
                    tile = ks
                ex = hz.expand_plan(M, Ci, Co, False) if (s == 1 and rs is None and ks is None) else None
                s16 = hz.stream16_plan(M, Ci, Co, False) if (s == 1 and rs is None and ks is None and dY_act is None and ex is None) else None
                if s16 is not None:
                    tile = s16
                if ex is not None:
                    tile = ex
                variant = 1 if (rs is not None and dY_act is None) else (2 if ks is not None else (4 if ex is not None else (3 if s16 is not None else 0)))

                def build(tile, variant, epi):
                    return ops.gemm(rt, dY, st.view(layer.W), dst, M, Ci, Co, 1, 0, Co, Ci, Ci, mapC=mp, actA=dY_act,
                                    residual=dst if acc else None, tile=tile, epi=epi, variant=variant,
                                    name='dgrad1x1_%d' % layer.layerNum,
                                    precision=self._gemm_prec(variant, Co, lazy=dY_act is not None, role='dgrad'))
                if variant in (2, 3, 4) and ops.gemm_variant_rows(rt, build(tile, variant, None)) != tile[0]:
                    variant, (tile, _) = 0, hz.gemm_plan(M, Ci, Co, allow_split=False)       # see the forward twin
                epi = None
                if (s == 1 and not acc and self._sole_consumer_bn_view(src)) or (acc and self._two_conv1x1_consumers(src)):
                    # ReLU mask + (sum G, sum G*xhat) of the BatchNorm backward in this kernel's epilogue.  A projection block
                    # feeds its BatchNorm output to two 1x1 convs (main path and shortcut, resnet.py:380-414): the first data
                    # gradient writes its share unmasked, the second accumulates onto it (residual = dst) and finishes the
                    # reduction on the sum; pixels the stride-2 row map skips hold zeros and contribute nothing.
                    nb2 = -(-M // tile[0])
                    tgt.fused_reduce = (rt.alloc((nb2, 2, Ci), zero=False), nb2)
                    epi = ops.epilogue(bn=src.bn, bn_x=src.base.buf, bn_relu=src.relu, bn_partial=tgt.fused_reduce[0])
                self.bwd.add(build(tile, variant, epi))
                tgt.grad_written = True
            if dY_keep is not None:
                assert need_dx
                emit_param_grads(dY_keep, None)               # after the data gradient, which wrote dY_keep
        else:
            bm = 64
            # bf16 mode: both operands on the bf16 matrix pipe where the transposed-image kernel takes the layer (16 / 32 / 64 channels, maps
            # at least 12 wide: at 256 x 256 input that includes the 64-channel layers of stages 3-4, which otherwise go to the row stream)
            p16 = int(bool(self.prec and hz.BF16_WGRAD3 and Ci == Co and rt.lib.dpp_conv3x3_wgrad_bf16_ok(N, Hi, Wi, Ci, Co)))
            rpw = hz.wgrad3_stream_rows(N * Hi * Wi, Ci) if (Ci == Co and not p16) else 0
            nblk = rt.lib.dpp_wgrad3_stream_slices(Co, Ci, N, Hi, Wi, rpw) if rpw > 0 else 0
            if nblk > 0:
                part = rt.alloc(nblk * Co * 9 * Ci, zero=False)
                self.bwd.add(ops.wgrad3_stream(rt, dY, Co, src.base.buf, Ci, N, Hi, Wi, rpw, part, actX=act), side=True)
            else:
                nblk = rt.lib.dpp_conv3x3_wgrad_blocks(N, Hi, Wi, Ci, Co, bm)
                part = rt.alloc(nblk * Co * 9 * Ci, zero=False)
                self.bwd.add(ops.conv3x3_wgrad(rt, src.base.buf, N, Hi, Wi, Ci, dY, Co, part, actX=act, bm=bm, precision=p16,
                                               name='wgrad3x3_%d' % layer.layerNum), side=True)
            self.reduce_jobs.add(part, nblk, Co * 9 * Ci, gW)
            if need_dx:
                tgt, dst = self._view_grad(src)
                acc = tgt.grad_written
                Wd = rt.alloc(Co * 9 * Ci, zero=False)
                # the mirrored weights only depend on the parameters: they are prepared on the side stream while the forward
                # pass runs (the side stream is idle then) instead of sitting in the data-gradient chain
                self._wtrans_jobs.append((st.view(layer.W), Co, Ci, Wd))       # one batched launch, see _emit_backward
                bmd = hz.conv3x3_bm(N * Hi * Wi, Ci, hw=(Hi, Wi), prec=self.prec)
                epi = None
                fuse = not acc and self._sole_consumer_bn_view(src)
                if not acc and self._conv3_stream(N, Hi, Wi, Ci, Co, dY, dst, src.base.buf if fuse else None):
                    if fuse:
                        nb2 = N * Hi * Wi // rt.lib.dpp_conv3x3_stream_rows(N, Hi, Wi, Ci)
                        tgt.fused_reduce = (rt.alloc((nb2, 2, Ci), zero=False), nb2)
                        epi = ops.epilogue(bn=src.bn, bn_x=src.base.buf, bn_relu=src.relu, bn_partial=tgt.fused_reduce[0])
                    self.bwd.add(ops.conv3x3_stream(rt, dY, N, Hi, Wi, Ci, Wd, dst, epi=epi, name='dgrad3x3_%d' % layer.layerNum))
                    tgt.grad_written = True
                    return
                if fuse:
                    nb2 = rt.lib.dpp_conv3x3_tiling(N, Hi, Wi, bmd, None, None, None)
                    tgt.fused_reduce = (rt.alloc((nb2, 2, Ci), zero=False), nb2)
                    epi = ops.epilogue(bn=src.bn, bn_x=src.base.buf, bn_relu=src.relu, bn_partial=tgt.fused_reduce[0])
                self.bwd.add(ops.conv3x3(rt, dY, N, Hi, Wi, Co, Wd, Ci, dst, residual=dst if acc else None, bm=bmd, epi=epi,
                                         name='dgrad3x3_%d' % layer.layerNum, precision=self.prec))
                tgt.grad_written = True

    def _bwd_fc(self, layer, io, src, dY):
        rt, st = self.rt, self.store
        Nb, K = src.shape
        Nout = layer.cfgParams.outputDim[1]
        gW, gb = self._grad_view(layer.W), self._grad_view(layer.b)
        act = self._act(src)
        self.bwd.fork()
        self._bias_grad(dY, Nb, Nout, gb)
        stream_kernel = self._fc1_stream(Nb, K, Nout)
        if stream_kernel and self.prec == 0 and hz.FC1_WGRAD_STREAM and (act is None or act.mode < 4) and rt.lib.dpp_fc_wgrad_stream_ok(Nb, K, Nout):
            # the reduction is only the batch: every 64 x 64 block of dW is owned by one wave of the row stream (csrc/wgrad.hip),
            # no LDS pipeline to fill and drain for four chunks, no partials
            op = self.bwd.add(ops.fc_wgrad_stream(rt, src.base.buf, dY, gW, Nb, K, Nout, actX=act, name='fc_wgrad_%d' % layer.layerNum), side=True)
            if K * Nout >= hz.EARLY_BUCKET_MIN:
                self._fc1_wgrad_op = op
        elif stream_kernel:
            op = self.bwd.add(ops.fc_gemm(rt, src.base.buf, dY, gW, K, Nout, Nb, 0, 0, K, Nout, Nout, actA=act, precision=self.prec,
                                          kchunk=hz.FC1_KCHUNK, name='fc_wgrad_%d' % layer.layerNum), side=True)
            if K * Nout >= hz.EARLY_BUCKET_MIN:
                self._fc1_wgrad_op = op
        else:
            tile, _ = hz.gemm_plan(K, Nout, Nb, allow_split=False)
            if K >= 4096 and Nout >= 64 and hz.knob('DPP_FC1_TILE128', '1') != '0':
                tile = (128, 64, 4)          # FC1: 67 MB of output, MFMA-bound (tools/gemm_micro.py fc: 107 -> 84 us)
            self.bwd.add(ops.gemm(rt, src.base.buf, dY, gW, K, Nout, Nb, 0, 0, K, Nout, Nout, actA=act, tile=tile,
                                  name='fc_wgrad_%d' % layer.layerNum), side=True)
        if self.dp is not None and self._early_slice is None and K * Nout >= hz.EARLY_BUCKET_MIN and hz.OVERLAP_ALLREDUCE:
            # data parallel: this gradient (FC1: 90 % of all parameter bytes) is final now -- start its all-reduce from the
            # side stream so that it overlaps the rest of the backward pass
            off = (gW.ptr - st.g.ptr) // 4
            self._early_slice = (off, off + K * Nout)
            self.bwd.add(self.dp.allreduce_sum_async_op(gW, self._early_work, 'grad_allreduce_early'), side=True)
        if src.base not in self.x_ins:
            tgt, dst = self._view_grad(src)
            acc = tgt.grad_written
            tile, splitk = hz.gemm_plan(Nb, K, Nout, allow_split=not acc)
            if stream_kernel:
                self.bwd.add(ops.fc_gemm(rt, dY, st.view(layer.W), dst, Nb, K, Nout, 1, 1, Nout, Nout, K, residual=dst if acc else None,
                                         precision=self.prec, kchunk=hz.FC1_KCHUNK, name='fc_dgrad_%d' % layer.layerNum))
            elif splitk > 1:
                part = self.scratch(splitk * Nb * K)
                self.bwd.add(ops.gemm(rt, dY, st.view(layer.W), None, Nb, K, Nout, 1, 1, Nout, Nout, K, splitk=splitk, partial=part,
                                      tile=tile, name='fc_dgrad_%d' % layer.layerNum))
                self.bwd.add(ops.reduce_partials(rt, part, splitk, Nb * K, dst))
            else:
                self.bwd.add(ops.gemm(rt, dY, st.view(layer.W), dst, Nb, K, Nout, 1, 1, Nout, Nout, K, residual=dst if acc else None,
                                      tile=tile, name='fc_dgrad_%d' % layer.layerNum))
            tgt.grad_written = True
            if getattr(src.base, 'concat_parts', None) is not None:
                self._split_concat_grad(src.base)
        if self._early_adam is None and K * Nout >= hz.EARLY_BUCKET_MIN and gW.ptr >= st.g.ptr and \
                gW.ptr + 4 * K * Nout <= st.g.ptr + 4 * st.n_w:
            # from here on nothing reads this weight or writes its gradient any more (filter gradient on the branch, data gradient
            # on the main stream, both issued above): the position where step_plan may put its ADAM update
            off = (gW.ptr - st.g.ptr) // 4
            self._early_adam = dict(after=self.bwd.ops[-1][0], lo=off, hi=off + K * Nout, W=layer.W)

    def _bwd_convpool(self, layer, io, dY):
        rt, st = self.rt, self.store
        g, src = io['geom'], io['in_view']
        nW = g['Co'] * g['kh'] * g['kw'] * g['Ci']
        self.bwd.fork()
        self._bias_grad(dY, io['out'].rows, g['Co'], self._grad_view(layer.b))
        nblk = rt.lib.dpp_convpool_wgrad_blocks(g['N'], g['Hp'], g['Wp'])
        part = rt.alloc(nblk * nW, zero=False)
        self.bwd.add(ops.convpool_wgrad(rt, src.base.buf, g['N'], g['H'], g['W'], g['Ci'], dY, io['ties'], g['kh'], g['kw'], g['pad'],
                                        g['Co'], g['pool'], part, actX=self._act(src), name='convpool_wgrad_%d' % layer.layerNum), side=True)
        self.reduce_jobs.add(part, nblk, nW, self._grad_view(layer.W))
        if src.base not in self.x_ins:
            tgt, dst = self._view_grad(src)
            if tgt.grad_written:
                raise NotImplementedError("ConvPoolLayer input with several consumers")
            self.bwd.add(ops.convpool_dgrad(rt, dY, io['ties'], g['N'], g['H'], g['W'], g['Ci'], st.view(layer.W), g['kh'], g['kw'],
                                            g['pad'], g['Co'], g['pool'], dst, name='convpool_dgrad_%d' % layer.layerNum))
            tgt.grad_written = True

    def _bwd_stem(self, layer, io, dY):
        if not io.get('stem', True):
            return self._bwd_convpool(layer, io, dY)
        rt, st = self.rt, self.store
        N, H, W, _ = self.in_shape
        Co = layer.cfgParams.nFilters
        out = io['out']
        self.bwd.fork()
        self._bias_grad(dY, out.rows, Co, self._grad_view(layer.b))
        tpb = 8
        nblk = rt.lib.dpp_stem_wgrad_blocks(N, H, W, tpb)
        part = rt.alloc(nblk * Co * 25, zero=False)
        # The first layer's filter gradient needs the very last data gradient, i.e. it cannot start before the main chain is
        # done -- and then the main stream has nothing left to do while the gradient branch still works off its backlog
        # (tools/tail_probe.py): it runs on the main stream.
        self.bwd.add(ops.stem_wgrad(rt, self.x_in.buf, N, H, W, dY, io['argmax'], Co, part, tpb),
                     side=hz.knob('DPP_STEM_WGRAD_SIDE', '0') == '1')
        self.reduce_jobs.add(part, nblk, Co * 25, self._grad_view(layer.W))

    # ------------------------------------------------------------------------------------------ execution
