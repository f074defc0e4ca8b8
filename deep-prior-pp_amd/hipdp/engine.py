"""
hipdp.engine -- compiles a net (the graph reachable from `net.output`) into HIP kernel launch plans.

What the reference gets from `theano.function` (one fused device function per call:
/root/reference/src/net/netbase.py:257-282 for inference, /root/reference/src/trainer/poseregnettrainer.py:146-170
for train_model) is produced here as three replayable plans:
    forward   stem -> [BN stats -> finalize -> conv with BN+ReLU prologue (+bias, +residual)]* -> FC head
    backward  loss gradient -> per layer: data gradient, BN backward (mask, sums, apply), filter/bias gradients
    update    one fused ADAM launch over the flat parameter buffer
BatchNorm / ReLU layers are never materialised: they become "views" (base tensor + pending BN + pending ReLU)
that the consuming conv / FC applies while staging its operand.  A residual `a + conv(...)` is fused into the
conv's epilogue.  Activations are NHWC; parameters live in ONE flat device buffer in kernel layout (so ADAM and a
data-parallel gradient all-reduce are single operations), with get_value()/set_value() converting to the
reference's layouts on the fly.

Round 6: the tensors / views / parameter store live in hipdp/store.py, the backward-pass emission in hipdp/backward.py (BackwardMixin);
this module keeps the graph walk, the forward emission, the loss / update plans and the run-time API of CompiledNet.
"""
import os
from ctypes import c_int as C_int

import numpy as np

from . import evalfuse, layout, ops
from . import heuristics as hz
from .lib import Act, RowMap
from .ops import Plan
from .runtime import default_runtime
from .backward import BackwardMixin
# (re-exported: tests and tools import these names from hipdp.engine)
from .store import BNState, ParamStore, TensorV, View, _collect, _dedupe_specs, _layer_kind, _pad4, _param_specs, bf16_bits_to_f32, get_store      # noqa: F401


class CompiledNet(BackwardMixin):
    def __init__(self, net, train=False, runtime=None, loss=None, weight_decay=0.0, dp=None, fuse_bn=True, bf16=None, optimizer=None,
                 fuse_blocks=None):
        """
        :param fuse_blocks: deterministic engines only: lower every recognised bottleneck block to one dpp_resblock_eval launch
                      (default: on; False keeps one tensor per layer, which the per-layer tests read)
        :param bf16:  True -> the 3x3 convolutions (forward, data gradient) and the HiddenLayer behind the last conv map (FC1:
                      forward, data and weight gradient) round their operands to bf16 and accumulate in f32 on the bf16 matrix
                      pipe (BASELINE config 5).  Opt-in (default: DPP_BF16=1 in the environment): the f32 path is the one that
                      meets the 1e-3 mm bar.
        :param optimizer: None (the reference's ADAM defaults) or trainer.optimizer.Optimizer(...).rule: dict(name='ADAM', beta1=, beta2=,
                      epsilon=, gamma=) / dict(name='RMSProp', decay=, epsilon=) -- the constants of the fused update kernel
        :param train: True -> BatchNorm uses batch statistics (and updates the running ones), dropout uses masks,
                      and the loss / backward / ADAM plans are built; False -> deterministic forward only.
        :param loss:  None, or dict(kind='embedding'|'joints', numJoints=, nDims=) -- the cost of
                      poseregnettrainer.py:92-99; also available in eval mode (validation cost / error).
        """
        self.rt = rt = runtime or default_runtime()
        self.optimizer = dict(optimizer) if optimizer else dict(name='ADAM')
        self.prec = int(bool(hz.BF16_DEFAULT if bf16 is None else bf16))
        self.store16 = self.grad16 = False                      # set below, once the layer list is known
        self.dp = dp                                  # hipdp.parallel.DataParallel or None
        # BatchNorm statistics / backward sums produced by the conv epilogues (off with sync-BN, whose partials are
        # all-gathered by row block, and for tests that want the stand-alone BatchNorm kernels)
        self.fuse_bn = bool(fuse_bn) and not (dp is not None and dp.sync_bn)
        self.net, self.train = net, train
        self.N = net.cfgParams.batch_size
        self.key = None
        self.weight_decay = float(weight_decay)
        self.order, self.consumers, self.layers = _collect(net)
        # bf16 storage is built for the ResNet path (stem + ConvLayer + FC head); a net with generic ConvPoolLayers / concatenations
        # (PoseRegNet, ScaleNet -- e.g. the refinement net of the config-5 cascade) keeps float32 tensors in the bf16 mode
        resnet_like = all(_layer_kind(l) != 'ConvPoolLayer' or (tuple(l.cfgParams.filterDim) == (5, 5) and tuple(l.cfgParams.poolsize) == (2, 2)
                                                                 and l.cfgParams.nFilters <= 32 and l.cfgParams.activation is None)
                          for l in self.layers) and not any(v.kind == 'concat' for v in self.order)
        self.store16 = bool(self.prec and hz.BF16_STORE and resnet_like)         # conv outputs held as bf16 (see hz.BF16_STORE)
        self.grad16 = bool(self.store16 and hz.BF16_GRADS)                        # ... and their gradients (see hz.BF16_GRADS)
        self.store = get_store(net, rt, self.layers)
        self.fwd, self.bwd, self.upd = Plan('forward'), Plan('backward'), Plan('update')
        self.views = {}
        self.tensors = []
        self.bn_states = {}
        self.layer_io = {}          # id(layer) -> dict(in_view, out_tensor, extra)
        self.view_grads = {}        # view key -> dict(view, dA, written)
        self._scratch = None
        self._bn_scratch = None
        self.colsum_of = {}                 # dY buffer ptr -> (column-sum partials, nb, C) emitted by bn_bwd_apply
        self.reduce_jobs = ops.ReduceJobs(rt)
        self.dropout_masks = {}
        self.step_ctr = rt.alloc(1, np.int64) if train else None     # device-resident step counter (dropout mask streams)
        in_dims = net.cfgParams.inputDim
        in_vars = net.inputVar
        if not isinstance(in_vars, (list, tuple)):
            in_dims, in_vars = [in_dims], [in_vars]
        self.x_ins, self.input_of = [], {}
        for k, (d, v) in enumerate(zip(in_dims, in_vars)):                 # ScaleNet has three inputs (scalenet.py:150-156)
            shp = (d[0], d[2], d[3], d[1])
            t = TensorV(rt.alloc(shp), shp, 'x%d' % k)
            self.x_ins.append(t)
            self.input_of[id(v)] = t
        self.x_in, self.in_shape = self.x_ins[0], self.x_ins[0].shape
        self._memo = {}
        self.fuse_blocks = (hz.EVAL_FUSE if fuse_blocks is None else bool(fuse_blocks)) and not train
        self.fused_blocks = []         # the blocks that became one launch (evalfuse.match_block dicts)
        self._bn_eval_jobs = []        # deterministic mode: (gamma, run_mean, run_inv_std, C, mean, inv_std, scale) of every BatchNorm
        self._beside = []           # (fork / join markers, launches) of the shortcuts moved beside the chain (see _emit_add)
        self._bwd_after = {}        # id(layer p) -> layer q: p's backward is emitted right after q's (see _emit_add)
        out_view = self._emit(net.output)
        if self._bn_eval_jobs:
            # one launch at the head of the pass instead of one per BatchNorm (61 of the 131 launches of the 128x128 ResNet's forward)
            self.fwd.ops[0:0] = [(ops.bn_eval_coeffs_multi(rt, self._bn_eval_jobs), False)]
        self.out = self._materialize_plain(out_view)
        self.out_dim = int(np.prod(self.out.shape[1:]))
        self.loss_cfg = loss
        if loss is not None:
            self.y_in = rt.alloc((self.N, self.out_dim))
            self.cost = rt.alloc(1)
            self.err = rt.alloc(2)
            self.lossplan = Plan('loss')
            gN = self.N * (dp.world if (dp is not None and train) else 1)      # cost normalised by the GLOBAL batch
            denom = gN if loss.get('kind', 'embedding') in ('embedding', 'scalar') else gN * loss['numJoints']
            if train:
                self.out.grad = rt.alloc(self.out.shape)
                self.out.grad_written = True
            if loss.get('kind') == 'scalar':
                # numJoints == nDims == 1 (poseregnettrainer.py:84-85, 92-93): the (B, 1) output broadcast against the vector y
                if self.out_dim != 1:
                    raise ValueError("the scalar cost needs a one-dimensional output")
                if dp is not None and train:
                    raise NotImplementedError("the broadcast cost of the scalar target couples all samples of the global batch")
                self.lossplan.add(ops.loss_sse_bcast(rt, self.out.buf, self.y_in, self.N, self.cost, self.out.grad if train else None,
                                                     err=None if train else self.err))
            else:
                last = self.fwd.ops[-1][0] if self.fwd.ops else None
                fus = getattr(self, '_loss_fusable', None)
                if (train and hz.FUSE_LOSS and fus is not None and fus[0] is last and fus[1]['out'] is self.out.buf
                        and self.N * self.out_dim <= 65536):
                    # the split-K reduction of the last HiddenLayer is the forward plan's last launch: it computes the cost and its
                    # gradient as well (one launch less between the forward and the backward chain; running the forward plan alone
                    # still leaves `out`)
                    f = fus[1]
                    self.fwd.ops[-1] = (ops.reduce_partials_loss(rt, f['partial'], f['nz'], self.N, self.out_dim, self.out.buf, f['bias'], self.y_in,
                                                                 denom, self.cost, self.out.grad), False)
                else:
                    self.lossplan.add(ops.loss_sse(rt, self.out.buf, self.y_in, self.N, self.out_dim, denom, self.cost,
                                                   self.out.grad if train else None))
            if self.weight_decay and not net.hasDropout():
                # cost += weightreg_factor * sum(W^2) over the conv / FC weights, only for nets without dropout
                # (poseregnettrainer.py:101-107); the kernel-layout W holds the same values in another order
                # (data parallel: every rank's cost is its share of the global cost, so each adds 1/world of the regulariser)
                share = self.weight_decay / (dp.world if (dp is not None and train) else 1)
                seg = self._weight_segments()
                if seg is not None:                      # all weights in one launch over the flat parameter buffer
                    self.lossplan.add(ops.sumsq_multi(rt, self.store.w, seg[0], seg[1], share, self.cost, 1))
                else:
                    for W in self._unique_weights():
                        self.lossplan.add(ops.sumsq(rt, self.store.view(W), int(np.prod(W.shape)), share, self.cost, 1))
            if not train and loss.get('kind') != 'scalar':
                d = loss['nDims'] if loss.get('kind', 'embedding') == 'joints' else self.out_dim
                rows = self.N * (loss['numJoints'] if loss.get('kind', 'embedding') == 'joints' else 1)
                self.lossplan.add(ops.error_l2(rt, self.out.buf, self.y_in, rows, d, self.err))
        if train:
            if loss is None:
                raise ValueError("a training engine needs a loss")
            self.store.ensure_train_buffers()
            self.hyper = rt.alloc(8)
            self.early_side = []
            self._early_slice, self._early_work = None, [None]
            self._early_adam = None
            self._wtrans_jobs = []
            self._emit_backward()
            if self._wtrans_jobs:
                self.early_side.append(ops.conv3x3_wtrans_multi(rt, self._wtrans_jobs))
            if self.early_side:
                self.fwd.ops[0:0] = [(ops.Fork(), False)] + [(o, True) for o in self.early_side]
                self.fwd.uses_side = True
                self.bwd.ops[0:0] = [(ops.Join(), False)]
            self._grad_allreduce = []
            if dp is not None:
                dp.broadcast_store(self.store)
                # what the early bucket (started inside the backward plan, see _bwd_fc) does not cover
                lo, hi = self._early_slice if self._early_slice is not None else (0, 0)
                n = self.store.g.size
                if self._early_slice is not None:
                    self._grad_allreduce.append(dp.wait_op(self._early_work, 'grad_allreduce_wait_early'))
                for a, b in ((0, lo), (hi, n)) if self._early_slice is not None else ((0, n),):
                    if b > a:
                        self._grad_allreduce.append(dp.allreduce_sum_op(self.store.g.view(a, (b - a,)), 'grad_allreduce'))
                self._grad_allreduce += self._wd_ops       # once, on the all-reduced gradient (see _emit_backward)
                for op in self._grad_allreduce:
                    self.upd.add(op)
            # ONE launch: the update over the whole flat buffer, whose last workgroup advances t (round 6: the adam_tick launch behind
            # it was 5 us at the end of every step)
            self.upd.add(ops.adam(rt, self.store.w, self.store.g, self.store.m, self.store.v, self.store.n_w, self.hyper, tick=hz.ADAM_TICKED))
            if not hz.ADAM_TICKED:
                self.upd.add(ops.adam_tick(rt, self.hyper))
            if self.dropout_masks:
                self.upd.add(ops.counter_add(rt, self.step_ctr, 1))
            self._lr = None
            self.reset_optimizer()

    # ------------------------------------------------------------------------------------------ helpers
    def scratch(self, nfloats):
        if self._scratch is None or self._scratch.size < nfloats:
            self._scratch = self.rt.alloc(max(nfloats, 1 << 20), zero=False)
        return self._scratch

    def scratch_side(self, nfloats):
        """Scratch of the parameter-gradient branch (runs on the side stream, so it must not share the main arena)."""
        if getattr(self, '_scratch2', None) is None or self._scratch2.size < nfloats:
            self._scratch2 = self.rt.alloc(max(nfloats, 1 << 20), zero=False)
        return self._scratch2

    def _new_tensor(self, shape, name, act=False):
        """act: a convolution's output map -- the tensors that are bf16-stored in the bf16 mode."""
        t = TensorV(self.rt.alloc(shape, np.uint16 if (act and self.store16) else np.float32, zero=False), shape, name)
        self.tensors.append(t)
        return t

    def _drop_tensor(self, t):
        """Forget a tensor that was allocated for a launch that is not emitted after all."""
        self.tensors = [u for u in self.tensors if u is not t]

    def _gemm_prec(self, variant, K, lazy=False, role='fwd'):
        """dpp_gemm_desc.precision of a 1x1-convolution product in this net: 1 (bf16 MFMA operands, f32 accumulation) in the bf16 mode
        wherever the kernel of `variant` has the path -- round 4: the wave-autonomous kernel (variant 4) with whole 32-deep steps; round 6:
        the LDS-tiled kernel with chunks of 32 / 64 (K > 16), the K-split kernel, the 16-column stream (K = 16 with a zero upper half) --
        never with the two-tensor BatchNorm-backward operand (`lazy`) or on the row-stream kernel (variant 1)."""
        if not (self.prec and hz.BF16_GEMM) or lazy or variant == 1:
            return 0
        if variant == 4:
            return int(K >= 32 or (K == 16 and hz.BF16_GEMM_ALL))          # (K = 16: a 32-deep step with a zero upper half, round 6)
        if not hz.BF16_GEMM_ALL or role not in hz.BF16_GEMM_ROLES or str(variant) not in hz.BF16_GEMM_VARIANTS:
            return 0
        return int(variant in (2, 3) or K > 16)

    def _act(self, view):
        mode = (Act.BN if view.bn is not None else 0) | (Act.RELU if view.relu else 0)
        if mode == 0:
            return None
        if view.bn is not None:
            b = view.bn
            return ops.act(mode, b.mean, b.scale, b.beta_buf, view.chan)
        return ops.act(mode, None, None, None, view.chan)

    def _grad_view(self, param):
        """Where a layer's backward pass writes the gradient of `param`: its slot of the flat gradient buffer -- or, for the
        second and later layers that SHARE the parameter, a private buffer that is added to the slot after the pass (the
        gradient of a shared weight is the sum over its uses)."""
        uses = self.__dict__.setdefault('_grad_uses', {})
        n = uses.get(param.auto_name, 0)
        uses[param.auto_name] = n + 1
        slot = self.store.view(param, 'g')
        if n == 0:
            return slot
        priv = self.rt.alloc(slot.size, zero=False)
        self.__dict__.setdefault('_shared_grad_adds', []).append((slot, priv))
        return priv

    def _unique_weights(self):
        seen, out = set(), []
        for l in self.layers:
            if hasattr(l, 'W') and l.W.auto_name not in seen:
                seen.add(l.W.auto_name)
                out.append(l.W)
        return out

    def _weight_segments(self):
        """(device table, count) of the (offset, length) pairs of every conv / FC weight inside the flat TRAINED buffer, for the
        one-launch regulariser (ops.sumsq_multi / axpy_multi); None when a weight lives elsewhere (frozen: per-layer launches)."""
        if '_wseg' not in self.__dict__:
            st = self.store
            Ws = self._unique_weights()
            ok = bool(Ws) and all(st.slots[st.by_param[W.auto_name]]['trained'] for W in Ws)
            self._wseg = ops.segment_table(self.rt, st.w, [st.view(W) for W in Ws]) if ok else None
        return self._wseg

    def _fc1_stream(self, Nb, K, Nout):
        if not hz.is_fc1_shape(Nb, K, Nout) or hz.FC1_STREAM == '0':
            return False
        if hz.FC1_STREAM == '1' or self.prec == 1:
            return True
        if hz.FC1_STREAM != 'auto':
            return False
        # f32: whole 128-row tiles in all three GEMMs (forward / data gradient rows = samples, filter gradient reduces over them),
        # whole 32-deep chunks per K slice.  This is the one kernel choice that DOES depend on the batch (hz.stream16_plan's rule, above,
        # is about the convolutions): batches that are no multiple of 128 stay on dpp_gemm, whose K order differs.  The bar that
        # bounds the difference: tests/test_full_size.py evaluates the same frames in batches of 8 (dpp_gemm) and 128 (this kernel)
        # and holds the joints to 1e-4 mm.
        splitk = max(1, min(hz.FC1_SLICES, K // 512))
        return Nb % 128 == 0 and K % 128 == 0 and Nout % 64 == 0 and K % splitk == 0 and (K // splitk) % 32 == 0

    def _single_consumer(self, var):
        return len(self.consumers.get(id(var), [])) == 1

    def _materialize_plain(self, view):
        if view.plain:
            return view.base
        raise NotImplementedError("the net output must be a materialised tensor (conv / hidden layer output)")

    # ------------------------------------------------------------------------------------------ forward
    def _emit(self, var, residual=None, out_var=None):
        if residual is None and id(var) in self._memo:
            return self._memo[id(var)]
        k = var.kind
        if k == 'input':
            v = View(self.input_of.get(id(var), self.x_in))
        elif k == 'flatten':
            src = self._emit(var.inputs[0])
            n = src.shape[0]
            v = View(src.base, src.bn, src.relu, shape=(n, int(np.prod(src.shape[1:]))), chan=src.base.C)
        elif k == 'relu':
            src = self._emit(var.inputs[0], residual)
            if src.relu:
                v = src
            else:
                v = View(src.base, src.bn, True, shape=src.shape, chan=src.chan)
        elif k == 'add':
            v = self._emit_add(var)
        elif k == 'concat':
            v = self._emit_concat(var)
        elif k == 'layer':
            v = self._emit_layer(var, residual, out_var)
        else:
            raise NotImplementedError("graph node '%s'" % k)
        if residual is None:
            self._memo[id(var)] = v
        return v

    def _emit_concat(self, var):
        """T.concatenate of flattened tower outputs (scalenet.py:167-171): every part is packed (with its pending ReLU) into
        its column range of one [N][sum] buffer, which the following HiddenLayer reads as a plain operand."""
        parts = [self._emit(p) for p in var.inputs]
        if any(p.base.is16 for p in parts):
            raise NotImplementedError("concatenation of bf16-stored maps")
        N = parts[0].shape[0]
        widths = [int(p.shape[1]) for p in parts]
        total = sum(widths)
        cat = self._new_tensor((N, total), 'concat')
        cat.concat_parts = []
        off = 0
        for p, w in zip(parts, widths):
            if p.bn is not None or len(p.shape) != 2 or int(np.prod(p.base.shape[1:])) != w:
                raise NotImplementedError("concatenation of anything but flattened (ReLU) maps")
            self.fwd.add(ops.copy2d(self.rt, p.base.buf, w, cat.buf.view(off, (N * total - off,)), total, N, w, relu=p.relu,
                                    name='concat_pack'))
            cat.concat_parts.append((p, off, w))
            off += w
        return View(cat)

    def _split_concat_grad(self, cat):
        """Backward of the concatenation: the column ranges of d(concat) go back to the parts' view gradients."""
        N, total = cat.shape
        for p, off, w in cat.concat_parts:
            if p.base in self.x_ins:
                continue
            tgt, dst = self._view_grad(p)
            if tgt.grad_written:
                raise NotImplementedError("concatenated value with another consumer")
            self.bwd.add(ops.copy2d(self.rt, cat.grad.view(off, (N * total - off,)), total, dst, w, N, w, name='concat_split'))
            tgt.grad_written = True

    def _emit_add(self, var):
        a, b = var.inputs
        if self.fuse_blocks:
            fused = evalfuse.emit_block(self, var)
            if fused is not None:
                return fused

        def fusable(x):
            return x.kind == 'layer' and _layer_kind(x.layer) in ('ConvLayer',) and self._single_consumer(x) \
                and x.layer.cfgParams.activation is None

        cand = [x for x in (a, b) if fusable(x)]
        if not cand:
            raise NotImplementedError("residual add whose operands are not conv outputs")
        # The conv at the end of the LONGER branch absorbs the add in its epilogue (ties: the later layer).  In a projection block
        # (resnet.py:117-123: c + sc) that is the bottleneck exit c, not the shortcut sc -- which then depends on nothing but the
        # block's input and runs on the second stream BESIDE the bottleneck's first two convolutions instead of behind them: the forward
        # pass is a pure dependent chain (profiles/r04_whatif.txt), and the three projection convolutions were 72 us of it.
        q = max(cand, key=lambda x: (self._depth(x) if hz.SIDE_SHORTCUT else 0, x.layer.layerNum))
        p = b if q is a else a
        beside = hz.SIDE_SHORTCUT and self.train and p in cand and id(p) not in self._memo
        if beside:
            self._emit(p.inputs[0])                          # its input chain belongs to the main stream (normally emitted already)
            n0 = len(self.fwd.ops)
        pv = self._emit(p)
        if not pv.plain:
            raise NotImplementedError("residual add on a non-materialised operand")
        if beside:
            moved = list(self.fwd.ops[n0:])
            if all(isinstance(op, ops.Launch) for op, _ in moved):
                fork = ops.Fork()
                self.fwd.ops[n0:] = [(fork, False)] + [(op, True) for op, _ in moved]
                self.fwd.uses_side = True
                self._emit(q.inputs[0])                      # the longer branch, on the main stream
                self.fwd.join()                              # ... whose last convolution reads the shortcut as its residual
                # (remembered: a step whose second stream is busy with other work during the forward pass -- step_plan(early=) --
                #  keeps the shortcuts in the chain instead of queueing them behind that work)
                self._beside.append(([id(fork), id(self.fwd.ops[-1][0])], [id(op) for op, _ in moved]))
        if p.kind == 'layer' and p.layer.layerNum > q.layer.layerNum:
            self._bwd_after[id(p.layer)] = q.layer           # backward: the sum's gradient reaches p through q's identity path
        out = self._emit(q, residual=pv.base, out_var=var)      # the conv's output tensor IS the sum
        self._memo[id(q)] = out
        return out

    def _depth(self, var):
        """Number of layers on the longest path from the inputs to this graph node."""
        memo = self.__dict__.setdefault('_depth_memo', {})
        if id(var) not in memo:
            d = max([self._depth(i) for i in getattr(var, 'inputs', None) or []] or [0])
            memo[id(var)] = d + (1 if var.kind == 'layer' else 0)
        return memo[id(var)]

    def _feeds_batchnorm(self, var):
        """Does a BatchNorm consume the tensor this var denotes?  Then its producer emits the statistics partials."""
        for c in self.consumers.get(id(var), []):
            if c.kind == 'layer' and _layer_kind(c.layer) == 'BatchNormLayer':
                return True
        return False

    def _emit_layer(self, var, residual, out_var=None):
        layer = var.layer
        kind = _layer_kind(layer)
        rt, st = self.rt, self.store
        src = self._emit(var.inputs[0])
        if kind == 'BatchNormLayer':
            if not src.plain:
                raise NotImplementedError("BatchNorm on a non-materialised input")
            C = src.base.C
            M = src.base.rows
            sync = self.train and self.dp is not None and self.dp.sync_bn
            b = BNState(layer, C, M, self.dp.world if sync else 1)
            b.mean, b.inv_std, b.scale = (rt.alloc(_pad4(C)) for _ in range(3))
            b.beta_buf, b.gamma_buf = st.view(layer.beta), st.view(layer.gamma)
            b.run_mean, b.run_inv_std = st.view(layer.mean), st.view(layer.inv_std)
            self.bn_states[id(layer)] = b
            fused = getattr(src.base, 'stats', None) if self.train else None
            if fused is not None:
                # the producing conv already wrote per-block (mean, M2): only the combine is left
                part, nblk, rows = fused
                self.fwd.add(ops.bn_finalize(rt, part, nblk, M, rows, C, b.gamma_buf, layer.cfgParams.epsilon, b.mean, b.inv_std,
                                             b.scale, b.run_mean, b.run_inv_std, layer.cfgParams.alpha))
            elif self.train:
                W = b.world
                part = self.scratch(b.nb * 2 * C * (W + 1)).view(0, (b.nb * 2 * C,))
                self.fwd.add(ops.bn_stats_partial(rt, src.base.buf, M, C, b.rpb, part))
                if W > 1:
                    allp = self.scratch(b.nb * 2 * C * (W + 1)).view(b.nb * 2 * C, (W * b.nb * 2 * C,))
                    self.fwd.add(self.dp.all_gather_op(part, allp, 'bn_stats_allgather'))
                    part = allp
                self.fwd.add(ops.bn_finalize(rt, part, b.nb, M * W, b.rpb, C, b.gamma_buf, layer.cfgParams.epsilon, b.mean, b.inv_std,
                                             b.scale, b.run_mean, b.run_inv_std, layer.cfgParams.alpha, nseg=W))
            elif hz.EVAL_FUSE:
                self._bn_eval_jobs.append((b.gamma_buf, b.run_mean, b.run_inv_std, C, b.mean, b.inv_std, b.scale))
            else:
                self.fwd.add(ops.bn_eval_coeffs(rt, b.gamma_buf, b.run_mean, b.run_inv_std, C, b.mean, b.inv_std, b.scale))
            v = View(src.base, b, False, shape=src.shape, chan=src.chan)
            v.var = var
            return v
        if kind == 'NonlinearityLayer':
            if layer.cfgParams.activation is None:
                return src
            if layer.cfgParams.activation_str != 'ReLU':
                raise NotImplementedError("only ReLU is on the hot path")
            v = View(src.base, src.bn, True, shape=src.shape, chan=src.chan)
            v.var = var
            return v
        if kind == 'ConvPoolLayer':
            return self._emit_stem(layer, src, out_var if out_var is not None else var)
        if kind == 'ConvLayer':
            return self._emit_conv(layer, src, residual, out_var if out_var is not None else var)
        if kind == 'HiddenLayer':
            return self._emit_fc(layer, src)
        if kind == 'DropoutLayer':
            return self._emit_dropout(layer, src)
        raise NotImplementedError(kind)

    def _emit_stem(self, layer, src, out_var=None):
        c = layer.cfgParams
        N, H, W, Ci = src.base.shape
        ok = (src.base is self.x_in and Ci == 1 and tuple(c.filterDim) == (5, 5) and c.border_mode == 'half' and
              tuple(c.poolsize) == (2, 2) and tuple(c.stride) == (1, 1) and c.activation is None and c.nFilters <= 32)
        if not ok:
            return self._emit_convpool(layer, src)
        Co = c.nFilters
        out = self._new_tensor((N, H // 2, W // 2, Co), 'stem', act=True)
        arg = self.rt.alloc((N, H // 2, W // 2, Co), np.uint8, zero=False) if self.train else None
        stats = None
        if self.train and self.fuse_bn and H % 16 == 0 and W % 16 == 0 and out_var is not None and self._feeds_batchnorm(out_var):
            nblk = N * (H // 16) * (W // 16)                   # one partial per workgroup = 64 pooled outputs
            out.stats = (self.rt.alloc((nblk, 2, Co), zero=False), nblk, 64)
            stats = out.stats[0]
        self.fwd.add(ops.stem_fwd(self.rt, src.base.buf.reshape(N, H, W), N, H, W, self.store.view(layer.W), self.store.view(layer.b), Co,
                                  out.buf, arg, stats))
        self.layer_io[id(layer)] = dict(in_view=src, out=out, argmax=arg, stem=True)
        return View(out)

    def _emit_convpool(self, layer, src):
        """Generic ConvPoolLayer (PoseRegNet's front end, poseregnet.py:62-78): VALU conv + max-pool + bias kernels."""
        c = layer.cfgParams
        N, H, W, Ci = src.base.shape
        kh, kw = c.filterDim[0], c.filterDim[1]
        if len(src.shape) != 4:
            raise NotImplementedError("ConvPoolLayer on a flattened input")
        if self.store16:
            raise NotImplementedError("bf16 storage is built for the ResNet path (stem + ConvLayer); the generic ConvPoolLayer kernels are f32")
        if c.border_mode not in ('valid', 'half') or tuple(c.stride) != (1, 1) or c.poolsize[0] != c.poolsize[1]:
            raise NotImplementedError("ConvPoolLayer border %s stride %s pool %s" % (c.border_mode, c.stride, c.poolsize))
        if c.border_mode == 'half' and (kh % 2 == 0 or kw % 2 == 0):
            raise NotImplementedError("'half' padding with an even filter size")
        pad = kh // 2 if c.border_mode == 'half' else 0
        if c.border_mode == 'half' and kh != kw:
            raise NotImplementedError("'half' padding with a non-square filter")
        pool, Co = int(c.poolsize[0]), c.nFilters
        _, Co_, Hp, Wp = c.outputDim
        assert Co_ == Co
        out = self._new_tensor((N, Hp, Wp, Co), 'convpool%d' % layer.layerNum)
        ties = self.rt.alloc((N, Hp, Wp, Co), np.uint16, zero=False) if (self.train and pool > 1) else None
        geom = dict(N=N, H=H, W=W, Ci=Ci, kh=kh, kw=kw, pad=pad, Co=Co, pool=pool, Hp=Hp, Wp=Wp)
        self.fwd.add(ops.convpool_fwd(self.rt, src.base.buf, N, H, W, Ci, self.store.view(layer.W), kh, kw, pad, Co, pool,
                                      self.store.view(layer.b), out.buf, ties, actX=self._act(src), name='convpool_%d' % layer.layerNum))
        self.layer_io[id(layer)] = dict(in_view=src, out=out, ties=ties, geom=geom, stem=False)
        return View(out)

    def _emit_conv(self, layer, src, residual, out_var=None):
        c = layer.cfgParams
        N, Hi, Wi, Ci = src.base.shape
        _, Co, Ho, Wo = c.outputDim
        k, s = tuple(c.filterDim), tuple(c.stride)
        if c.border_mode != 'half' or k not in ((1, 1), (3, 3)) or s[0] != s[1] or (k == (3, 3) and s != (1, 1)):
            raise NotImplementedError("ConvLayer %s stride %s border %s" % (k, s, c.border_mode))
        out = self._new_tensor((N, Ho, Wo, Co), 'conv%d' % layer.layerNum, act=True)
        act = self._act(src)
        rt, st = self.rt, self.store
        res = residual.buf if residual is not None else None
        # fused BatchNorm statistics of the tensor being written (not with sync-BN: its partials are all-gathered by block)
        want_stats = self.train and self.fuse_bn and out_var is not None and self._feeds_batchnorm(out_var)
        M = N * Ho * Wo
        epi = None
        if k == (1, 1):
            tile, _ = hz.gemm_plan(M, Co, Ci, allow_split=False)
            rs = hz.rowstream_plan(M, Co, Ci, True)
            if rs is not None:
                tile = rs
            ks = hz.ksplit_plan(M, Co, Ci) if (s[0] == 1 and rs is None) else None
            if ks is not None:
                tile = ks
            ex = hz.expand_plan(M, Co, Ci, True) if (s[0] == 1 and rs is None and ks is None) else None
            s16 = hz.stream16_plan(M, Co, Ci, True) if (s[0] == 1 and rs is None and ks is None and ex is None) else None
            if s16 is not None:
                tile = s16
            if ex is not None:
                tile = ex
            mp = RowMap.strided(s[0], Ho, Wo, Hi, Wi) if s[0] != 1 else None
            variant = 1 if rs is not None else (2 if ks is not None else (4 if ex is not None else (3 if s16 is not None else 0)))

            def build(tile, variant, epi):
                # (bf16 mode: bf16 MFMA operands wherever the kernel of the variant has them, _gemm_prec)
                return ops.gemm(rt, src.base.buf, st.view(layer.W), out.buf, M, Co, Ci, 1, 1, Ci, Ci, Co, mapA=mp, actA=act,
                                bias=st.view(layer.b), residual=res, tile=tile, epi=epi, variant=variant,
                                name='conv1x1_%d' % layer.layerNum, precision=self._gemm_prec(variant, Ci))
            if variant in (2, 3, 4) and ops.gemm_variant_rows(rt, build(tile, variant, None)) != tile[0]:
                # the shape asks for the kernel, the buffers rule it out (alignment / prologue): the generic tile, not a failed build
                variant, (tile, _) = 0, hz.gemm_plan(M, Co, Ci, allow_split=False)
            if want_stats:
                nblk = -(-M // tile[0])
                out.stats = (rt.alloc((nblk, 2, Co), zero=False), nblk, tile[0])
                epi = ops.epilogue(stats=out.stats[0])
            self.fwd.add(build(tile, variant, epi))
        elif self._conv3_stream(N, Hi, Wi, Ci, Co, src.base.buf, out.buf) and residual is None:
            rows = rt.lib.dpp_conv3x3_stream_rows(N, Hi, Wi, Ci)
            if want_stats:
                out.stats = (rt.alloc((M // rows, 2, Co), zero=False), M // rows, rows)
                epi = ops.epilogue(stats=out.stats[0])
            self.fwd.add(ops.conv3x3_stream(rt, src.base.buf, N, Hi, Wi, Ci, st.view(layer.W), out.buf, actX=act, bias=st.view(layer.b), epi=epi,
                                            name='conv3x3_%d' % layer.layerNum))
        else:
            bm = hz.conv3x3_bm(M, Co, hw=(Hi, Wi), prec=self.prec)
            if want_stats:
                th, tw, img = (C_int() for _ in range(3))
                nblk = rt.lib.dpp_conv3x3_tiling(N, Hi, Wi, bm, th, tw, img)
                if Hi % th.value == 0 and Wi % tw.value == 0 and (N % img.value == 0 or nblk == 1):   # every block holds bm pixels
                    out.stats = (rt.alloc((nblk, 2, Co), zero=False), nblk, bm)
                    epi = ops.epilogue(stats=out.stats[0])
            self.fwd.add(ops.conv3x3(rt, src.base.buf, N, Hi, Wi, Ci, st.view(layer.W), Co, out.buf, actX=act, bias=st.view(layer.b),
                                     residual=res, bm=bm, epi=epi, name='conv3x3_%d' % layer.layerNum, precision=self.prec))
        self.layer_io[id(layer)] = dict(in_view=src, out=out, residual=residual)
        return View(out)

    def _conv3_stream(self, N, H, W, Ci, Co, *bufs):
        """Whether a 3x3 layer (or its data gradient) runs on dpp_conv3x3_stream: a narrow square layer on float32 tensors in the
        float32 mode (the bf16 mode multiplies the 3x3 layers on the bf16 matrix pipe of the LDS-tiled kernel)."""
        if Ci != Co or Ci not in hz.CONV3_STREAM_C or self.prec:
            return False
        if any(b is not None and b.dtype == ops.BF16 for b in bufs):
            return False
        return self.rt.lib.dpp_conv3x3_stream_rows(N, H, W, Ci) > 0

    def _emit_fc(self, layer, src):
        c = layer.cfgParams
        if len(src.shape) != 2:
            raise NotImplementedError("HiddenLayer on a non-flattened input")
        Nb, K = src.shape
        Nout = c.outputDim[1]
        assert K == c.inputDim[1], (K, c.inputDim)
        out = self._new_tensor((Nb, Nout), 'fc%d' % layer.layerNum)
        rt, st = self.rt, self.store
        act = self._act(src)
        if self._fc1_stream(Nb, K, Nout):
            splitk = max(1, min(hz.FC1_SLICES, K // 512))
            part = self.scratch(splitk * Nb * Nout)
            self.fwd.add(ops.fc_gemm(rt, src.base.buf, st.view(layer.W), None, Nb, Nout, K, 1, 0, K, Nout, Nout, actA=act, splitk=splitk,
                                     partial=part, precision=self.prec, kchunk=hz.FC1_KCHUNK, name='fc_%d' % layer.layerNum))
            self.fwd.add(ops.reduce_partials(rt, part, splitk, Nb * Nout, out.buf, bias=st.view(layer.b), nbias=Nout))
            self.layer_io[id(layer)] = dict(in_view=src, out=out)
            return View(out)
        tile, splitk = hz.gemm_plan(Nb, Nout, K)
        if K >= 4096 and Nb <= 128 and Nout >= 64:
            # weight-streaming shape (FC1: 67 MB of W for 128 rows): wide column tiles read W in 256 B rows, K split 512 deep
            # (tools/gemm_micro.py fc: all 128 rows in one tile halve the passes over W through L2, 96 -> 79 us)
            tile, splitk = ((128, 64, 4) if Nb > 64 and hz.knob('DPP_FC1_TILE128', '1') != '0' else (64, 64, 4)), max(1, K // 512)
        act = self._act(src)
        if splitk > 1:
            part = self.scratch(splitk * Nb * Nout)
            self.fwd.add(ops.gemm(rt, src.base.buf, st.view(layer.W), None, Nb, Nout, K, 1, 0, K, Nout, Nout, actA=act, splitk=splitk,
                                  partial=part, tile=tile, name='fc_%d' % layer.layerNum))
            red = self.fwd.add(ops.reduce_partials(rt, part, splitk, Nb * Nout, out.buf, bias=st.view(layer.b), nbias=Nout))
            self._loss_fusable = (red, dict(partial=part, nz=splitk, bias=st.view(layer.b), out=out.buf))       # see the loss plan (__init__)
        else:
            self.fwd.add(ops.gemm(rt, src.base.buf, st.view(layer.W), out.buf, Nb, Nout, K, 1, 0, K, Nout, Nout, actA=act,
                                  bias=st.view(layer.b), tile=tile, name='fc_%d' % layer.layerNum))
        self.layer_io[id(layer)] = dict(in_view=src, out=out)
        return View(out)

    def _emit_dropout(self, layer, src):
        if src.bn is not None or len(src.shape) != 2:
            raise NotImplementedError("DropoutLayer after BatchNorm / on a 4-D map")
        n = int(np.prod(src.shape))
        out = self._new_tensor(src.shape, 'drop%d' % layer.layerNum)
        keep = np.float32(1.0 - layer.cfgParams.p)
        mask = None
        if self.train:
            mask = self.rt.alloc(src.shape)
            self.dropout_masks[id(layer)] = (mask, float(keep), layer.mask_seed)
            # a fresh Bernoulli(1-p) mask per step (dropoutlayer.py:98-103): stream keyed by (layer seed, layer, device step counter)
            # (data parallel: ranks build the net from the same seed, so the rank enters the stream key -- shards get independent masks)
            rank = self.dp.rank if self.dp is not None else 0
            self.fwd.add(ops.bernoulli_mask(self.rt, mask, n, float(keep), layer.mask_seed, (layer.layerNum << 40) + (rank << 32), self.step_ctr))
        self.fwd.add(ops.scale(self.rt, src.base.buf, out.buf, n, a=keep, relu=src.relu, mask=mask))
        self.layer_io[id(layer)] = dict(in_view=src, out=out, mask=mask)
        return View(out)

    # ------------------------------------------------------------------------------------------ backward
    def set_input(self, x):
        """x: (N, C, H, W) host array (the reference's NCHW crops)."""
        xs = x if isinstance(x, (list, tuple)) else [x]
        if len(xs) != len(self.x_ins):
            raise ValueError("the net takes %d inputs, got %d" % (len(self.x_ins), len(xs)))
        for t, a in zip(self.x_ins, xs):
            a = np.asarray(a, np.float32)
            want = (t.shape[0], t.shape[3], t.shape[1], t.shape[2])
            if tuple(a.shape) != want:
                raise ValueError("input shape %s, expected %s" % (a.shape, want))
            t.buf.set(a if a.shape[1] == 1 else layout.nchw_to_nhwc(a))

    def forward(self, x=None):
        self.store.check_live()
        if x is not None:
            self.set_input(x)
        self.fwd.run(self.rt)
        return self.out.buf.get()

    def reset_optimizer(self, lr=0.0):
        """ADAM state of optimizer.py:58-90: t = 1, m = v = 0; python-float constants become floatX (float32) constants,
        so gamma = 1 - 1e-8 is exactly 1 as in the reference."""
        f = np.float32
        o = self.optimizer
        if o.get('name', 'ADAM') == 'RMSProp':              # optimizer.py:92-116; slots 3 / 4 = decay / epsilon, slot 6 selects the rule
            self.hyper.set(np.array([f(lr), f(1.0), 0, f(o.get('decay', 0.9)), f(o.get('epsilon', 1.0 / 100.)), 0, 1, 0], np.float32))
        else:
            self.hyper.set(np.array([f(lr), f(1.0), f(o.get('beta1', 0.9)), f(o.get('beta2', 0.999)), f(o.get('epsilon', 1e-8)),
                                     f(o.get('gamma', 1 - 1e-8)), 0, 0], np.float32))
        self.store.m.zero()
        self.store.v.zero()
        self._lr = float(lr)

    def set_lr(self, lr):
        if self._lr != float(np.float32(lr)):
            self.hyper.view(0, (1,)).set(np.array([lr], np.float32))
            self._lr = float(np.float32(lr))

    def run_step_plans(self, allreduce=None):
        """forward + loss + backward (+ gradient all-reduce) + ADAM on x_in / y_in; no host<->device traffic, so the
        whole sequence can be captured into a hipGraph (runtime.capture) and replayed."""
        st = self.rt
        self.store.check_live()
        if allreduce is None:
            self.step_plan().run(st)
            return
        self.fwd.run(st)
        self.lossplan.run(st)
        self.bwd.run(st)
        allreduce(self.store.g)
        self.upd.run(st)

    def step_plan(self, before=None, prefetch=None, early=None):
        """forward + loss + backward + update as ONE plan (one native call per step); `before`: a Plan issued ahead of the
        forward pass inside the same call (the augmentation kernels).  `prefetch`: a Plan whose launches run on the gradient
        branch between the backward pass and the update -- the augmentation of the NEXT step's minibatch, written straight into
        x_in / y_in (their last readers, the first layer's filter gradient and the loss, are done by then) while the main stream is
        busy with the ADAM update; the plan then opens with a join, so that the forward pass waits for the prefetch of the previous
        call.  What the reference does with its background augmentation processes (nettrainer.py:601-628).
        `early`: a Plan issued on the gradient branch at the START of the step, beside the forward pass (that stream is idle until
        the backward pass): work for the next minibatch that does not touch x_in / y_in -- the refinement cascade cropping into
        staging buffers, which `prefetch` then only has to copy."""
        key = (id(before) if before is not None else 0, id(prefetch) if prefetch is not None else 0, id(early) if early is not None else 0)
        cache = self.__dict__.setdefault('_step_plans', {})
        if key not in cache:
            bwd, upd = self._early_adam_plans()
            fwd = self.fwd
            if early is not None and self._beside:
                # the second stream belongs to `early` during the forward pass: the projection shortcuts stay in the chain
                marks = set(m for ms, _ in self._beside for m in ms)
                chain = set(o for _, os_ in self._beside for o in os_)
                fwd = Plan('forward')
                fwd.ops = [(op, side and id(op) not in chain) for (op, side) in self.fwd.ops if id(op) not in marks]
                fwd.uses_side = getattr(self.fwd, 'uses_side', False)
            parts = ([before] if before is not None else []) + [fwd, self.lossplan, bwd]
            if prefetch is not None:
                pre, post = Plan('prefetch_join'), Plan('prefetch')
                pre.join()
                post.fork()
                for op in prefetch.steps():
                    post.add(op, side=True)
                parts = [pre] + parts + [post]
            if early is not None:
                head = Plan('early')
                head.fork()
                for op in early.steps():
                    head.add(op, side=True)
                at = 1 if prefetch is not None else 0          # behind the join that waits for the previous call's prefetch
                parts = parts[:at] + [head] + parts[at:]
            plan = Plan.concat('step', parts + [upd])
            skip = tuple(x for x in hz.knob('DPP_WHATIF_SKIP', '').split(',') if x)
            if skip:
                # ablation for tools/whatif.sh ONLY (results are wrong): the step without the launches whose name starts with one of
                # the prefixes -- what the step would cost if those kernels were free
                plan.ops = [(op, side) for (op, side) in plan.ops if not str(getattr(op, 'name', '')).startswith(skip)]
            cache[key] = (before, prefetch, early, plan)
        return cache[key][-1]

    def _early_adam_plans(self):
        """(backward, update) plans of a whole step.  With hz.EARLY_ADAM the FC1 weight is updated inside the backward pass, on the
        gradient branch, right after the two kernels that use it (its weight-decay term first), and the update plan covers the rest of
        the flat buffer.  Single process only: with data parallelism the gradient is not final before its all-reduce."""
        ea = self._early_adam
        if not hz.EARLY_ADAM or ea is None or self.dp is not None:
            return self.bwd, self.upd
        if '_early_adam_cache' in self.__dict__:
            return self._early_adam_cache
        rt, st = self.rt, self.store
        lo, hi, n = ea['lo'], ea['hi'], st.n_w
        sl = lambda b, a, z: b.view(a, (z - a,))          # noqa: E731
        bwd = Plan('backward')
        early = [o for o in self._wd_ops if self._wd_of.get(id(o)) == ea['W'].auto_name]
        early.append(ops.adam(rt, sl(st.w, lo, hi), sl(st.g, lo, hi), sl(st.m, lo, hi), sl(st.v, lo, hi), hi - lo, self.hyper, name='adam_fc1'))
        pos = 1 + [i for i, (op, _) in enumerate(self.bwd.ops) if op is ea['after']][0]      # (an index would go stale: plans are edited)
        head = list(self.bwd.ops[:pos])
        tail = [(op, side) for (op, side) in self.bwd.ops[pos:] if not any(op is e for e in early)]
        bwd.ops = head + [(ops.Fork(), False)] + [(o, True) for o in early] + tail
        bwd.uses_side = True
        upd = Plan('update')
        for a, z in ((0, lo), (hi, n)):
            if z > a:
                upd.add(ops.adam(rt, sl(st.w, a, z), sl(st.g, a, z), sl(st.m, a, z), sl(st.v, a, z), z - a, self.hyper))
        if hz.ADAM_TICKED:
            upd.add(ops.adam_tick(rt, self.hyper))          # the update is several launches here: t advances behind the last of them
        for (op, side) in self.upd.ops:
            if getattr(op, 'name', '') != 'adam':
                upd.add(op, side)
        self._early_adam_cache = (bwd, upd)
        return self._early_adam_cache

    def train_step_device(self, lr, allreduce=None):
        self.set_lr(lr)
        self.run_step_plans(allreduce)

    def train_step(self, x, y, lr):
        self.set_input(x)
        self.y_in.set(np.asarray(y, np.float32).reshape(self.N, self.out_dim))
        self.train_step_device(lr)
        return float(self.cost.get()[0])

    def cost_and_grads(self, x, y):
        """forward + loss + backward only (for the parity tests): returns (cost, out)."""
        self.set_input(x)
        self.y_in.set(np.asarray(y, np.float32).reshape(self.N, self.out_dim))
        st = self.rt
        self.fwd.run(st)
        self.lossplan.run(st)
        self.bwd.run(st)
        return float(self.cost.get()[0]), self.out.buf.get()

    def global_cost(self):
        """The cost of the GLOBAL minibatch: eng.cost holds this rank's share (its shard's sum over the global batch size), the
        sum over ranks is the reference's cost.  A collective: every rank must call it."""
        c = float(self.cost.get()[0])
        if self.dp is None:
            return c
        import torch
        t = torch.tensor([c], dtype=torch.float64)
        if self.dp.dist.get_backend() == 'nccl':
            t = t.cuda()
        self.dp.dist.all_reduce(t, op=self.dp.dist.ReduceOp.SUM)
        return float(t.item())

    def allreduce_grads(self):
        """Finish summing the flat gradient buffer over the data-parallel ranks after a backward pass (no-op without dp):
        joins the early bucket started inside the backward plan and all-reduces the rest."""
        for op in self._grad_allreduce:
            op(self.rt.stream)

    def evaluate(self, x, y):
        """Deterministic forward + cost + error on one batch (validation functions of setupValidate)."""
        self.set_input(x)
        self.y_in.set(np.asarray(y, np.float32).reshape(self.N, self.out_dim))
        self.fwd.run(self.rt)
        self.lossplan.run(self.rt)
        return float(self.cost.get()[0]), float(self.err.get()[0])

    def num_launches(self):
        return dict(forward=len(self.fwd), backward=len(self.bwd), update=len(self.upd))

    def all_launches(self):
        return [('fwd', o) for o in self.fwd.launches()] + [('loss', o) for o in self.lossplan.launches()] + \
               [('bwd', o) for o in self.bwd.launches()] + [('upd', o) for o in self.upd.launches()]
