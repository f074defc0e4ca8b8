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
"""
import os
from ctypes import c_int as C_int

import numpy as np

from . import evalfuse, layout, ops
from . import heuristics as hz
from .lib import Act, RowMap
from .ops import Plan
from .runtime import default_runtime



def _pad4(n):
    return (n + 3) // 4 * 4


def bf16_bits_to_f32(a):
    """uint16 bfloat16 bit patterns -> float32 (exact)."""
    return (np.ascontiguousarray(a, np.uint16).astype(np.uint32) << 16).view(np.float32)


class TensorV(object):
    """A materialised device tensor: NHWC (N,H,W,C) or (N,D); float32, or bfloat16 bits (dtype uint16) for a bf16-STORED activation
    tensor (CompiledNet.store16)."""

    def __init__(self, buf, shape, name):
        self.buf, self.shape, self.name = buf, tuple(shape), name
        self.grad = None
        self.grad_written = False
        self.pending = []              # gradient buffers to be added (identity paths of fused residuals)

    @property
    def rows(self):
        return int(np.prod(self.shape[:-1]))

    @property
    def C(self):
        return self.shape[-1]

    @property
    def is16(self):
        return self.buf.dtype == ops.BF16

    def get_f32(self):
        """The tensor's values as float32, whatever its storage."""
        a = self.buf.get()
        return bf16_bits_to_f32(a).reshape(self.shape) if self.is16 else a


class BNState(object):
    def __init__(self, layer, C, M, world=1):
        self.layer, self.C, self.M = layer, C, M
        self.rpb = max(32, -(-M // hz.BN_RPB_TARGET_BLOCKS))
        if world > 1:
            # sync-BN concatenates the partials of all ranks: every block must hold exactly rpb rows
            self.rpb = M if M < 32 else 32
            while self.rpb * 2 <= max(32, M // hz.BN_RPB_TARGET_BLOCKS) and M % (self.rpb * 2) == 0:
                self.rpb *= 2
            if M % self.rpb:
                raise NotImplementedError("sync-BN needs the per-rank pixel count to be a multiple of %d" % self.rpb)
        self.nb = -(-M // self.rpb)
        self.world = world


class View(object):
    """base tensor + pending BatchNorm + pending ReLU; `shape` may be the flattened 2-D shape."""

    def __init__(self, base, bn=None, relu=False, shape=None, chan=None):
        self.base, self.bn, self.relu = base, bn, relu
        self.shape = tuple(shape) if shape is not None else base.shape
        self.chan = chan if chan is not None else base.C      # channel modulus along the contiguous dim

    @property
    def plain(self):
        return self.bn is None and not self.relu

    def key(self):
        return (id(self.base), id(self.bn) if self.bn is not None else 0, self.relu)


class ParamStore(object):
    """Flat device buffers for all parameters of a net.
    w / g / m / v : trained parameters (kernel layouts), gradients, ADAM moments; nt: running BN statistics."""

    def __init__(self, rt, slots_spec):
        self.rt = rt
        self.slots = []
        off = {'w': 0, 'nt': 0}
        for (param, kind, info, trained) in slots_spec:
            size = int(np.prod(param.shape))
            space = 'w' if trained else 'nt'
            self.slots.append(dict(param=param, kind=kind, info=info, trained=trained, off=off[space], size=size,
                                   shape=tuple(param.shape)))
            off[space] += _pad4(size)
        self.n_w, self.n_nt = max(4, off['w']), max(4, off['nt'])
        self.w = rt.alloc(self.n_w)
        self.nt = rt.alloc(self.n_nt)
        self.g = self.m = self.v = None
        self.by_param = {}
        for i, s in enumerate(self.slots):
            self.by_param[s['param'].auto_name] = i
            host = s['param']._host
            s['param']._binding = None
            self._write(i, host)
        for i, s in enumerate(self.slots):
            s['param']._binding = (self, i)

    def ensure_train_buffers(self):
        if self.g is None:
            self.g = self.rt.alloc(self.n_w)
            self.m = self.rt.alloc(self.n_w)
            self.v = self.rt.alloc(self.n_w)

    def view(self, param, space=None):
        s = self.slots[self.by_param[param.auto_name]]
        base = {'w': self.w, 'g': self.g, 'm': self.m, 'v': self.v, 'nt': self.nt}[space or ('w' if s['trained'] else 'nt')]
        return base.view(s['off'], (s['size'],))

    def _to_kernel(self, s, value):
        value = np.asarray(value, np.float32)
        if s['kind'] == 'conv_w':
            return layout.conv_w_to_kernel(value).reshape(-1)
        if s['kind'] == 'fc_w' and s['info'] is not None:
            return _fc_rows(value, s['info'], layout.fc_rows_nchw_to_nhwc).reshape(-1)
        return value.reshape(-1)

    def _from_kernel(self, s, flat):
        return layout.from_kernel(s['kind'], s['info'], s['shape'], flat)

    def _write(self, i, value):
        s = self.slots[i]
        base = self.w if s['trained'] else self.nt
        base.view(s['off'], (s['size'],)).set(self._to_kernel(s, value))

    # SharedParam binding interface
    def read_param(self, i):
        s = self.slots[i]
        base = self.w if s['trained'] else self.nt
        return self._from_kernel(s, base.view(s['off'], (s['size'],)).get())

    def write_param(self, i, value):
        self._write(i, value)

    def read_grad(self, param):
        s = self.slots[self.by_param[param.auto_name]]
        return self._from_kernel(s, self.g.view(s['off'], (s['size'],)).get())

    def bulk_values(self):
        """{auto_name: value in the reference's layout} of every parameter from TWO device -> host copies (the flat trained and
        non-trained buffers) instead of one round trip per parameter: what a checkpoint needs (NetBase.save)."""
        flat = {'w': self.w.get(), 'nt': self.nt.get()}
        out = {}
        for s in self.slots:
            base = flat['w' if s['trained'] else 'nt']
            out[s['param'].auto_name] = self._from_kernel(s, base[s['off']:s['off'] + s['size']])
        return out

    def snapshot(self, into=None):
        """Device copy of all TRAINED parameters (one device-to-device copy of the flat buffer): the epoch loop's "best weights so
        far" (nettrainer.py:871-876 pulls `weightVals` = all_params, the trained parameters only, to the host for that; the BatchNorm
        running statistics are not part of it, so early stopping keeps the FINAL statistics -- the same semantic as the host path)."""
        if into is None:
            into = (self.rt.alloc(self.n_w, zero=False),)
        self.rt.copy(into[0], self.w)
        return into

    def restore(self, snap):
        self.check_live()
        self.rt.copy(self.w, snap[0])

    released = False

    def release(self):
        """Pull every value back to the host copies and unbind (before the store is rebuilt).  Engines compiled on this store keep
        views into its buffers: they refuse to run from here on (`check_live`) instead of computing with weights nobody updates."""
        for i, s in enumerate(self.slots):
            s['param']._host = np.asarray(self.read_param(i), np.float32)
            s['param']._binding = None
        self.released = True

    def check_live(self):
        if self.released:
            raise RuntimeError("this engine was compiled before the net's parameter list changed (a layer was added or removed and the "
                               "device parameter store was rebuilt): compile the net again")


_fc_rows = layout.fc_rows


def _layer_kind(layer):
    return layer.__class__.__name__


def _collect(net):
    """Vars reachable from net.output, consumer counts, and the layers in list (= topological) order."""
    consumers = {}
    seen = {}
    order = []

    def walk(v):
        if id(v) in seen:
            return
        seen[id(v)] = v
        for i in v.inputs:
            consumers.setdefault(id(i), []).append(v)
            walk(i)
        order.append(v)

    walk(net.output)
    used = set(id(v.layer) for v in order if v.kind == 'layer')
    layers = [l for l in net.layers if id(l) in used]
    return order, consumers, layers


def _param_specs(net, layers):
    specs = []
    for l in layers:
        k = _layer_kind(l)
        if k in ('ConvLayer', 'ConvPoolLayer'):
            specs.append((l.W, 'conv_w', None, True))
            specs.append((l.b, 'vec', None, True))
        elif k == 'HiddenLayer':
            info = None
            iv = l.inputVar
            if iv.kind == 'flatten' and iv.inputs[0].shape is not None and len(iv.inputs[0].shape) == 4:
                _, Cc, H, W = iv.inputs[0].shape
                if H * W > 1:
                    info = (Cc, H, W)
            elif iv.kind == 'concat':
                info = tuple(tuple(f.inputs[0].shape[1:]) for f in iv.inputs)
            specs.append((l.W, 'fc_w', info, True))
            specs.append((l.b, 'vec', None, True))
        elif k == 'BatchNormLayer':
            specs.append((l.beta, 'vec', None, True))
            specs.append((l.gamma, 'vec', None, True))
            specs.append((l.mean, 'vec', None, False))
            specs.append((l.inv_std, 'vec', None, False))
    return specs


def _dedupe_specs(specs):
    """A parameter shared by several layers (copyLayer: ScaleNet's shared_conv, scalenet.py:176-180) owns ONE slot."""
    seen, out = set(), []
    for sp in specs:
        if sp[0].auto_name not in seen:
            seen.add(sp[0].auto_name)
            out.append(sp)
    return out


def get_store(net, rt, layers):
    """The device parameter store of `net` (created on first use, rebuilt when the parameter list changed).  A net built as the
    `twin` of another one (copyLayer = twin.layers[i] for every layer: the same SharedParam objects) lives in THAT net's store --
    one copy of the weights, whichever instance trains or evaluates.  (Each compiled engine owns the gradient buffer during its
    backward pass: gradients of two twins trained in one step are not summed.)"""
    specs = _dedupe_specs(_param_specs(net, layers))
    names = [p.auto_name for (p, _, _, _) in specs]
    owner = net
    while getattr(owner, '_twin', None) is not None:
        owner = owner._twin
    store = getattr(owner, '_param_store', None)
    same_list = store is not None and [s['param'].auto_name for s in store.slots] == names and \
        [s['info'] for s in store.slots] == [i for (_, _, i, _) in specs]
    if same_list and store.rt is rt:
        net._param_store = store
        return store
    if store is not None:
        if owner is not net and not same_list:
            raise RuntimeError("twin net: its parameter list differs from the one of the net it shares its weights with (%d vs %d "
                               "parameters); the shared store is not rebuilt under the owner's engines" % (len(names), len(store.slots)))
        # same parameters on ANOTHER runtime (a twin, or the net itself, compiled on a different device / stream set): the store moves
        # there -- values are pulled back to the host copies first, engines of the old runtime refuse to run (check_live)
        store.release()
    store = ParamStore(rt, specs)
    owner._param_store = store
    net._param_store = store
    return store


class CompiledNet(object):
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
            return int(K >= 32)
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
            bm = hz.conv3x3_bm(M, Co)
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
                self.bwd.add(ops.wgrad_stream(rt, dy, Co, src.base.buf, Ci, M, rpw, part, mapX=mp, actX=act,
                                              name='wgrad1x1_%d' % layer.layerNum), side=True)
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
                bmd = hz.conv3x3_bm(N * Hi * Wi, Ci)
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
