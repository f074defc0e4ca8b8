"""
hipdp.store -- the device tensors, views and the flat parameter store the engine works on (split out of hipdp/engine.py in round 6).

Activations are NHWC (`TensorV`); BatchNorm / ReLU layers are never materialised: they become `View`s (base tensor + pending BatchNorm +
pending ReLU) that the consuming conv / FC applies while staging its operand.  Parameters live in ONE flat device buffer in kernel
layout (`ParamStore`: so ADAM and a data-parallel gradient all-reduce are single operations), with get_value() / set_value() converting to
the reference's layouts on the fly (/root/reference/src/net/netbase.py:318-346 for what a checkpoint holds).
"""
import numpy as np

from . import layout, ops
from . import heuristics as hz

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
