"""Test helper: read the ReLU / max-pool decisions the DEVICE made in a training forward pass out of a CompiledNet, in the
form oracle.nets.forward / oracle.torch_ref.forward accept as `masks` (NCHW, indexed by the oracle's layer numbers, which are
the reference's layerNum).  With the decisions pinned, the device's float32 gradients and the float64 oracle's differ by
arithmetic only, so they can be compared at float32 round-off for any batch size (see oracle/torch_ref.forward)."""
import numpy as np


def _layer_index(net):
    return {id(l): i for i, l in enumerate(net.layers)}


def _f32(t):
    """A device tensor's values as float32, whatever its storage (bf16-stored activation tensors widen exactly)."""
    return t.get_f32() if hasattr(t, 'get_f32') else t.buf.get()


def device_masks(eng, net):
    idx = _layer_index(net)
    masks = {}
    for var in eng.order:
        if var.kind == 'layer' and var.layer.__class__.__name__ == 'NonlinearityLayer':
            v = eng._memo[id(var)]
            if not v.relu:
                continue
            x = _f32(v.base)                                      # NHWC (or [N][D]) pre-activation tensor
            if v.bn is not None:
                C = v.bn.C
                mean, scale, beta = (b.get()[:C] for b in (v.bn.mean, v.bn.scale, v.bn.beta_buf))
                dx = (x.astype(np.float32) - mean.astype(np.float32)).astype(np.float32)        # the kernel's first rounding
                val = dx.astype(np.float64) * scale.astype(np.float64) + beta.astype(np.float64)   # fma: one more rounding, sign exact
            else:
                val = x
            m = val >= 0
            masks[idx[id(var.layer)]] = np.ascontiguousarray(np.moveaxis(m, -1, 1)) if m.ndim == 4 else m
        elif var.kind == 'relu':
            # activation wrapped around a layer's own output (HiddenLayer / ConvPoolLayer with ReLU): the layer's materialised
            # output is the pre-activation
            lay = var.inputs[0].layer
            io = eng.layer_io[id(lay)]
            pre = _f32(io['out'])
            m = pre >= 0
            if lay.__class__.__name__ == 'HiddenLayer':
                masks[idx[id(lay)]] = m
            elif lay.__class__.__name__ == 'ConvPoolLayer':
                # conv -> pool -> bias -> ReLU: the ReLU decision on the pooled, biased map (key ('relu', layer))
                masks[('relu', idx[id(lay)])] = np.ascontiguousarray(np.moveaxis(m, -1, 1))
    for lay in net.layers:
        io = eng.layer_io.get(id(lay))
        if io is not None and io.get('stem') and io.get('argmax') is not None:
            bits = io['argmax'].get()                              # [N][Hp][Wp][Co] uint8, bit j = window element j (row-major)
            t = np.stack([(bits >> j) & 1 for j in range(4)], axis=-1).astype(bool)      # [N][Hp][Wp][Co][4]
            masks[idx[id(lay)]] = np.ascontiguousarray(np.moveaxis(t, 3, 1))              # [N][Co][Hp][Wp][4]
        elif io is not None and io.get('ties') is not None:
            g = io['geom']
            pp = g['pool'] * g['pool']
            bits = io['ties'].get().astype(np.uint32)              # [N][Hp][Wp][Co] uint16, bit j = window element j (row-major)
            t = np.stack([(bits >> j) & 1 for j in range(pp)], axis=-1).astype(bool)
            masks[idx[id(lay)]] = np.ascontiguousarray(np.moveaxis(t, 3, 1))
    return masks


# ---- bf16 operands (BASELINE config 5) ------------------------------------------------------------------------------------------
def _activated_operand(view):
    """The float32 value the kernels form from a (base tensor, pending BatchNorm, pending ReLU) view while they stage it, NCHW (or
    [N][D] for a flattened view): (x - mean) in float32, then ONE fused multiply-add with scale / beta, then the ReLU -- the
    arithmetic of dpp_act4 (csrc/dpp_common.h), reproduced with exact float64 products rounded once."""
    x = _f32(view.base).astype(np.float32)
    if view.bn is not None:
        C = view.bn.C
        mean, scale, beta = (b.get()[:C].astype(np.float32) for b in (view.bn.mean, view.bn.scale, view.bn.beta_buf))
        dx = (x - mean).astype(np.float32)
        x = (dx.astype(np.float64) * scale.astype(np.float64) + beta.astype(np.float64)).astype(np.float32)
    if view.relu:
        x = np.maximum(x, np.float32(0))
    if x.ndim == 4:
        x = np.moveaxis(x, -1, 1)                          # NHWC -> NCHW
        if len(view.shape) == 2:
            x = x.reshape(x.shape[0], -1)                  # the oracle flattens NCHW (its FC rows are in that order)
    return np.ascontiguousarray(x)


def device_quant(eng, net):
    """Which conv / FC layers of the compiled net run on bf16 MFMA operands, read off its launches, in the form
    oracle.torch_ref.forward takes as `quant`: {layer index: dict(fwd=, dgrad=, wgrad=, pin=)}.  `pin` is the device's own rounded
    forward operand of the layer (bf16 values as float32), rebuilt from the tensors the engine holds; `pin_dy` the rounded gradient
    w.r.t. its output, the operand of its backward products."""
    from oracle import layers as L
    by_num = {l.layerNum: (i, l) for i, l in enumerate(net.layers)}
    roles = {'conv3x3_': 'fwd', 'conv1x1_': 'fwd', 'fc_': 'fwd', 'dgrad3x3_': 'dgrad', 'dgrad1x1_': 'dgrad', 'fc_dgrad_': 'dgrad',
             'wgrad1x1_': 'wgrad', 'wgrad3x3_': 'wgrad', 'fc_wgrad_': 'wgrad'}
    quant = {}
    launches = eng.all_launches() if getattr(eng, 'train', False) else [('fwd', o) for o in eng.fwd.launches()]
    for _, l in launches:
        if 'bf16' not in (l.meta or {}).get('kernel', ''):
            continue
        for prefix in sorted(roles, key=len, reverse=True):
            if l.name.startswith(prefix) and l.name[len(prefix):].isdigit():
                i, layer = by_num[int(l.name[len(prefix):])]
                quant.setdefault(i, dict(fwd=False, dgrad=False, wgrad=False, pin=None, pin_dy=None))[roles[prefix]] = True
                break
        else:
            raise AssertionError("a bf16 launch the oracle cannot place: %r" % (l.name,))
    for i, q in quant.items():
        io = eng.layer_io[id(net.layers[i])]
        if q['fwd'] or q['wgrad']:
            q['pin'] = L.bf16_round(_activated_operand(io['in_view']))
        if (q['dgrad'] or q['wgrad']) and io['out'].grad is not None and getattr(eng, 'train', False):
            # the gradient w.r.t. the layer's output as the backward kernels read it (call after the backward pass has run)
            g = io['out'].grad.get()
            if g.dtype == np.uint16:                                # a bf16-stored gradient tensor: its bits, widened
                g = (g.astype(np.uint32) << 16).view(np.float32)
            g = g.astype(np.float32)
            q['pin_dy'] = L.bf16_round(np.ascontiguousarray(np.moveaxis(g, -1, 1)) if g.ndim == 4 else g)
    return quant


# ---- bf16 STORAGE (BASELINE config 5, ABI v9) -----------------------------------------------------------------------------------
def device_store(eng, net, pin=True):
    """The layers whose materialised output the compiled net holds as bfloat16, in the form oracle.torch_ref.forward takes as `store`:
    {layer index: the device's stored tensor as float32 NCHW} (pin=False: {layer index: None}, the oracle rounds by itself)."""
    store = {}
    for i, l in enumerate(net.layers):
        io = eng.layer_io.get(id(l))
        if io is None or not getattr(io.get('out'), 'is16', False):
            continue
        store[i] = np.ascontiguousarray(np.moveaxis(io['out'].get_f32(), -1, 1)) if pin else None
    return store


def store_agreement(store, stored_out):
    """Per layer, the fraction of elements where the oracle's OWN rounding of the tensor it would have stored equals the device's
    stored tensor (the pin): the un-pinned check of the pins themselves.  A float32 and a float64 evaluation of the same value land on
    different bfloat16 neighbours only when it sits within float32 round-off of a rounding boundary."""
    out = {}
    for i, p in store.items():
        if p is None:
            continue
        p, mine = np.asarray(p, np.float32), np.asarray(stored_out[i], np.float32)
        # ... or when the element is so small against its tensor that float32 accumulation noise (a few 1e-7 of the terms' magnitude)
        # reaches its bfloat16 spacing: a channel whose outputs nearly cancel (the stage-1 projection shortcut of the 256x256 net has
        # one: |y| ~ 1e-3 against a tensor of O(1), 59 % of it lands on the neighbouring bfloat16).  Those count as agreeing; a wrong
        # rounding rule or a wrong operand shows up at the tensor's own magnitude, a thousand times this bound.
        noise = np.float32(2e-6) * np.abs(p).max()
        out[i] = float(((mine == p) | (np.abs(mine - p) <= noise)).mean())
    return out


def device_grad_pins(eng, net):
    """The bf16-stored GRADIENT tensors of the compiled net after a backward pass, for oracle.torch_ref.forward(grad_pins=): (G, dV) with
    G {index of a BatchNorm layer: its masked gradient as the backward kernels read it} and dV {index of a stored layer: the gradient
    of its tensor}, float32 NCHW.  Only the tensors the device really holds as bfloat16 are pinned."""
    from hipdp.engine import bf16_bits_to_f32
    from hipdp import ops

    def nchw(buf, shape):
        return np.ascontiguousarray(np.moveaxis(bf16_bits_to_f32(buf.get()).reshape(shape), -1, 1))
    G, dV = {}, {}
    for i, l in enumerate(net.layers):
        if l.__class__.__name__ == 'BatchNormLayer':
            g = getattr(eng, 'bn_view_grad', {}).get(id(l))
            b = eng.bn_states.get(id(l))
            if g is not None and g.dtype == ops.BF16:
                G[i] = nchw(g, g.shape)
        io = eng.layer_io.get(id(l))
        t = io.get('out') if io is not None else None
        if t is not None and t.grad is not None and t.grad.dtype == ops.BF16:
            dV[i] = nchw(t.grad, t.shape)
    return G, dV
