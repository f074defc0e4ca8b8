"""Test helper: read the ReLU / max-pool decisions the DEVICE made in a training forward pass out of a CompiledNet, in the
form oracle.nets.forward / oracle.torch_ref.forward accept as `masks` (NCHW, indexed by the oracle's layer numbers, which are
the reference's layerNum).  With the decisions pinned, the device's float32 gradients and the float64 oracle's differ by
arithmetic only, so they can be compared at float32 round-off for any batch size (see oracle/torch_ref.forward)."""
import numpy as np


def _layer_index(net):
    return {id(l): i for i, l in enumerate(net.layers)}


def device_masks(eng, net):
    idx = _layer_index(net)
    masks = {}
    for var in eng.order:
        if var.kind == 'layer' and var.layer.__class__.__name__ == 'NonlinearityLayer':
            v = eng._memo[id(var)]
            if not v.relu:
                continue
            x = v.base.buf.get()                                  # NHWC (or [N][D]) pre-activation tensor
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
            pre = io['out'].buf.get()
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
