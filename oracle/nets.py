"""
oracle/nets.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restatement of the reference's network graphs and of one `train_model` call.

  build_resnet      /root/reference/src/net/resnet.py:92-195, 342-414   (types 0, 1)
  build_poseregnet  /root/reference/src/net/poseregnet.py:44-145        (types 0, 11)
  init_params       /root/reference/src/net/layer.py:70-124 (getInitVals) + the layer ctors
  forward/backward  the Theano graph + T.grad (poseregnettrainer.py:84-111)
  train_step        poseregnettrainer.py:146-159 + optimizer.py:58-90 + batchnormlayer.py:164-172
  compute_output    /root/reference/src/net/netbase.py:217-316

A net is a list of layer dicts in the reference's `layers` order (so `layerNum`
== list index == the '<layerNum>-values' checkpoint key).  Each layer has a
'src' value reference:  ('input',) | ('layer', i) | ('add', refA, refB) |
('flatten', ref).
"""
import numpy as np
from . import layers as L


# --------------------------------------------------------------------------- graph builders
def _conv_out(in_dim, nf, k, stride, border):
    oh, ow = L._out_hw(in_dim[2], in_dim[3], k[0], k[1], stride, border)
    return (in_dim[0], nf, oh, ow)


def _res_block(layers, src, in_dim, out_filters, stride):
    """res_block, /root/reference/src/net/resnet.py:349-414.  Returns (out_ref, out_dim)."""
    nb = out_filters // 4

    def bn(s, d):
        layers.append(dict(kind='bn', src=s, in_dim=d, out_dim=d))
        return ('layer', len(layers) - 1)

    def relu(s, d):
        layers.append(dict(kind='relu', src=s, in_dim=d, out_dim=d))
        return ('layer', len(layers) - 1)

    def conv(s, d, nf, k, st):
        od = _conv_out(d, nf, k, st, 'half')
        layers.append(dict(kind='conv', src=s, in_dim=d, out_dim=od, nf=nf, k=k, stride=st, border='half'))
        return ('layer', len(layers) - 1), od

    if in_dim[1] == out_filters:
        # identity block; the stride argument is ignored (resnet.py:353-379)
        r = relu(bn(src, in_dim), in_dim)
        c1, d1 = conv(r, in_dim, nb, (1, 1), (1, 1))
        r = relu(bn(c1, d1), d1)
        c2, d2 = conv(r, d1, nb, (3, 3), (1, 1))
        r = relu(bn(c2, d2), d2)
        c3, d3 = conv(r, d2, out_filters, (1, 1), (1, 1))
        return ('add', src, c3), d3
    # projection block (resnet.py:380-414): common BN+ReLU feeds main path and shortcut
    h = relu(bn(src, in_dim), in_dim)
    c1, d1 = conv(h, in_dim, nb, (1, 1), (stride, stride))
    r = relu(bn(c1, d1), d1)
    c2, d2 = conv(r, d1, nb, (3, 3), (1, 1))
    r = relu(bn(c2, d2), d2)
    c3, d3 = conv(r, d2, out_filters, (1, 1), (1, 1))
    sc, dsc = conv(h, in_dim, out_filters, (1, 1), (stride, stride))   # layers[-8] = h
    assert dsc == d3
    return ('add', c3, sc), dsc


def build_resnet(type=0, nChan=1, wIn=128, hIn=128, batchSize=128, numJoints=16, nDims=3):
    """ResNet types 0-4, /root/reference/src/net/resnet.py:120-336.  n = (47-2)/9 = 5 (py2 int div).  Types 2-4 put a
    DropoutLayer behind each 1024-wide layer (:221, :231); type 3 narrows stages 3-4 to 128 filters (:249), which makes
    stage 3 a run of identity blocks without down-sampling; types 1 and 4 add the 30-D bottleneck."""
    if type not in (0, 1, 2, 3, 4):
        raise NotImplementedError("ResNet type %r" % (type,))
    n = (47 - 2) // 9
    st = [32, 64, 128, 128, 128] if type == 3 else [32, 64, 128, 256, 256]
    dropout, bottleneck = type in (2, 3, 4), type in (1, 4)
    layers = []
    in_dim = (batchSize, nChan, hIn, wIn)
    cd = _conv_out(in_dim, st[0], (5, 5), (1, 1), 'half')
    od = (cd[0], cd[1], cd[2] // 2, cd[3] // 2)
    layers.append(dict(kind='convpool', src=('input',), in_dim=in_dim, out_dim=od, nf=st[0], k=(5, 5),
                       stride=(1, 1), border='half', pool=(2, 2), act=None))
    ref, dim = ('layer', 0), od
    for s in range(1, 5):
        ref, dim = _res_block(layers, ref, dim, st[s], 2)
        for _ in range(1, n):
            ref, dim = _res_block(layers, ref, dim, st[s], 1)
    layers.append(dict(kind='bn', src=ref, in_dim=dim, out_dim=dim))
    layers.append(dict(kind='relu', src=('layer', len(layers) - 1), in_dim=dim, out_dim=dim))
    flat = (dim[0], int(np.prod(dim[1:])))
    widths = [1024, 1024] + ([30] if bottleneck else []) + [numJoints * nDims]
    acts = ['relu', 'relu'] + ([None] if bottleneck else []) + [None]
    src, d = ('flatten', ('layer', len(layers) - 1)), flat
    for w, a in zip(widths, acts):
        layers.append(dict(kind='fc', src=src, in_dim=d, out_dim=(batchSize, w), act=a))
        src, d = ('layer', len(layers) - 1), (batchSize, w)
        if dropout and w == 1024:
            layers.append(dict(kind='dropout', src=src, in_dim=d, out_dim=d, p=0.3))
            src = ('layer', len(layers) - 1)
    return dict(layers=layers, out=('layer', len(layers) - 1), batch_size=batchSize,
                in_dim=in_dim, out_dim=(batchSize, numJoints * nDims), name='ResNet')


def build_poseregnet(type=0, nChan=1, wIn=128, hIn=128, batchSize=128, numJoints=16, nDims=3):
    """PoseRegNet types 0/11, /root/reference/src/net/poseregnet.py:60-143."""
    if type not in (0, 11):
        raise NotImplementedError("not implemented")
    layers = []
    d = (batchSize, nChan, hIn, wIn)
    src = ('input',)
    for k, pool in (((5, 5), (4, 4)), ((5, 5), (2, 2)), ((3, 3), (1, 1))):
        cd = _conv_out(d, 8, k, (1, 1), 'valid')
        od = (cd[0], cd[1], cd[2] // pool[0], cd[3] // pool[1])
        layers.append(dict(kind='convpool', src=src, in_dim=d, out_dim=od, nf=8, k=k, stride=(1, 1),
                           border='valid', pool=pool, act='relu'))
        src, d = ('layer', len(layers) - 1), od
    src, d = ('flatten', src), (d[0], d[1] * d[2] * d[3])
    for _ in range(2):
        layers.append(dict(kind='fc', src=src, in_dim=d, out_dim=(batchSize, 1024), act='relu'))
        src, d = ('layer', len(layers) - 1), (batchSize, 1024)
        layers.append(dict(kind='dropout', src=src, in_dim=d, out_dim=d, p=L.DROPOUT_P))
        src = ('layer', len(layers) - 1)
    if type == 11:
        layers.append(dict(kind='fc', src=src, in_dim=d, out_dim=(batchSize, 30), act=None))
        src, d = ('layer', len(layers) - 1), (batchSize, 30)
    layers.append(dict(kind='fc', src=src, in_dim=d, out_dim=(batchSize, numJoints * nDims), act=None))
    return dict(layers=layers, out=('layer', len(layers) - 1), batch_size=batchSize,
                in_dim=(batchSize, nChan, hIn, wIn), out_dim=(batchSize, numJoints * nDims), name='PoseRegNet')


def build_scalenet(type=1, nChan=1, wIn=128, hIn=128, batchSize=128, numJoints=1, nDims=3, resizeFactor=2, shared_conv=False):
    """ScaleNet type 1, /root/reference/src/net/scalenet.py:49-127, 150-180: three conv-pool towers on the crop and its
    1/2 and 1/4 centre crops, flattened and concatenated, then FC 1024 - dropout - FC 1024 - dropout - FC out."""
    if type != 1:
        raise NotImplementedError("not implemented")
    layers, tails = [], []
    towers = (((5, 5), (4, 4)), ((5, 5), (2, 2)), ((3, 3), (1, 1))), (((5, 5), (2, 2)), ((5, 5), (2, 2)), ((3, 3), (1, 1))), \
        (((5, 5), (2, 2)), ((5, 5), (1, 1)), ((3, 3), (1, 1)))
    in_dims = []
    for t, tower in enumerate(towers):
        d = (batchSize, nChan, hIn // resizeFactor ** t, wIn // resizeFactor ** t)
        in_dims.append(d)
        src = ('input', t)
        for k, pool in tower:
            cd = _conv_out(d, 8, k, (1, 1), 'valid')
            od = (cd[0], cd[1], cd[2] // pool[0], cd[3] // pool[1])
            layers.append(dict(kind='convpool', src=src, in_dim=d, out_dim=od, nf=8, k=k, stride=(1, 1), border='valid', pool=pool,
                               act='relu'))
            if shared_conv and t > 0:
                # scalenet.py:176-180: towers 2 and 3 are built with copyLayer = the first tower's layer: the SAME W and b
                layers[-1]['share'] = (len(layers) - 1) % len(tower)
            src, d = ('layer', len(layers) - 1), od
        tails.append((src, d[1] * d[2] * d[3]))
    src = ('concat',) + tuple(('flatten', r) for r, _ in tails)
    d = (batchSize, sum(n for _, n in tails))
    for _ in range(2):
        layers.append(dict(kind='fc', src=src, in_dim=d, out_dim=(batchSize, 1024), act='relu'))
        src, d = ('layer', len(layers) - 1), (batchSize, 1024)
        layers.append(dict(kind='dropout', src=src, in_dim=d, out_dim=d, p=L.DROPOUT_P))
        src = ('layer', len(layers) - 1)
    layers.append(dict(kind='fc', src=src, in_dim=d, out_dim=(batchSize, numJoints * nDims), act=None))
    return dict(layers=layers, out=('layer', len(layers) - 1), batch_size=batchSize, in_dim=in_dims,
                out_dim=(batchSize, numJoints * nDims), name='ScaleNet')


def scalenet_inputs(x):
    """The three inputs of ScaleNet from the full crop: itself and its 1/2 and 1/4 CENTRE crops (no resampling),
    /root/reference/src/trainer/scalenettrainer.py:239-251, /root/reference/src/util/handdetector.py:654-666."""
    H, W = x.shape[2], x.shape[3]
    out = [x]
    for f in (2, 4):
        h, w = H // f, W // f
        xs, ys = int(H / 2 - h / 2), int(W / 2 - w / 2)
        out.append(np.ascontiguousarray(x[:, :, ys:ys + w, xs:xs + h]))
    return out


def has_dropout(net):
    return any(l['kind'] == 'dropout' for l in net['layers'])


# --------------------------------------------------------------------------- parameters
def init_params(net, rng, dtype=np.float32):
    """
    Draws from `rng` in layer-construction order exactly as the layer ctors do:
      conv / convpool: He normal, std = sqrt(2 / (C*kh*kw))   (layer.py:82-86; init_method='He' for the
                       ResNet, activation ReLU -> 'He' for PoseRegNet conv-pools)
      fc + ReLU:       normal std 0.01                         (layer.py:87-88)
      fc linear:       uniform +-sqrt(6/(n_in+n_out))          (layer.py:111-118)
      dropout:         consumes rng.randint(999999)            (dropoutlayer.py:98)
      bn:              beta 0, gamma 1, mean 0, inv_std 1      (batchnormlayer.py:133-142)
    Returns {layerNum: [arrays in the reference's params + params_nontrained order]}.
    """
    P = {}
    for i, l in enumerate(net['layers']):
        k = l['kind']
        if 'share' in l:
            P[i] = P[l['share']]                     # copyLayer: no initialiser call, the rng is not consumed
            continue
        if k in ('conv', 'convpool'):
            shape = (l['nf'], l['in_dim'][1], l['k'][0], l['k'][1])
            bound = np.sqrt(2. / np.prod(shape[1:]))
            P[i] = [np.asarray(rng.normal(loc=0.0, scale=bound, size=shape), dtype=dtype),
                    np.zeros((shape[0],), dtype=dtype)]
        elif k == 'fc':
            shape = (l['in_dim'][1], l['out_dim'][1])
            if l['act'] == 'relu':
                W = np.asarray(rng.normal(loc=0.0, scale=0.01, size=shape), dtype=dtype)
            else:
                b = np.sqrt(6. / np.sum(shape))
                W = np.asarray(rng.uniform(low=-b, high=b, size=shape), dtype=dtype)
            P[i] = [W, np.zeros((shape[1],), dtype=dtype)]
        elif k == 'bn':
            C = l['in_dim'][1]
            P[i] = [np.zeros(C, dtype), np.ones(C, dtype), np.zeros(C, dtype), np.ones(C, dtype)]
        elif k == 'dropout':
            rng.randint(999999)
    return P


def perturb_bn(P, net, rng, dtype=None):
    """Move BN parameters / running statistics away from identity so the BN path is exercised
    ('seeded synthetic weights')."""
    for i, l in enumerate(net['layers']):
        if l['kind'] == 'bn':
            C = l['in_dim'][1]
            dt = dtype or P[i][0].dtype
            P[i] = [np.asarray(rng.normal(0, 0.2, C), dt), np.asarray(rng.uniform(0.6, 1.4, C), dt),
                    np.asarray(rng.normal(0, 0.3, C), dt), np.asarray(rng.uniform(0.5, 2.0, C), dt)]
        elif l['kind'] in ('conv', 'convpool', 'fc'):
            dt = dtype or P[i][1].dtype
            P[i][1] = np.asarray(rng.normal(0, 0.05, P[i][1].shape), dt)
    return P


def cast_params(P, dtype):
    out, seen = {}, {}
    for i, v in P.items():
        if id(v) not in seen:
            seen[id(v)] = [np.asarray(a, dtype) for a in v]
        out[i] = seen[id(v)]                         # layers that share parameters keep sharing them
    return out


def trained_param_list(net, P):
    """The (layer, slot) pairs of net.params: conv/fc [W, b], bn [beta, gamma] (batchnormlayer.py:146-151)."""
    out = []
    for i, l in enumerate(net['layers']):
        if l['kind'] in ('conv', 'convpool', 'fc', 'bn') and 'share' not in l:      # a shared parameter is trained once
            out += [(i, 0), (i, 1)]
    return out


# --------------------------------------------------------------------------- forward / backward
def fused_convs(net):
    """'conv' layers whose output the device never holds on its own: the engine fuses `a + conv(...)` into the epilogue of the conv at
    the end of the LONGER branch of the sum (ties: the later one; hipdp/engine.py:_emit_add), so the tensor it materialises -- and, in
    the bf16 storage mode, rounds -- is the sum."""
    fused = set()
    depth = {}

    def ref_depth(ref):
        if ref[0] == 'layer':
            return depth[ref[1]]
        return max([ref_depth(r) for r in ref[1:] if isinstance(r, tuple)] or [0])

    for i, l in enumerate(net['layers']):
        depth[i] = 1 + ref_depth(l['src'])

    def walk(ref):
        if ref[0] == 'add':
            cand = [r[1] for r in ref[1:] if r[0] == 'layer' and net['layers'][r[1]]['kind'] == 'conv']
            if cand:
                fused.add(max(cand, key=lambda i: (depth[i], i)))
        for r in ref[1:]:
            if isinstance(r, tuple):
                walk(r)

    for l in net['layers']:
        walk(l['src'])
    walk(net['out'])
    return fused


def forward(net, P, x, train, dropout_masks=None, masks=None, bf16=None, store16=False):
    """Returns (output, cache).  train=True <=> unsetDeterministic (BN batch statistics, dropout masks).
    masks (optional): {layer index: bool array} pins the pass / block decision of a 'relu' layer or of an 'fc' layer's ReLU
    to the given pattern (the device's own, when comparing gradients: see oracle/torch_ref.forward).
    bf16 (optional): the set of 'conv' / 'fc' layer indices whose two MFMA operands (activated input, weight) are rounded to
    bfloat16 before the product, accumulation and everything else staying in the array dtype -- the arithmetic of the bf16 kernels
    (BASELINE config 5).  Forward only: backward() differentiates the unrounded graph; the gradient-parity tests use
    oracle/torch_ref.py, which models the rounded backward operands as well.
    store16: the bf16 STORAGE mode of config 5 -- every tensor a convolution materialises (the stem's pooled map, a conv's output, or
    the residual sum a conv's epilogue forms) is rounded to bfloat16 when it is written; the (training-mode) BatchNorm that reads it
    takes its statistics from the UNROUNDED values and normalises the rounded ones, as the kernels do."""
    vals, cache, memo, raw = {}, {}, {}, {}
    masks = masks or {}
    bf16 = bf16 or ()
    fused = fused_convs(net) if store16 else set()

    def get(ref):
        if ref[0] == 'input':
            return x[ref[1]] if len(ref) > 1 else x        # multi-input nets (ScaleNet) pass a list of arrays
        if ref[0] == 'layer':
            return vals[ref[1]]
        if ref[0] == 'add':
            if ref not in memo:
                memo[ref] = get(ref[1]) + get(ref[2])
                if store16:
                    raw[ref] = memo[ref]
                    memo[ref] = L.bf16_round(memo[ref])
            return memo[ref]
        if ref[0] == 'flatten':
            v = get(ref[1])
            return v.reshape(v.shape[0], -1)
        if ref[0] == 'concat':                              # T.concatenate(..., axis=1), scalenet.py:167-171
            return np.concatenate([get(r) for r in ref[1:]], axis=1)
        raise ValueError(ref)

    def stored(i):
        """Layer i's output as the device holds it (store16: rounded, unless it only exists inside a later residual sum)."""
        if store16 and i not in fused:
            raw[('layer', i)] = vals[i]
            vals[i] = L.bf16_round(vals[i])

    for i, l in enumerate(net['layers']):
        a = get(l['src'])
        k = l['kind']
        if k == 'convpool':
            vals[i], cache[i] = L.convpool_fwd(a, P[i][0], P[i][1], l['stride'], l['border'], l['pool'],
                                               l['act'] == 'relu')
            cache[i] = (a,) + cache[i]
            stored(i)
        elif k == 'conv':
            if i in bf16:
                vals[i] = L.conv2d_fwd(L.bf16_round(a), L.bf16_round(P[i][0]), P[i][1], l['stride'], l['border'])
            else:
                vals[i] = L.conv2d_fwd(a, P[i][0], P[i][1], l['stride'], l['border'])
            cache[i] = (a,)
            stored(i)
        elif k == 'bn':
            beta, gamma, rm, ris = P[i]
            if train and store16 and l['src'] in raw:
                # statistics of the values the epilogue formed, normalisation of the tensor it stored
                _, mean, inv_std = L.bn_fwd_train(raw[l['src']], gamma, beta)
                vals[i] = L.bn_fwd_eval(a, gamma, beta, mean, inv_std)
                cache[i] = (a, mean, inv_std)
            elif train:
                vals[i], mean, inv_std = L.bn_fwd_train(a, gamma, beta)
                cache[i] = (a, mean, inv_std)
            else:
                vals[i] = L.bn_fwd_eval(a, gamma, beta, rm, ris)
        elif k == 'relu':
            vals[i] = L.relu_fwd(a) if i not in masks else a * masks[i]
            cache[i] = (a,) if i not in masks else (a, masks[i])
        elif k == 'fc':
            pre = L.fc_fwd(L.bf16_round(a), L.bf16_round(P[i][0]), P[i][1]) if i in bf16 else L.fc_fwd(a, P[i][0], P[i][1])
            if l['act'] == 'relu':
                vals[i] = L.relu_fwd(pre) if i not in masks else pre * masks[i]
            else:
                vals[i] = pre
            cache[i] = (a, pre) if i not in masks else (a, pre, masks[i])
        elif k == 'dropout':
            if train:
                m = dropout_masks[i]
                vals[i] = L.dropout_fwd_train(a, m)
                cache[i] = (m,)
            else:
                vals[i] = L.dropout_fwd_eval(a, l['p'])
        else:
            raise NotImplementedError(k)
    cache['vals'] = vals
    return get(net['out']), cache


def backward(net, P, cache, dout, weight_decay=0.0):
    """Reverse-mode gradients of all trained parameters.  Returns {layerNum: [g_slot0, g_slot1]}."""
    vals = cache['vals']
    gl = {}
    shapes = {}

    def shape_of(ref):
        if ref[0] == 'layer':
            return vals[ref[1]].shape
        if ref[0] == 'add':
            return shape_of(ref[1])
        if ref[0] == 'flatten':
            s = shape_of(ref[1])
            return (s[0], int(np.prod(s[1:])))
        if ref[0] == 'concat':
            ss = [shape_of(r) for r in ref[1:]]
            return (ss[0][0], sum(q[1] for q in ss))
        return None

    def push(ref, g):
        if ref[0] == 'input':
            return
        if ref[0] == 'layer':
            gl[ref[1]] = g if ref[1] not in gl else gl[ref[1]] + g
        elif ref[0] == 'add':
            push(ref[1], g)
            push(ref[2], g)
        elif ref[0] == 'flatten':
            push(ref[1], g.reshape(shape_of(ref[1])))
        elif ref[0] == 'concat':
            o = 0
            for r in ref[1:]:
                n = shape_of(r)[1]
                push(r, g[:, o:o + n])
                o += n

    push(net['out'], dout)
    G = {}
    for i in reversed(range(len(net['layers']))):
        l = net['layers'][i]
        if i not in gl:
            continue
        dy = gl.pop(i)
        k = l['kind']
        need_dx = l['src'][0] != 'input'
        if k == 'convpool':
            a = cache[i][0]
            dx, dW, db = L.convpool_bwd(a, P[i][0], dy, cache[i][1:], l['stride'], l['border'], l['pool'],
                                        l['act'] == 'relu', need_dx=need_dx)
            G[i] = [dW + 2 * weight_decay * P[i][0] if weight_decay else dW, db]
        elif k == 'conv':
            a = cache[i][0]
            dx, dW, db = L.conv2d_bwd(a, P[i][0], dy, l['stride'], l['border'], need_dx=need_dx)
            G[i] = [dW + 2 * weight_decay * P[i][0] if weight_decay else dW, db]
        elif k == 'bn':
            a, mean, inv_std = cache[i]
            dx, dgamma, dbeta = L.bn_bwd_train(a, P[i][1], mean, inv_std, dy)
            G[i] = [dbeta, dgamma]
        elif k == 'relu':
            dx = L.relu_bwd(cache[i][0], dy) if len(cache[i]) == 1 else dy * cache[i][1]
        elif k == 'fc':
            a, pre = cache[i][:2]
            if l['act'] == 'relu':
                g = L.relu_bwd(pre, dy) if len(cache[i]) == 2 else dy * cache[i][2]
            else:
                g = dy
            dx, dW, db = L.fc_bwd(a, P[i][0], g)
            G[i] = [dW + 2 * weight_decay * P[i][0] if weight_decay else dW, db]
        elif k == 'dropout':
            dx = dy * cache[i][0]
        else:
            raise NotImplementedError(k)
        if need_dx:
            push(l['src'], dx)
    for i, l in enumerate(net['layers']):            # shared parameters: the gradient is the sum over the layers that use them
        if 'share' in l and i in G:
            src = l['share']
            G[src] = [a + b for a, b in zip(G[src], G.pop(i))]
    return G


def cost_and_grads(net, P, x, y, train=True, dropout_masks=None, weight_decay=0.0, joints=None, masks=None):
    """The training cost (poseregnettrainer.py:92-107) and its gradients; `masks`: see forward()."""
    out, cache = forward(net, P, x, train, dropout_masks, masks)
    if joints is None:
        cost, dout = L.loss_embedding(out, y)
    else:
        cost, dout = L.loss_joints(out, y, *joints)
    wd = weight_decay if not has_dropout(net) else 0.0       # poseregnettrainer.py:106-107
    if wd:
        for i, l in enumerate(net['layers']):
            if l['kind'] in ('conv', 'convpool', 'fc'):
                cost = cost + wd * (P[i][0] ** 2).sum()
    return cost, backward(net, P, cache, dout, wd), cache, out


def new_adam_state(net, P):
    plist = trained_param_list(net, P)
    return dict(t=1.0, m={pl: np.zeros_like(P[pl[0]][pl[1]]) for pl in plist},
                v={pl: np.zeros_like(P[pl[0]][pl[1]]) for pl in plist})


def train_step(net, P, state, x, y, lr, dropout_masks=None, weight_decay=0.0):
    """One `train_model(index, lr)` call: cost, T.grad, ADAM updates and the BN running-statistics
    default_updates, all from pre-step values (theano.function(updates=...))."""
    cost, G, cache, _ = cost_and_grads(net, P, x, y, True, dropout_masks, weight_decay)
    plist = trained_param_list(net, P)
    params = [P[i][s] for i, s in plist]
    grads = [G[i][s] for i, s in plist]
    m = [state['m'][pl] for pl in plist]
    v = [state['v'][pl] for pl in plist]
    state['t'] = L.adam_step(params, grads, m, v, state['t'], lr)
    for j, pl in enumerate(plist):
        P[pl[0]][pl[1]] = params[j]
        state['m'][pl], state['v'][pl] = m[j], v[j]
    for i, l in enumerate(net['layers']):
        if l['kind'] == 'bn':
            _, mean, inv_std = cache[i]
            P[i][2], P[i][3] = L.bn_running_update(P[i][2], P[i][3], mean, inv_std)
    return cost, G


def compute_output(net, P, inputs, bf16=None, store16=False):
    """NetBase.computeOutput, /root/reference/src/net/netbase.py:217-316: deterministic forward in
    batches of batch_size, the last batch padded by repeating the last sample, result trimmed.  bf16: see forward()."""
    bs = net['batch_size']
    multi = isinstance(inputs, (list, tuple))
    ins = list(inputs) if multi else [inputs]
    n = ins[0].shape[0]
    pad = int(bs * np.ceil(n / float(bs)))
    out = np.zeros((pad,) + tuple(net['out_dim'][1:]), dtype=ins[0].dtype)
    for i in range(pad // bs):
        chunks = []
        for a in ins:
            chunk = a[i * bs:(i + 1) * bs]
            if chunk.shape[0] < bs:
                chunk = np.concatenate([chunk, np.repeat(a[-1:], bs - chunk.shape[0], axis=0)], axis=0)
            chunks.append(chunk)
        o, _ = forward(net, P, chunks if multi else chunks[0], train=False, bf16=bf16, store16=store16)
        out[i * bs:(i + 1) * bs] = o.reshape((bs,) + tuple(net['out_dim'][1:]))
    return out[:n]


# --------------------------------------------------------------------------- synthetic inputs
def synthetic_crops(rng, n, h=128, w=128, dtype=np.float32):
    """SURVEY.md section 8(d) cfg 1-2: far-plane background +1.0 (dataset.py:98-100) and a random
    convex blob (ellipse) covering roughly 25-45 % of the pixels with values U(-1, 0.6)."""
    x = np.ones((n, 1, h, w), dtype=dtype)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        frac = rng.uniform(0.25, 0.45)
        ar = rng.uniform(0.6, 1.6)
        area = frac * h * w
        a = np.sqrt(area * ar / np.pi)
        b = area / (np.pi * a)
        cy, cx = h / 2. + rng.uniform(-8, 8), w / 2. + rng.uniform(-8, 8)
        th = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        msk = (u / a) ** 2 + (v / b) ** 2 <= 1.0
        vals = rng.uniform(-1, 0.6, size=(h, w))
        x[i, 0][msk] = vals[msk]
    return x
