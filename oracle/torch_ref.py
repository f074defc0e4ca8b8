"""
oracle/torch_ref.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

An INDEPENDENT restatement of the same graphs on PyTorch-CPU ops with autograd.
Two jobs:
  1. cross-check oracle/nets.py (hand-written analytic backward) against autograd
     in float64 -- the pin that replaces the reference's missing tests;
  2. the reported-only CPU baseline of bench.py ("CPU restatement (PyTorch-CPU),
     not Theano"): the identical fp32 graph, BN in training mode,
     sum-squared-error loss, the reference's ADAM.

Semantics follow the same reference lines as oracle/layers.py; torch's conv2d is a
cross-correlation, so kernels are flipped to get Theano's true convolution.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = float(np.float32(1e-4))


def to_torch(P, dtype=torch.float64, requires_grad=True):
    T, seen = {}, {}
    for i, v in P.items():
        if id(v) in seen:                            # layers that share parameters (copyLayer) share the leaf tensors
            T[i] = seen[id(v)]
            continue
        T[i] = [torch.tensor(np.asarray(a), dtype=dtype) for a in v]
        for t in T[i][:2]:
            t.requires_grad_(requires_grad)
        seen[id(v)] = T[i]
    return T


def _conv(a, W, stride, border):
    kh, kw = W.shape[2], W.shape[3]
    pad = (kh // 2, kw // 2) if border in ('half', 'same') else (0, 0)
    full = F.conv2d(a, torch.flip(W, dims=(2, 3)), None, stride=1, padding=pad)
    return full[:, :, ::stride[0], ::stride[1]]        # Theano subsample


class _MaxPoolTies(torch.autograd.Function):
    """Non-overlapping max-pool whose gradient goes to EVERY window element equal to the maximum (Theano's MaxPoolGrad;
    torch's own max_pool2d picks one index, which differs on the constant background of a depth crop).  `ties` (optional,
    bool [N][C][oh][ph][ow][pw]) pins the tie pattern to one decided elsewhere (the device's, see `masks` in forward)."""

    @staticmethod
    def forward(ctx, c, ph, pw, ties=None):
        N, C, H, W = c.shape
        oh, ow = H // ph, W // pw
        v = c[:, :, :oh * ph, :ow * pw].reshape(N, C, oh, ph, ow, pw)
        y = v.amax(dim=(3, 5))
        ctx.save_for_backward(v == y[:, :, :, None, :, None] if ties is None else ties)
        ctx.dims = (N, C, H, W, oh, ph, ow, pw)
        return y

    @staticmethod
    def backward(ctx, gy):
        (ties,) = ctx.saved_tensors
        N, C, H, W, oh, ph, ow, pw = ctx.dims
        g = torch.zeros((N, C, H, W), dtype=gy.dtype)
        g[:, :, :oh * ph, :ow * pw] = (ties * gy[:, :, :, None, :, None]).reshape(N, C, oh * ph, ow * pw)
        return g, None, None, None


def bf16r(t):
    """Round to bfloat16 (nearest even) and come back in t's dtype: what the bf16 kernels do to an MFMA operand when they stage it."""
    return t.detach().to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _QBilinear(torch.autograd.Function):
    """y = fn(a, W) for a bilinear fn (a convolution without bias, a matrix product) evaluated the way the bf16 kernels of BASELINE
    config 5 do it -- operands rounded to bfloat16, accumulation in the working precision -- in EACH of the three passes, per
    layer as the device's launches are configured (tests/pinning.py:device_quant reads that off the compiled net):
        forward        fn(q(a), q(W))                        q = bf16r if 'fwd' else identity
        data gradient  fn^T_a(q(dY), q(W))                   if 'dgrad', else the unrounded fn^T_a(dY, W)
        filter grad.   fn^T_W(q(a), q(dY))                   if 'wgrad', else fn^T_W(a, dY)
    (rounding is NOT differentiated: the data gradient goes straight to `a`, as on the device).  `pin`: the device's own rounded
    forward operand -- a float32 and a float64 evaluation of the same activation round to different bfloat16 neighbours for a few
    elements in 1e5, and one such flip is 100x a float32 rounding error; with the device's operand handed in, what is compared is
    the arithmetic of the kernels (the same idea as the ReLU / pooling masks of forward()).  `pin_dy`: likewise the device's rounded
    output gradient, the operand of the two backward products."""

    @staticmethod
    def forward(ctx, a, W, fn, q):
        # the rounded activation: the forward product's operand if 'fwd', the filter gradient's if 'wgrad' (round 6: a layer's filter
        # gradient may run on bf16 operands while its forward product does not -- K = 16 layers -- and the other way round)
        ar = q['pin'].to(a.dtype) if q.get('pin') is not None else bf16r(a)
        aq = ar if q.get('fwd') else a.detach()
        Wq = bf16r(W) if q.get('fwd') else W.detach()
        ctx.fn, ctx.q = fn, q
        ctx.save_for_backward(a.detach(), W.detach(), ar if q.get('wgrad') else a.detach())
        return fn(aq, Wq)

    @staticmethod
    def backward(ctx, gy):
        a, W, aq = ctx.saved_tensors
        fn, q = ctx.fn, ctx.q
        with torch.enable_grad():
            gq = q['pin_dy'].to(gy.dtype) if q.get('pin_dy') is not None else bf16r(gy)
            av = a.clone().requires_grad_(True)
            (da,) = torch.autograd.grad(fn(av, bf16r(W) if q.get('dgrad') else W), av, gq if q.get('dgrad') else gy)
            Wv = W.clone().requires_grad_(True)
            (dW,) = torch.autograd.grad(fn(aq, Wv), Wv, gq if q.get('wgrad') else gy)          # aq: rounded iff 'wgrad' (see forward)
        return da, dW, None, None


class _StoredBN(torch.autograd.Function):
    """BatchNorm over a bf16-STORED tensor (BASELINE config 5, ABI v9 DPP_ST_*): the conv epilogue forms v in the working precision,
    takes the batch statistics FROM v and stores y = bf16(v); every later reader -- this BatchNorm's normalisation in the consumer's
    prologue, the ReLU-mask and the xhat of its backward pass -- sees y.  Forward: (y - mean(v)) * gamma * inv_std(v) + beta.
    Backward: what the kernels compute (csrc/bn.hip, the data-gradient epilogues): with xhat = (y - mean) * inv_std,
        dbeta = sum g, dgamma = sum g * xhat, dv = gamma * inv_std * (g - mean(g) - xhat * mean(g * xhat))
    -- the batchnormlayer.py:154-192 gradient with y in place of x (rounding is not differentiated; with y == v it IS the exact
    gradient).  Deterministic mode (running statistics): dv = gamma * inv_std * g."""

    @staticmethod
    def forward(ctx, v, y, gamma, beta, rm, ris, train, pin_g=None, rec=None):
        ctx.pin_g, ctx.rec = pin_g, rec
        if train:
            mean = v.mean(dim=(0, 2, 3))
            inv_std = 1.0 / torch.sqrt(v.var(dim=(0, 2, 3), unbiased=False) + BN_EPS)
        else:
            mean, inv_std = rm, ris
        xhat = (y - mean[None, :, None, None]) * inv_std[None, :, None, None]
        ctx.save_for_backward(xhat, gamma, inv_std)
        ctx.train = train
        ctx.stats = (mean.detach(), inv_std.detach())
        return xhat * gamma[None, :, None, None] + beta[None, :, None, None]

    @staticmethod
    def backward(ctx, g):
        xhat, gamma, inv_std = ctx.saved_tensors
        if ctx.rec is not None:
            ctx.rec(bf16r(g))                    # what THIS evaluation would have stored as the masked gradient G
        if ctx.pin_g is not None:                # bf16-stored gradients: the device's own G (see forward(): grad_pins)
            g = ctx.pin_g.to(g.dtype)
        dbeta = g.sum(dim=(0, 2, 3))
        dgamma = (g * xhat).sum(dim=(0, 2, 3))
        M = g.shape[0] * g.shape[2] * g.shape[3]
        s = (gamma * inv_std)[None, :, None, None]
        if ctx.train:
            dv = s * (g - (dbeta / M)[None, :, None, None] - xhat * (dgamma / M)[None, :, None, None])
        else:
            dv = s * g
        return dv, None, dgamma, dbeta, None, None, None, None, None


def _relu(a, mask=None):
    """T.maximum(a, 0); with `mask` the pass / block decision of every element is the given one (value a where it passes,
    gradient 1 there): pins the ReLU pattern of a float32 evaluation when comparing gradients, see forward()."""
    return torch.clamp_min(a, 0) if mask is None else a * mask.to(a.dtype)


def forward(net, T, x, train, masks=None, quant=None, store=None, grad_pins=None):
    """grad_pins (optional, with `store`): (G, dV) -- the bf16-stored GRADIENT tensors of the device, pinned like the stored activations:
    G {index of a 'bn' layer: the masked gradient its backward kernels read}, dV {index of a stored layer: the gradient of its tensor
    as the device holds it, i.e. the BatchNorm-backward dX plus the identity path of a residual sum}.  Each replaces the oracle's own
    value at that point of the backward pass (the oracle's own rounding is collected in stats['stored_grads'] for the un-pinned
    agreement check), so what is compared downstream is the arithmetic of each layer on the device's own operands.
    store (optional): the bf16 STORAGE mode -- a dict {layer index: pin or None} naming every layer whose materialised output the
    device holds as bfloat16: the stem ('convpool') and every 'conv' (for a conv fused into a residual add the tensor is the SUM, see
    oracle.nets.fused_convs).  The value is the device's own stored tensor (NCHW, bf16 values as float32) when the comparison pins it
    -- a float32 and a float64 evaluation of v round to different bfloat16 neighbours for a few elements in 1e4, and one flip is worth
    a hundred float32 rounding errors (the reason for `pin` in _QBilinear) -- or None: round here.  Statistics come from the unrounded
    value, see _StoredBN.  `stored_out` of the returned stats dict collects what this evaluation would have stored (unpinned), so
    that a test can ALSO check the pins themselves against it.
    quant (optional): {layer index of a 'conv' / 'fc' layer: dict(fwd=, dgrad=, wgrad=, pin=)} -- the layers whose products run
    on bf16 operands, see _QBilinear.
    masks (optional): {layer index: bool array} -- for a 'relu' layer or an 'fc' layer with ReLU the elements that pass,
    for a 'convpool' layer with pooling the tie pattern [N][C][oh][ow][ph*pw] of the windows (and, under the key
    ('relu', layer index), the pass pattern of its own ReLU).  The gradient of a deep ReLU net
    is discontinuous in its inputs: two float32 evaluations (or one float32 and one float64) disagree on the sign of the
    handful of activations that sit within rounding of zero, and every such flip moves all upstream gradients by a fraction
    of a percent.  Handing the device's own decisions to the oracle removes that (legitimate) difference, so that what is
    left is the arithmetic of the kernels, comparable at float32 round-off."""
    vals, stats = {}, {}
    masks = masks or {}
    quant = quant or {}
    raw, memo, own_round, own_grads = {}, {}, {}, {}
    gpin, dvpin = grad_pins if grad_pins is not None else ({}, {})
    fused = set()
    if store is not None:
        from oracle.nets import fused_convs
        fused = fused_convs(net)

    def keep(key, i, v):
        """Materialise v as layer i's stored tensor: forward value = the rounded one (pinned or own), gradient straight through."""
        own = bf16r(v)
        own_round[i] = own
        pin = store.get(i)
        y = own if pin is None else torch.as_tensor(np.asarray(pin)).to(v.dtype)
        raw[key] = v
        if i in dvpin and v.requires_grad:
            dpin = torch.as_tensor(np.asarray(dvpin[i])).to(v.dtype)

            def hook(grad, i=i, dpin=dpin):
                own_grads[('dv', i)] = bf16r(grad)
                return dpin
            v.register_hook(hook)
        return v + (y - v).detach()

    def qd(i):
        q = quant.get(i)
        if q is None:
            return None
        q = dict(q)
        for key in ('pin', 'pin_dy'):
            if q.get(key) is not None:
                q[key] = torch.as_tensor(np.asarray(q[key]))
        return q

    def mk(i):
        m = masks.get(i)
        return None if m is None else torch.as_tensor(np.asarray(m))

    def get(ref):
        if ref[0] == 'input':
            return x[ref[1]] if len(ref) > 1 else x
        if ref[0] == 'layer':
            return vals[ref[1]]
        if ref[0] == 'add':
            if store is None:
                return get(ref[1]) + get(ref[2])
            if ref not in memo:                              # one tensor, however many consumers: rounded once
                cand = [r[1] for r in ref[1:] if r[0] == 'layer' and r[1] in fused]
                memo[ref] = keep(ref, max(cand), get(ref[1]) + get(ref[2]))
            return memo[ref]
        if ref[0] == 'flatten':
            return get(ref[1]).flatten(1)
        if ref[0] == 'concat':
            return torch.cat([get(r) for r in ref[1:]], dim=1)
        raise ValueError(ref)

    for i, l in enumerate(net['layers']):
        a = get(l['src'])
        k = l['kind']
        if k == 'convpool':
            c = _conv(a, T[i][0], l['stride'], l['border'])
            if tuple(l['pool']) != (1, 1):
                ph, pw = int(l['pool'][0]), int(l['pool'][1])
                ties = mk(i)
                if ties is not None:
                    N_, C_, oh, ow = ties.shape[:4]
                    ties = ties.reshape(N_, C_, oh, ow, ph, pw).permute(0, 1, 2, 4, 3, 5)
                c = _MaxPoolTies.apply(c, ph, pw, ties)
            c = c + T[i][1][None, :, None, None]
            vals[i] = _relu(c, mk(('relu', i))) if l['act'] == 'relu' else c
            if store is not None and i in store:
                vals[i] = keep(('layer', i), i, vals[i])
        elif k == 'conv':
            q = qd(i)
            if q is None:
                c = _conv(a, T[i][0], l['stride'], l['border'])
            else:
                c = _QBilinear.apply(a, T[i][0], (lambda u, w, l=l: _conv(u, w, l['stride'], l['border'])), q)
            vals[i] = c + T[i][1][None, :, None, None]
            if store is not None and i in store and i not in fused:
                vals[i] = keep(('layer', i), i, vals[i])
        elif k == 'bn':
            beta, gamma, rm, ris = T[i]
            if store is not None and l['src'] in raw:
                pg = torch.as_tensor(np.asarray(gpin[i])) if i in gpin else None
                fn = _StoredBN.apply(raw[l['src']], a.detach(), gamma, beta, rm, ris, bool(train), pg,
                                     (lambda t, i=i: own_grads.__setitem__(('g', i), t)) if i in gpin else None)
                if train:
                    v_ = raw[l['src']].detach()
                    stats[i] = (v_.mean(dim=(0, 2, 3)), 1.0 / torch.sqrt(v_.var(dim=(0, 2, 3), unbiased=False) + BN_EPS))
                vals[i] = fn
                continue
            if train:
                mean = a.mean(dim=(0, 2, 3))
                var = a.var(dim=(0, 2, 3), unbiased=False)
                inv_std = 1.0 / torch.sqrt(var + BN_EPS)
                stats[i] = (mean.detach(), inv_std.detach())
            else:
                mean, inv_std = rm, ris
            vals[i] = (a - mean[None, :, None, None]) * (gamma * inv_std)[None, :, None, None] \
                + beta[None, :, None, None]
        elif k == 'relu':
            vals[i] = _relu(a, mk(i))
        elif k == 'fc':
            q = qd(i)
            pre = (a @ T[i][0] if q is None else _QBilinear.apply(a, T[i][0], (lambda u, w: u @ w), q)) + T[i][1]
            vals[i] = _relu(pre, mk(i)) if l['act'] == 'relu' else pre
        elif k == 'dropout':
            vals[i] = a * float(np.float32(1.0 - l['p'])) if not train else a   # masks not modelled here
        else:
            raise NotImplementedError(k)
    if store is not None:
        stats['stored_out'] = {i: t.detach().numpy() for i, t in own_round.items()}
        stats['stored_grads'] = own_grads                    # filled by the backward pass
    return get(net['out']), stats


def cost_and_grads(net, P, x, y, dtype=torch.float64, masks=None, quant=None, store=None, stored_out=None, grad_pins=None, stored_grads=None):
    """Returns (cost, {layer: [g0, g1]}, out) with autograd gradients (numpy arrays); `masks`, `quant`, `store`: see forward().
    stored_out (a dict, optional) receives forward()'s own roundings of the stored tensors."""
    T = to_torch(P, dtype)
    xt = [torch.as_tensor(a, dtype=dtype) for a in x] if isinstance(x, (list, tuple)) else torch.tensor(x, dtype=dtype)
    yt = torch.tensor(y, dtype=dtype)
    out, st = forward(net, T, xt, True, masks, quant, store, grad_pins)
    if stored_out is not None and store is not None:
        stored_out.update(st['stored_out'])
    cost = ((out - yt) ** 2).sum(dim=1).mean()
    cost.backward()
    if stored_grads is not None and store is not None:
        stored_grads.update({k: t.detach().numpy() for k, t in st['stored_grads'].items()})
    G = {i: [T[i][0].grad.numpy(), T[i][1].grad.numpy()] for i in T}
    return float(cost.detach()), G, out.detach().numpy()


class TorchTrainer(object):
    """fp32 train step on PyTorch-CPU for the CPU baseline: forward (BN batch stats) + autograd
    backward + the reference's ADAM (optimizer.py:58-90) + BN running-stat EMA."""

    def __init__(self, net, P, dtype=torch.float32):
        self.net = net
        self.T = to_torch(P, dtype)
        self.params = [t for i in sorted(self.T) for t in self.T[i][:2]]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 1.0
        self.dtype = dtype

    def step(self, x, y, lr):
        out, stats = forward(self.net, self.T, x, True)
        cost = ((out - y) ** 2).sum(dim=1).mean()
        grads = torch.autograd.grad(cost, self.params)
        b1, b2, eps = float(np.float32(0.9)), float(np.float32(0.999)), float(np.float32(1e-8))
        with torch.no_grad():
            c1 = 1.0 - b1 ** self.t
            c2 = 1.0 - b2 ** self.t
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                m.mul_(b1).add_(g, alpha=1.0 - b1)
                v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                p.sub_(lr * (m / c1) / (torch.sqrt(v / c2) + eps))
            for i, (mean, inv_std) in stats.items():
                self.T[i][2].mul_(0.9).add_(mean, alpha=0.1)
                self.T[i][3].mul_(0.9).add_(inv_std, alpha=0.1)
        self.t += 1.0
        return float(cost)

    def forward_eval(self, x):
        with torch.no_grad():
            out, _ = forward(self.net, self.T, x, False)
        return out
