"""
hipdp.evalfuse -- the deterministic-mode (test-time) lowering of a whole residual block to ONE launch.

`computeOutput` (/root/reference/src/net/netbase.py:217-316) evaluates the net with every BatchNormLayer in deterministic mode
(/root/reference/src/net/batchnormlayer.py:158-159): a stored per-channel affine.  The training engine has to split a bottleneck block
(/root/reference/src/net/resnet.py:349-414) at every BatchNorm -- the batch statistics are a grid-wide dependency -- and the inference
engine used to issue that same decomposition (~6 launches per block).  Here the graph of

    inputVar + conv1x1(relu(bn(conv3x3(relu(bn(conv1x1(relu(bn(inputVar)))))))))                      identity block
    conv1x1(relu(bn(conv3x3(relu(bn(conv1x1_s(h))))))) + conv1x1_s(h),   h = relu(bn(inputVar))      projection block

is recognised on the `add` node and handed to dpp_resblock_eval (csrc/resblock.hip), which keeps the 16- / 32- / 64-channel
intermediates in LDS.  Anything that does not match (other widths, dropout in between, a consumer of an intermediate value) stays on
the layer-by-layer path.  Round 6: the bf16 mode of BASELINE config 5 (bf16-stored tensors, bf16 MFMA operands) has its form of the kernel.
"""
from . import ops


def _kind(layer):
    return layer.__class__.__name__


def _layer_of(var, kind):
    return var.layer if (var is not None and var.kind == 'layer' and _kind(var.layer) == kind) else None


def _bn_relu_under(eng, var, sole=True):
    """var = NonlinearityLayer(ReLU)(BatchNormLayer(x)) -> (bn layer, x var), else None.  `sole`: the ReLU output and the BatchNorm
    output must feed nothing else."""
    nl = _layer_of(var, 'NonlinearityLayer')
    if nl is None or nl.cfgParams.activation is None or nl.cfgParams.activation_str != 'ReLU':
        return None
    if sole and not eng._single_consumer(var):
        return None
    bv = var.inputs[0]
    bn = _layer_of(bv, 'BatchNormLayer')
    if bn is None or not eng._single_consumer(bv):
        return None
    return bn, bv.inputs[0]


def _conv(var, k):
    l = _layer_of(var, 'ConvLayer')
    if l is None:
        return None
    c = l.cfgParams
    if tuple(c.filterDim) != (k, k) or c.border_mode not in ('half', 'same') or c.activation is not None or not c.hasBias:
        return None
    if c.stride[0] != c.stride[1]:
        return None
    return l


def match_block(eng, add_var):
    """The bottleneck block ending in this `add` node as dict(root=, bn0=, conv1=, bn1=, conv2=, bn2=, conv3=, shortcut=, stride=), or None."""
    a, b = add_var.inputs
    for q, p in ((a, b), (b, a)):
        conv3 = _conv(q, 1)
        if conv3 is None or conv3.cfgParams.stride[0] != 1 or not eng._single_consumer(q):
            continue
        m2 = _bn_relu_under(eng, q.inputs[0])
        if m2 is None:
            continue
        bn2, v2 = m2
        conv2 = _conv(v2, 3)
        if conv2 is None or conv2.cfgParams.stride[0] != 1 or not eng._single_consumer(v2):
            continue
        m1 = _bn_relu_under(eng, v2.inputs[0])
        if m1 is None:
            continue
        bn1, v1 = m1
        conv1 = _conv(v1, 1)
        if conv1 is None or not eng._single_consumer(v1):
            continue
        hvar = v1.inputs[0]
        shortcut = _conv(p, 1)
        proj = shortcut is not None and p.inputs[0] is hvar and eng._single_consumer(p)
        m0 = _bn_relu_under(eng, hvar, sole=False)
        if m0 is None:
            continue
        bn0, root = m0
        n_h = len(eng.consumers.get(id(hvar), []))
        stride = conv1.cfgParams.stride[0]
        if proj:
            if n_h != 2 or shortcut.cfgParams.stride[0] != stride or shortcut.cfgParams.nFilters != conv3.cfgParams.nFilters:
                continue
        else:
            if n_h != 1 or p is not root or stride != 1:
                continue
        return dict(root=root, bn0=bn0, conv1=conv1, bn1=bn1, conv2=conv2, bn2=bn2, conv3=conv3, shortcut=shortcut if proj else None,
                    stride=stride, q=q, p=p)
    return None


def emit_block(eng, add_var):
    """Lower the block ending in `add_var` to one dpp_resblock_eval launch and return the View of its output, or None (not this shape:
    the caller emits the block layer by layer)."""
    if eng.train:
        return None
    if eng.prec or eng.store16:
        # bf16 mode (round 6): the fused block models the DEFAULT bf16 path -- bf16-stored conv outputs and bf16 MFMA operands in every
        # product of the block; any other combination of the bf16 knobs stays layer by layer
        from . import heuristics as hz
        if not (eng.prec and eng.store16 and hz.BF16_GEMM and hz.BF16_GEMM_ALL and hz.EVAL_FUSE_BF16):
            return None
    m = match_block(eng, add_var)
    if m is None:
        return None
    rt, st = eng.rt, eng.store
    c1, c3 = m['conv1'].cfgParams, m['conv3'].cfgParams
    Cin, Nb, Cout, s = c1.inputDim[1], c1.nFilters, c3.nFilters, m['stride']
    if m['conv2'].cfgParams.nFilters != Nb or m['conv2'].cfgParams.inputDim[1] != Nb:
        return None
    if not rt.lib.dpp_resblock_eval_ok(Cin, Cout, Nb, s, 1 if m['shortcut'] is not None else 0):
        return None
    src = eng._emit(m['root'])
    if not src.plain or len(src.shape) != 4 or (src.base.is16 and not eng.store16):
        return None
    N, H, W, Ci = src.base.shape
    if Ci != Cin:
        return None
    _, Co_, Ho, Wo = c3.outputDim
    if not (Co_ == Cout and Ho == -(-H // s) and Wo == -(-W // s)):
        return None

    def bn(layer):
        return ops.bn_eval(st.view(layer.mean), st.view(layer.inv_std), st.view(layer.gamma), st.view(layer.beta))
    out = eng._new_tensor((N, Ho, Wo, Cout), 'block%d' % m['conv3'].layerNum, act=True)
    kw = {}
    if m['shortcut'] is not None:
        kw = dict(Wsc=st.view(m['shortcut'].W), bsc=st.view(m['shortcut'].b))
    launch = ops.resblock_eval(rt, src.base.buf, N, H, W, Cin, s, Cout, Nb, bn(m['bn0']), bn(m['bn1']), bn(m['bn2']),
                               st.view(m['conv1'].W), st.view(m['conv1'].b), st.view(m['conv2'].W), st.view(m['conv2'].b),
                               st.view(m['conv3'].W), st.view(m['conv3'].b), out.buf, name='resblock_%d' % m['conv3'].layerNum, **kw)
    if rt.lib.dpp_resblock_eval_check(launch.args[0]) != 0:
        # the kernel refuses THIS descriptor (pointer alignment, LDS size, 32-bit offsets, storage combination): layer by layer, decided
        # while the net is compiled -- not a failed forward pass later
        eng._drop_tensor(out)
        return None
    eng.fwd.add(launch)
    eng.fused_blocks.append(m)
    from .engine import View
    return View(out)
