#!/usr/bin/env python3
"""
Generate the committed golden fixtures under tests/golden/ by IMPORTING the parts of the
reference that can run in this container.  Runs only here:
/root/reference does not exist on the GPU box, and nothing from the reference is copied --
the fixtures hold inputs and the reference's OUTPUTS only.

What the reference executes here, and what it pins:
  shapes.json      the reference's own *LayerParams classes (net/convlayer.py:131-163,
                   net/convpoollayer.py:145-181, net/batchnormlayer.py:40-58, net/hiddenlayer.py:40-79,
                   net/nonlinearitylayer.py:42-72, net/dropoutlayer.py:39-61) evaluate outputDim /
                   filter_shape / getMemoryRequirement for every layer of ResNet type 0/1 (128^2, 256^2);
                   and the reference's real PoseRegNetParams (net/poseregnet.py:44-145) builds its own
                   layer list for types 0 and 11.
  geometry.npz     data/transformations.py rotatePoint2D / transformPoint2D, the importers' pinhole
                   (un)projections (data/importers.py:80-119, 756-793, 1187-1224) and
                   HandDetector.comToBounds (util/handdetector.py:204-226) on seeded inputs.
  chunks.json      util/helpers.py chunks() as used by NetTrainer.chunksForMP (nettrainer.py:726-744).
  crop.npz         HandDetector.__init__ depth-range preprocessing, comToBounds, getCrop, calculateCoM and
                   refineCoMIterative (util/handdetector.py:53-130, 204-226, 260-296, 540-558) on seeded synthetic frames
                   (NumPy / SciPy only; the cv2 resize of cropArea3D cannot run here).
  poses.npz        HandDetector.sampleRandomPoses (util/handdetector.py:805-909) with the ICVL and NYU importers.
  shapes.json also holds the reference's real ScaleNetParams (net/scalenet.py:33-127).
  init.npz         Layer.getInitVals (net/layer.py:70-124) for the He / Xavier / sigmoid / tanh rules and the orthogonal option.
  evaluation.json  HandposeEvaluation metrics (util/handpose_evaluation.py:92-228).
  trainer.json     NetTrainerParams.lr_of_ep and NetTrainer.alignData (trainer/nettrainer.py:47-72, 365-413).

Python-2-only modules (netbase.py, handdetector.py, importers.py use print statements / cPickle) are
converted IN MEMORY with lib2to3 and exec'd; nothing is written to disk.  `cv2`, `progressbar` and
`theano` are absent: empty placeholder modules satisfy the import statements and are never called by
the functions exercised here (pure NumPy arithmetic).  Python-2 integer division is NOT reproduced by
such an import, so comToTransform (handdetector.py:246,249) is not taken from it here: make_golden_r2.py
runs it with those two int / int sites rewritten to `//` (transform.npz).
"""
import builtins
import json
import os
import sys
import types

import numpy

REF = '/root/reference/src'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, '..', '..'))


class _Cast(dict):
    def __getitem__(self, k):
        return lambda x: numpy.asarray(x, dtype=k)


numpy.cast = _Cast()            # removed in NumPy 2; util/theano_helpers.py:35 needs it at import
numpy.float = float             # handdetector.py uses numpy.float
builtins.xrange = range


def load_py2_module(name, relpath):
    """exec a Python-2 source file of the reference after an in-memory lib2to3 pass."""
    from lib2to3 import refactor
    fixers = refactor.get_fixers_from_package('lib2to3.fixes')
    tool = refactor.RefactoringTool(fixers)
    src = open(os.path.join(REF, relpath)).read() + '\n'
    code = str(tool.refactor_string(src, relpath))
    mod = types.ModuleType(name)
    mod.__file__ = os.path.join(REF, relpath)
    sys.modules[name] = mod
    exec(compile(code, mod.__file__, 'exec'), mod.__dict__)
    return mod


for missing in ('cv2', 'progressbar', 'theano'):
    if missing not in sys.modules:
        sys.modules[missing] = types.ModuleType(missing)     # never called below

from net.convlayer import ConvLayerParams                      # noqa: E402
from net.convpoollayer import ConvPoolLayerParams              # noqa: E402
from net.batchnormlayer import BatchNormLayerParams            # noqa: E402
from net.hiddenlayer import HiddenLayerParams                  # noqa: E402
from net.nonlinearitylayer import NonlinearityLayerParams      # noqa: E402
from net.dropoutlayer import DropoutLayerParams                # noqa: E402
from util.theano_helpers import ReLU                           # noqa: E402
from data.transformations import rotatePoint2D, transformPoint2D   # noqa: E402
from util.helpers import chunks                                # noqa: E402

from oracle import nets                                        # noqa: E402


def ref_layer_params(l):
    """Instantiate the reference's Params class for one oracle layer spec."""
    k = l['kind']
    if k == 'convpool':
        return ConvPoolLayerParams(inputDim=l['in_dim'], nFilters=l['nf'], filterDim=l['k'], stride=l['stride'],
                                   poolsize=l['pool'], border_mode='same' if l['border'] == 'half' else l['border'],
                                   activation=ReLU if l['act'] == 'relu' else None, init_method='He')
    if k == 'conv':
        return ConvLayerParams(inputDim=l['in_dim'], nFilters=l['nf'], filterDim=l['k'], stride=l['stride'],
                               border_mode='same', activation=None, init_method='He')
    if k == 'bn':
        return BatchNormLayerParams(inputDim=l['in_dim'])
    if k == 'relu':
        return NonlinearityLayerParams(inputDim=l['in_dim'], activation=ReLU)
    if k == 'fc':
        return HiddenLayerParams(inputDim=l['in_dim'], outputDim=l['out_dim'],
                                 activation=ReLU if l['act'] == 'relu' else None)
    if k == 'dropout':
        return DropoutLayerParams(inputDim=l['in_dim'], outputDim=l['out_dim'])
    raise ValueError(k)


def describe(p):
    d = dict(cls=p.__class__.__name__, inputDim=[int(v) for v in p.inputDim],
             outputDim=[int(v) for v in p.outputDim])
    if hasattr(p, 'filter_shape'):
        d['filter_shape'] = [int(v) for v in p.filter_shape]
        d['stride'] = [int(v) for v in p.stride]
        d['border_mode'] = p.border_mode
    if hasattr(p, 'poolType'):
        d['poolType'] = int(p.poolType)
        d['poolsize'] = [int(v) for v in p.poolsize]
    if hasattr(p, 'getMemoryRequirement'):
        d['mem'] = int(p.getMemoryRequirement())
    if hasattr(p, 'activation'):
        d['activation'] = p.activation_str
    if hasattr(p, 'epsilon'):
        d['epsilon'], d['alpha'] = p.epsilon, p.alpha
    if hasattr(p, 'p'):
        d['p'] = p.p
    return d


def make_shapes():
    out = {}
    for name, kw in (('resnet_t0_128', dict(type=0, numJoints=1, nDims=30)),
                     ('resnet_t1_128', dict(type=1, numJoints=14, nDims=3)),
                     ('resnet_t1_256', dict(type=1, numJoints=14, nDims=3, wIn=256, hIn=256)),
                     ('resnet_t0_64_b4', dict(type=0, numJoints=1, nDims=30, wIn=64, hIn=64, batchSize=4))):
        net = nets.build_resnet(**kw)
        out[name] = dict(args=kw, layers=[describe(ref_layer_params(l)) for l in net['layers']])
    # the reference's real PoseRegNetParams builds its own list
    load_py2_module('net.netbase', 'net/netbase.py')
    from net.poseregnet import PoseRegNetParams
    for name, kw in (('poseregnet_t0', dict(type=0, numJoints=1, nDims=30)),
                     ('poseregnet_t11', dict(type=11, numJoints=14, nDims=3)),
                     ('poseregnet_t0_b16', dict(type=0, numJoints=16, nDims=3, batchSize=16))):
        cfg = PoseRegNetParams(**kw)
        out[name] = dict(args=kw, layers=[describe(p) for p in cfg.layers],
                         inputDim=[int(v) for v in cfg.inputDim], outputDim=[int(v) for v in cfg.outputDim])
        # (NetBaseParams.getMemoryRequirement raises for nets with dropout in the reference itself:
        #  DropoutLayerParams has no getMemoryRequirement, net/netbase.py:65-73)
    from net.scalenet import ScaleNetParams
    for name, kw in (('scalenet_t1', dict(type=1, numJoints=1, nDims=3)),
                     ('scalenet_t1_96_b8', dict(type=1, numJoints=1, nDims=3, wIn=96, hIn=96, batchSize=8))):
        cfg = ScaleNetParams(**kw)
        out[name] = dict(args=kw, layers=[describe(p) for p in cfg.layers], inputDim=[[int(v) for v in d] for d in cfg.inputDim],
                         outputDim=[int(v) for v in cfg.outputDim])
    json.dump(out, open(os.path.join(HERE, 'shapes.json'), 'w'), indent=0, sort_keys=True)
    return out


def make_crop_and_poses():
    """Functions of the reference's HandDetector that are NumPy / SciPy only, on the oracle's seeded synthetic frames."""
    from oracle import augment as A
    hd_mod = sys.modules.get('util.handdetector') or load_py2_module('util.handdetector', 'util/handdetector.py')
    imp = sys.modules.get('data.importers') or load_py2_module('data.importers', 'data/importers.py')
    d = {}
    cam = A.Camera.icvl()
    frames, coms = A.synthetic_frames(numpy.random.RandomState(77), 4, cam, 120, 160, (250., 250., 250.))
    d['frames'], d['coms'] = frames, coms
    pre, rng_, nd, bounds, crops, com_full, com_crop, com_it = [], [], [], [], [], [], [], []
    for i in range(4):
        hd = hd_mod.HandDetector(frames[i].copy(), 241.42, 241.42)
        pre.append(hd.dpt.copy())
        rng_.append((hd.minDepth, hd.maxDepth))
        nd.append(0.)       # getNDValue needs the scipy.stats.mode of 2014 (returns a scalar today): not taken from the reference
        b = hd.comToBounds(coms[i], (250., 250., 250.))
        bounds.append(b)
        c = hd.getCrop(hd.dpt, *b)
        crops.append(numpy.asarray(c.shape + (0,), 'float64'))          # ragged: keep shape, then the flat data
        d['crop_%d' % i] = c
        com_full.append(hd.calculateCoM(hd.dpt))
        com_crop.append(hd.calculateCoM(c))
        com_it.append(hd.refineCoMIterative(coms[i].astype('float64'), 3, (250., 250., 250.)))
    d['pre'], d['range'], d['nd'] = numpy.stack(pre), numpy.asarray(rng_, 'float64'), numpy.asarray(nd, 'float64')
    d['bounds'] = numpy.asarray(bounds, 'float64')
    d['com_full'], d['com_crop'], d['com_it'] = numpy.stack(com_full), numpy.stack(com_crop), numpy.stack(com_it)
    numpy.savez_compressed(os.path.join(HERE, 'crop.npz'), **d)

    p = {}
    for nm, cls, J, args in (('icvl', imp.ICVLImporter, 16, (241.42, 241.42, 160., 120.)), ('nyu', imp.NYUImporter, 14, (588.03, 587.07, 320., 240.))):
        o = cls.__new__(cls)
        imp.DepthImporter.__init__(o, *args)
        camx = A.Camera.icvl() if nm == 'icvl' else A.Camera.nyu()
        _, c3, cubes, _, gts = A.synthetic_augment_inputs(numpy.random.RandomState(5), 12, camx, cube=(250., 250., 250.), joints=J)
        p['%s_com' % nm], p['%s_cube' % nm], p['%s_gt' % nm] = c3, cubes, gts
        for tag, modes in (('main', ['com', 'rot', 'none']), ('all', ['com', 'rot', 'sc', 'none', 'rot+com', 'rot+com+sc'])):
            out = hd_mod.HandDetector.sampleRandomPoses(o, numpy.random.RandomState(9), gts, c3, cubes, 300, modes)
            p['%s_%s' % (nm, tag)] = out
    numpy.savez_compressed(os.path.join(HERE, 'poses.npz'), **p)
    return d, p


def make_geometry():
    rng = numpy.random.RandomState(23455)
    hd_mod = load_py2_module('util.handdetector', 'util/handdetector.py')
    imp = load_py2_module('data.importers', 'data/importers.py')
    d = {}
    # 2-D point transforms
    pts = rng.uniform(0, 128, (32, 3)).astype('float32')
    ctr = rng.uniform(40, 90, (32, 2)).astype('float32')
    ang = rng.uniform(-180, 180, 32)
    d['rot_pts'], d['rot_ctr'], d['rot_ang'] = pts, ctr, ang
    d['rot_out'] = numpy.stack([rotatePoint2D(pts[i], ctr[i], ang[i]) for i in range(32)])
    d['rot_out64'] = numpy.stack([rotatePoint2D(pts[i].astype('float64'), ctr[i], ang[i]) for i in range(32)])
    Ms = rng.normal(0, 1, (32, 3, 3))
    Ms[:, 2, :] = [0, 0, 1]
    d['tp_M'] = Ms
    d['tp_out'] = numpy.stack([transformPoint2D(pts[i], Ms[i]) for i in range(32)])
    # projections: construct importers without touching the disk-facing code
    cams = {}
    for nm, cls in (('icvl', imp.ICVLImporter), ('msra', imp.MSRA15Importer), ('nyu', imp.NYUImporter)):
        o = cls.__new__(cls)
        if nm == 'nyu':
            imp.DepthImporter.__init__(o, 588.03, 587.07, 320., 240.)
        else:
            imp.DepthImporter.__init__(o, 241.42, 241.42, 160., 120.)
        cams[nm] = o
    uvd = numpy.stack([rng.uniform(20, 300, 64), rng.uniform(20, 220, 64), rng.uniform(250, 900, 64)], 1)
    uvd[5, 2] = 0.
    for dt in ('float32', 'float64'):
        s = uvd.astype(dt)
        d['proj_in_' + dt] = s
        for nm, o in cams.items():
            x3 = numpy.stack([o.jointImgTo3D(s[i]) for i in range(s.shape[0])])
            d['to3d_%s_%s' % (nm, dt)] = x3
            back_in = x3.astype(dt)
            d['toimg_%s_%s' % (nm, dt)] = numpy.stack([o.joint3DToImg(back_in[i]) for i in range(s.shape[0])])
    # comToBounds (float arithmetic only)
    dpt = numpy.ones((128, 128), 'float32') * 500.
    coms = numpy.stack([rng.uniform(30, 290, 64), rng.uniform(30, 210, 64), rng.uniform(250, 900, 64)], 1).astype('float32')
    cubes = rng.choice([150., 200., 250., 300.], 64)
    bounds = []
    for nm, fx, fy in (('icvl', 241.42, 241.42), ('nyu', 588.03, 587.07)):
        hd = hd_mod.HandDetector(dpt.copy(), fx, fy)
        bounds.append(numpy.array([hd.comToBounds(coms[i], (cubes[i],) * 3) for i in range(64)], 'float64'))
    d['ctb_com'], d['ctb_cube'], d['ctb_icvl'], d['ctb_nyu'] = coms, cubes, bounds[0], bounds[1]
    numpy.savez_compressed(os.path.join(HERE, 'geometry.npz'), **d)
    return d


def make_chunks():
    out = []
    for n, k in ((1024, 128), (1000, 125), (7, 3), (72757, 9095), (128, 16), (5, 8)):
        out.append(dict(n=n, k=k, chunks=[[c[0], c[-1] + 1] for c in chunks(list(range(n)), k)]))
    json.dump(out, open(os.path.join(HERE, 'chunks.json'), 'w'))
    return out


def make_trainer():
    """NetTrainerParams.lr_of_ep and NetTrainer.alignData (trainer/nettrainer.py:47-72, 365-413): schedule and the seeded
    padding of the last minibatch, from the reference's own code (Theano is only imported, never called)."""
    for missing in ('sharedmem', 'psutil'):
        if missing not in sys.modules:
            try:
                __import__(missing)
            except ImportError:
                sys.modules[missing] = types.ModuleType(missing)
    for nm in ('net.convlayer', 'net.convpoollayer'):
        __import__(nm)
    mod = load_py2_module('trainer.nettrainer', 'trainer/nettrainer.py')
    out = {}
    p = mod.NetTrainerParams()
    p.learning_rate = 0.001
    out['lr_of_ep'] = [[ep, float(p.lr_of_ep(ep))] for ep in (0, 1, 1.5, 2, 3, 10, 50, 100)]
    cases = []
    for n, align, pad_random in ((6, 8, True), (300, 128, True), (300, 128, False), (256, 128, True), (1000, 1024, True), (5, 4, True)):
        dummy = types.SimpleNamespace(cfgParams=types.SimpleNamespace(pad_random=pad_random, batch_size=4))
        data = numpy.arange(n, dtype='float32').reshape(n, 1) + 1.          # sample i holds i + 1: the padding shows its source
        padded = mod.NetTrainer.alignData(dummy, data, alignSize=align)
        cases.append(dict(n=n, align=align, pad_random=pad_random, padded=[float(v) for v in padded[:, 0]]))
    out['alignData'] = cases
    json.dump(out, open(os.path.join(HERE, 'trainer.json'), 'w'))
    return out


def make_eval():
    """HandposeEvaluation's numeric metrics (util/handpose_evaluation.py:92-228) from the reference's own class."""
    for m in ('vtk', 'matplotlib', 'matplotlib.pyplot', 'mpl_toolkits', 'mpl_toolkits.mplot3d', 'pylab'):
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:      # noqa: BLE001
                sys.modules[m] = types.ModuleType(m)
    for nm, rel in (('util.handdetector', 'util/handdetector.py'), ('data.importers', 'data/importers.py')):
        if nm not in sys.modules:
            load_py2_module(nm, rel)
    mod = load_py2_module('util.handpose_evaluation', 'util/handpose_evaluation.py')
    rng = numpy.random.RandomState(11)
    gt = rng.normal(0, 60, (25, 14, 3))
    jt = gt + rng.normal(0, 9, (25, 14, 3))
    jt[3, 2] = numpy.nan                                    # the metrics are nan-aware
    hpe = mod.HandposeEvaluation(list(gt), list(jt))
    import scipy.stats
    if not hasattr(scipy.stats, 'nanmedian'):
        scipy.stats.nanmedian = numpy.nanmedian          # removed from SciPy; same definition
    out = dict(gt=gt.tolist(), joints=jt.tolist(), mean=float(hpe.getMeanError()), std=float(hpe.getStdError()),
               median=float(hpe.getMedianError()), max=float(hpe.getMaxError()),
               mean_over_seq=[float(v) for v in hpe.getMeanErrorOverSeq()], max_over_seq=[float(v) for v in hpe.getMaxErrorOverSeq()],
               joint_mean=[float(hpe.getJointMeanError(j)) for j in range(14)], joint_max=[float(hpe.getJointMaxError(j)) for j in range(14)],
               within=[[d, int(hpe.getNumFramesWithinMaxDist(d))] for d in (10, 20, 30, 40)])
    json.dump(out, open(os.path.join(HERE, 'evaluation.json'), 'w'))
    return out


def make_init():
    """Layer.getInitVals (net/layer.py:70-124): the reference's own initialiser with a placeholder `theano.config.floatX`."""
    sys.modules['theano'].config = types.SimpleNamespace(floatX='float32')
    from net.layer import Layer
    d = {}
    for tag, shape, mode, act, method, orth in (('conv_he', (8, 1, 5, 5), 'conv', 'ReLU', 'He', False), ('conv_he_res', (64, 16, 3, 3), 'conv', None, 'He', False),
                                                ('fc_he', (968, 64), 'fc', 'ReLU', None, False), ('fc_linear', (30, 42), 'fc', None, 'tanh', False),
                                                ('conv_xavier', (16, 8, 3, 3), 'conv', None, 'Xavier', False), ('fc_sigmoid', (20, 10), 'fc', 'sigmoid', None, False),
                                                ('conv_orth', (8, 4, 3, 3), 'conv', 'ReLU', 'He', True)):
        lay = Layer(numpy.random.RandomState(23455))
        d[tag] = lay.getInitVals(shape, mode, act_fn=act, method=method, orthogonal=orth)
        d[tag + '_next'] = lay.rng.uniform(size=3)          # the generator state after the call
    numpy.savez_compressed(os.path.join(HERE, 'init.npz'), **d)
    return d


if __name__ == '__main__':
    s = make_shapes()
    print('shapes:', {k: len(v['layers']) for k, v in s.items()})
    g = make_geometry()
    print('geometry:', sorted(g.keys()))
    print('chunks:', len(make_chunks()))
    c, p = make_crop_and_poses()
    print('crop:', sorted(c.keys())[:6], '... poses:', sorted(p.keys()))
    print('init:', sorted(make_init().keys())[:4])
    e = make_eval()
    print('evaluation: mean %.3f max %.3f' % (e['mean'], e['max']))
    t = make_trainer()
    print('trainer:', t['lr_of_ep'][:3], [(c['n'], c['align'], len(c['padded'])) for c in t['alignData']])
