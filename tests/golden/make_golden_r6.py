#!/usr/bin/env python3
"""
Round-6 additions to the golden fixtures (run in the build container only; make_golden.py has the rules: the reference is IMPORTED
here, the fixtures hold inputs and the reference's OUTPUTS, never its source).

  rot3d.npz   the reference's own rotatePoints3D / getRotationMatrix (/root/reference/src/data/transformations.py:105-155) and
              HandDetector.sampleRandomPoses(..., rot3D=True) (/root/reference/src/util/handdetector.py:805-909) executed here.

getRotationMatrix imports `transforms3d.euler.euler2mat` at call time.  transforms3d (an un-vendored third-party dependency of the
reference, https://github.com/matthew-brett/transforms3d, 0.3.x at the reference's date) is not in this image, so a STAND-IN module is
registered for that one function below: its published algorithm in its general form (the axis tables and Shoemake's products for any
axes string), checked here against elementary rotation matrices.  It is a stand-in, not the package; everything else that runs is the
reference's code.
"""
import math
import os
import sys
import types

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

_NEXT_AXIS = [1, 2, 0, 1]
_AXES2TUPLE = {
    'sxyz': (0, 0, 0, 0), 'sxyx': (0, 0, 1, 0), 'sxzy': (0, 1, 0, 0), 'sxzx': (0, 1, 1, 0), 'syzx': (1, 0, 0, 0), 'syzy': (1, 0, 1, 0),
    'syxz': (1, 1, 0, 0), 'syxy': (1, 1, 1, 0), 'szxy': (2, 0, 0, 0), 'szxz': (2, 0, 1, 0), 'szyx': (2, 1, 0, 0), 'szyz': (2, 1, 1, 0),
    'rzyx': (0, 0, 0, 1), 'rxyx': (0, 0, 1, 1), 'ryzx': (0, 1, 0, 1), 'rxzx': (0, 1, 1, 1), 'rxzy': (1, 0, 0, 1), 'ryzy': (1, 0, 1, 1),
    'rzxy': (1, 1, 0, 1), 'ryxy': (1, 1, 1, 1), 'ryxz': (2, 0, 0, 1), 'rzxz': (2, 0, 1, 1), 'rxyz': (2, 1, 0, 1), 'rzyz': (2, 1, 1, 1)}


def euler2mat_standin(ai, aj, ak, axes='sxyz'):
    firstaxis, parity, repetition, frame = _AXES2TUPLE[axes]
    i = firstaxis
    j = _NEXT_AXIS[i + parity]
    k = _NEXT_AXIS[i - parity + 1]
    if frame:
        ai, ak = ak, ai
    if parity:
        ai, aj, ak = -ai, -aj, -ak
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs = ci * ck, ci * sk
    sc, ss = si * ck, si * sk
    M = numpy.eye(3)
    if repetition:
        M[i, i] = cj
        M[i, j] = sj * si
        M[i, k] = sj * ci
        M[j, i] = sj * sk
        M[j, j] = -cj * ss + cc
        M[j, k] = -cj * cs - sc
        M[k, i] = -sj * ck
        M[k, j] = cj * sc + cs
        M[k, k] = cj * cc - ss
    else:
        M[i, i] = cj * ck
        M[i, j] = sj * sc - cs
        M[i, k] = sj * cc + ss
        M[j, i] = cj * sk
        M[j, j] = sj * ss + cc
        M[j, k] = sj * cs - sc
        M[k, i] = -sj
        M[k, j] = cj * si
        M[k, k] = cj * ci
    return M


def _elementary(axis, t):
    c, s = math.cos(t), math.sin(t)
    return {'x': numpy.array([[1, 0, 0], [0, c, -s], [0, s, c]]), 'y': numpy.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            'z': numpy.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def check_standin():
    rng = numpy.random.RandomState(4)
    for _ in range(20):
        a, b, c = rng.uniform(-math.pi, math.pi, 3)
        # static frame: rotations about the fixed x, then y, then z; rotating frame 'rxyz': about x, the new y, the new z
        assert numpy.allclose(euler2mat_standin(a, b, c, 'sxyz'), _elementary('z', c) @ _elementary('y', b) @ _elementary('x', a), atol=1e-14)
        assert numpy.allclose(euler2mat_standin(a, b, c, 'rxyz'), _elementary('x', a) @ _elementary('y', b) @ _elementary('z', c), atol=1e-14)
        assert numpy.allclose(euler2mat_standin(a, b, c, 'szxz'), _elementary('z', c) @ _elementary('x', b) @ _elementary('z', a), atol=1e-14)


def make_rot3d():
    import make_golden as G                                    # placeholder modules, numpy.cast shim, REF on sys.path
    from oracle import augment as A
    check_standin()
    t3 = types.ModuleType('transforms3d')
    t3e = types.ModuleType('transforms3d.euler')
    t3e.euler2mat = euler2mat_standin
    t3.euler = t3e
    sys.modules['transforms3d'], sys.modules['transforms3d.euler'] = t3, t3e
    tr = G.load_py2_module('data.transformations', 'data/transformations.py')
    hd_mod = G.load_py2_module('util.handdetector', 'util/handdetector.py')
    imp = sys.modules.get('data.importers') or G.load_py2_module('data.importers', 'data/importers.py')
    rng = numpy.random.RandomState(61)
    d = {}
    pts = rng.uniform(-120, 120, (24, 16, 3)).astype('float32') + numpy.float32([0, 0, 600])
    ctr = (rng.uniform(-40, 40, (24, 3)) + [0, 0, 600]).astype('float32')
    ang = rng.uniform(-180, 180, (24, 3))
    d['pts'], d['ctr'], d['ang'] = pts, ctr, ang
    d['R'] = numpy.stack([tr.getRotationMatrix(*ang[i]) for i in range(24)])
    d['out'] = numpy.stack([tr.rotatePoints3D(pts[i], ctr[i], *ang[i]) for i in range(24)])
    d['out_point64'] = numpy.stack([tr.rotatePoint3D(pts[i, 0].astype('float64'), ctr[i].astype('float64'), *ang[i]) for i in range(24)])
    Ms = rng.normal(0, 1, (8, 4, 4))
    d['tp3_M'], d['tp3_out'] = Ms, numpy.stack([tr.transformPoint3D(pts[i, 0], Ms[i]) for i in range(8)])
    d['gtm_args'] = numpy.array([[64., 60., 0.3, 3., -2., 1.1], [10., 90., -2.0, -5., 7., 0.8]])
    d['gtm_out'] = numpy.stack([tr.getTransformationMatrix(a[0:2], a[2], a[3:5], a[5]) for a in d['gtm_args']])
    for nm, cls, J, args in (('icvl', imp.ICVLImporter, 16, (241.42, 241.42, 160., 120.)), ('nyu', imp.NYUImporter, 14, (588.03, 587.07, 320., 240.))):
        o = cls.__new__(cls)
        imp.DepthImporter.__init__(o, *args)
        camx = A.Camera.icvl() if nm == 'icvl' else A.Camera.nyu()
        _, c3, cubes, _, gts = A.synthetic_augment_inputs(numpy.random.RandomState(5), 12, camx, cube=(250., 250., 250.), joints=J)
        d['%s_com' % nm], d['%s_cube' % nm], d['%s_gt' % nm] = c3, cubes, gts
        for tag, modes in (('main', ['com', 'rot', 'none']), ('all', ['com', 'rot', 'sc', 'none', 'rot+com', 'rot+com+sc'])):
            poses, ncom, ncube, rot = hd_mod.HandDetector.sampleRandomPoses(o, numpy.random.RandomState(9), gts, c3, cubes, 300, modes,
                                                                            retall=True, rot3D=True)
            d['%s_%s' % (nm, tag)], d['%s_%s_com' % (nm, tag)], d['%s_%s_cube' % (nm, tag)] = poses, ncom, ncube
    numpy.savez_compressed(os.path.join(HERE, 'rot3d.npz'), **d)
    return d


def make_baseline():
    """baseline.npz: the importers' loadBaseline / loadBaseline2D (/root/reference/src/data/importers.py:422-484, 1079-1175) run by the
    reference's own code on small synthetic result files written to a temporary directory; the fixture keeps the numbers that went into
    those files (text rows, the .mat arrays, the depth frames) and the reference's outputs."""
    import tempfile
    import scipy.io
    from PIL import Image
    import make_golden as G
    imp = sys.modules.get('data.importers') or G.load_py2_module('data.importers', 'data/importers.py')
    rng = numpy.random.RandomState(17)
    d = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ICVL: text, 16 joints, optional leading file name, a blank line in between
        o = imp.ICVLImporter.__new__(imp.ICVLImporter)
        imp.DepthImporter.__init__(o, 241.42, 241.42, 160., 120.)
        o.numJoints = 16
        rows = numpy.round(numpy.concatenate([rng.uniform(10, 300, (5, 16, 2)), rng.uniform(200, 900, (5, 16, 1))], 2), 4)
        d['icvl_rows'] = rows
        for first in (False, True):
            for blank in (True, False):        # (the 2-D reader of the reference does not skip blank lines)
                fn = os.path.join(tmp, 'lrf_%d_%d.txt' % (first, blank))
                with open(fn, 'w') as fh:
                    for k, r in enumerate(rows):
                        fh.write((('image_%04d.png ' % k) if first else '') + ' '.join('%.4f' % v for v in r.reshape(-1)) + '\n')
                        if k == 2 and blank:
                            fh.write('\n')
                if blank:
                    d['icvl_3d_%d' % first] = numpy.stack(o.loadBaseline(fn, firstName=first))
                else:
                    d['icvl_2d_%d' % first] = numpy.stack(o.loadBaseline2D(fn, firstName=first))
        # NYU: text form
        n = imp.NYUImporter.__new__(imp.NYUImporter)
        imp.DepthImporter.__init__(n, 588.03, 587.07, 320., 240.)
        rows = numpy.round(numpy.concatenate([rng.uniform(10, 600, (4, 14, 2)), rng.uniform(400, 1200, (4, 14, 1))], 2), 4)
        d['nyu_rows'] = rows
        fn = os.path.join(tmp, 'pred.txt')
        with open(fn, 'w') as fh:
            for r in rows:
                fh.write(' '.join('%.4f' % v for v in r.reshape(-1)) + '\n')
        # (Python 2 integer division at importers.py:1128; under Python 3 the reference would size the array with a float)
        real_zeros = imp.np.zeros
        imp.np.zeros = lambda shape, *a, **k: real_zeros(tuple(int(v) for v in shape) if isinstance(shape, tuple) else shape, *a, **k)
        try:
            d['nyu_text_3d'] = numpy.stack(n.loadBaseline(fn))
        finally:
            imp.np.zeros = real_zeros
        # NYU: the .mat of 2-D predictions + depth frames + ground truth (frame 3's image is missing)
        F, S, Jn = 4, 20, 14
        H, W = 48, 64
        uvc = numpy.zeros((F, S, 3))
        for f in range(F):
            used = sorted(rng.choice(S, Jn, replace=False))
            uvc[f, used] = numpy.stack([rng.uniform(1, W - 2, Jn), rng.uniform(1, H - 2, Jn), rng.uniform(0.1, 1, Jn)], 1)
        names = numpy.empty((1, Jn), dtype=object)
        for j in range(Jn):
            names[0, j] = 'J%d' % j
        pj = numpy.empty((1,), dtype=object)
        pj[0] = uvc
        cn = numpy.empty((1,), dtype=object)
        cn[0] = names[0]
        scipy.io.savemat(os.path.join(tmp, 'test_predictions.mat'), {'conv_joint_names': cn.reshape(1, 1), 'pred_joint_uvconf': pj.reshape(1, 1)})
        chk = scipy.io.loadmat(os.path.join(tmp, 'test_predictions.mat'))
        assert chk['conv_joint_names'][0].shape[0] in (1, Jn)
        # the reference reads mat['conv_joint_names'][0] (length = joints) and mat['pred_joint_uvconf'][0] (frames x slots x 3)
        scipy.io.savemat(os.path.join(tmp, 'test_predictions.mat'), {'conv_joint_names': names, 'pred_joint_uvconf': uvc[None]})
        chk = scipy.io.loadmat(os.path.join(tmp, 'test_predictions.mat'))
        assert chk['conv_joint_names'][0].shape[0] == Jn and chk['pred_joint_uvconf'][0].shape == (F, S, 3)
        depth = rng.randint(500, 900, (F, H, W)).astype(numpy.int32)
        depth[:, ::7, ::5] = 2001                                  # background hits: replaced by the ground-truth depth
        for f in range(F):
            if f == 2:
                continue
            rgb = numpy.zeros((H, W, 3), numpy.uint8)
            rgb[..., 1], rgb[..., 2] = depth[f] >> 8, depth[f] & 255
            Image.fromarray(rgb).save(os.path.join(tmp, 'depth_1_%07d.png' % (f + 1)))
        gt = numpy.concatenate([rng.uniform(1, 60, (F, Jn, 2)), rng.uniform(650, 750, (F, Jn, 1))], 2)
        d['nyu_uvc'], d['nyu_depth'], d['nyu_gt'] = uvc, depth, gt
        d['nyu_mat_3d'] = numpy.stack(n.loadBaseline(os.path.join(tmp, 'test_predictions.mat'), gt))
        d['nyu_mat_2d'] = numpy.stack(n.loadBaseline2D(os.path.join(tmp, 'test_predictions.mat')))
    numpy.savez_compressed(os.path.join(HERE, 'baseline.npz'), **d)
    return d


if __name__ == '__main__':
    out = make_rot3d()
    print('rot3d.npz:', {k: v.shape for k, v in out.items()})
    out = make_baseline()
    print('baseline.npz:', {k: v.shape for k, v in out.items()})
