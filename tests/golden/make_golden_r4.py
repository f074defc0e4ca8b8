#!/usr/bin/env python3
"""
Round-4 additions to the golden fixtures (run in the build container only; rules as in make_golden.py: the reference is IMPORTED
here, the fixtures hold inputs and the reference's OUTPUTS, never its source).

  helpers.json    util/helpers.py shuffle_many_inplace (/root/reference/src/util/helpers.py:87-108): the permutation it applies for
                  a given RandomState seed and length, recorded by shuffling arange(n) -- plus the RandomState position afterwards
                  (the next randint), so a rewrite has to consume the stream the same way.
  dataset.npz     data/dataset.py Dataset.imgStackDepthOnly (/root/reference/src/data/dataset.py:72-111), both normalisations,
                  on seeded synthetic sequences (crops with undefined pixels, per-frame com, two cube sizes), and Dataset.imgSeq.
"""
import json
import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True


def make_helpers():
    import make_golden as G                                    # noqa: F401  (REF on sys.path, py2 shims)
    from util.helpers import shuffle_many_inplace              # the reference's
    out = []
    for seed, n in ((0, 1), (1, 2), (23455, 17), (7, 64), (99, 257)):
        rng = numpy.random.RandomState(seed)
        a = numpy.arange(n)
        b = numpy.arange(n * 3, dtype=numpy.float32).reshape(n, 3)
        shuffle_many_inplace([a, b], random_state=rng)
        assert (b[:, 0] == 3 * a).all()
        out.append(dict(seed=seed, n=n, perm=[int(v) for v in a], next_draw=int(rng.randint(1 << 30))))
    json.dump(out, open(os.path.join(HERE, 'helpers.json'), 'w'))
    return out


def synthetic_sequences():
    """The inputs of dataset.npz, rebuilt by the test from the same seed (basetypes are the product's / the reference's
    namedtuples with the same fields)."""
    from collections import namedtuple
    Frame = namedtuple('Frame', ['dpt', 'gtorig', 'gtcrop', 'T', 'gt3Dorig', 'gt3Dcrop', 'com', 'fileName', 'subSeqName', 'side', 'extraData'])
    Seq = namedtuple('Seq', ['name', 'data', 'config'])
    rng = numpy.random.RandomState(404)
    seqs = []
    for name, n, cube, hw in (('train', 5, (250, 250, 250), (16, 16)), ('test_1', 3, (300, 300, 300), (8, 12))):
        frames = []
        for i in range(n):
            com = numpy.array([rng.uniform(100, 200), rng.uniform(80, 160), rng.uniform(400, 900)], dtype=numpy.float32)   # importers' jointImgTo3D returns float32
            dpt = (com[2] + rng.uniform(-cube[2] / 2., cube[2] / 2., hw)).astype(numpy.float32)
            dpt[rng.rand(*hw) < 0.4] = 0.                      # undefined depth -> far plane
            gt3Dcrop = rng.uniform(-cube[2] / 2., cube[2] / 2., (14, 3)).astype(numpy.float32)
            gtorig = rng.uniform(0, 300, (14, 3)).astype(numpy.float32)
            frames.append(Frame(dpt, gtorig, gtorig * 0.5, numpy.eye(3, dtype=numpy.float32), gt3Dcrop + com, gt3Dcrop, com,
                                '%s_%d.png' % (name, i), '', 'left', {}))
        seqs.append(Seq(name, frames, {'cube': cube}))
    return seqs


def make_dataset():
    import make_golden as G                                    # noqa: F401
    import types
    # data/dataset.py imports the three importers by name only; the module itself (cv2, PIL, cPickle, print statements) is not needed
    if 'data.importers' not in sys.modules:
        stub = types.ModuleType('data.importers')
        stub.NYUImporter = stub.ICVLImporter = stub.MSRA15Importer = lambda basepath: ('importer', basepath)
        sys.modules['data.importers'] = stub
    ds_mod = G.load_py2_module('data.dataset', 'data/dataset.py')
    seqs = synthetic_sequences()
    ds = ds_mod.Dataset(seqs)
    out = {}
    for s in seqs:
        for nz in (False, True):
            d = ds_mod.Dataset(seqs, localCache=False)
            img, lab = d.imgStackDepthOnly(s.name, normZeroOne=nz)
            out['%s_img_%d' % (s.name, nz)] = img
            out['%s_lab_%d' % (s.name, nz)] = lab
    assert ds.imgSeq('test_1') is seqs[1] and ds.imgSeq('nope') == [] and ds.imgStackDepthOnly('nope') == []
    # the cache: a second call returns the same objects
    a = ds.imgStackDepthOnly('train')
    assert ds.imgStackDepthOnly('train')[0] is a[0]
    for cls, default in ((ds_mod.ICVLDataset, '../../data/ICVL/'), (ds_mod.MSRA15Dataset, '../../data/MSRA15/'), (ds_mod.NYUDataset, '../../data/NYU/')):
        assert cls().lmi == ('importer', default) and cls(basepath='/x/').lmi == ('importer', '/x/')
    numpy.savez_compressed(os.path.join(HERE, 'dataset.npz'), **out)
    return out


if __name__ == '__main__':
    make_helpers()
    make_dataset()
    print('wrote helpers.json, dataset.npz')
