#!/usr/bin/env python3
"""
Round-2 additions to the golden fixtures (run in the build container only; see make_golden.py for the rules: the reference is
IMPORTED here, the fixtures hold inputs and the reference's OUTPUTS, never its source).

  transform.npz   HandDetector.comToTransform (/root/reference/src/util/handdetector.py:228-258) executed by the reference's own
                  code.  The module is Python 2: its two crop-size expressions `hb * dsize[0] / wb` and `wb * dsize[1] / hb`
                  (lines 246, 249) divide INTEGERS, i.e. floor-divide under Python 2.  The in-memory lib2to3 pass does not
                  change `/`, so exactly those two sites are rewritten to `//` before the source is exec'd -- the only edit --
                  and the function then computes under Python 3 what it computed under Python 2.
  net_py2.pkl     A checkpoint in the byte layout Python 2's cPickle.dump(state, f, protocol=2) produced for
                  NetBase.save (/root/reference/src/net/netbase.py:405-424): str keys / values as SHORT_BINSTRING / BINSTRING
                  (not BINUNICODE), NumPy arrays as numpy.core.multiarray._reconstruct(ndarray, (0,), 'b') + the 5-tuple state
                  (1, shape, dtype, False, raw bytes as a str) with the dtype reduced the Python-2 way.  Python 3 cannot emit that
                  layout with pickle.dump, so the opcode stream is assembled by hand below (net_py2.json holds the expected
                  content).  It is what NetBase.load has to read with encoding='latin1'.
"""
import json
import os
import struct
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True


def make_transform():
    import make_golden as G                                    # placeholder modules, numpy.cast shim, REF on sys.path
    from lib2to3 import refactor
    import types
    rel = 'util/handdetector.py'
    src = open(os.path.join(G.REF, rel)).read() + '\n'
    sites = [("sz = (dsize[0], hb * dsize[0] / wb)", "sz = (dsize[0], hb * dsize[0] // wb)"),
             ("sz = (wb * dsize[1] / hb, dsize[1])", "sz = (wb * dsize[1] // hb, dsize[1])")]
    for old, new in sites:
        # comToTransform and cropArea3D / the resize helper carry the same expression: every occurrence is an int / int
        assert src.count(old) >= 1, old
        src = src.replace(old, new)
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    mod = types.ModuleType('util.handdetector_py2div')
    exec(compile(str(tool.refactor_string(src, rel)), os.path.join(G.REF, rel), 'exec'), mod.__dict__)
    rng = numpy.random.RandomState(31)
    d = dict(com=[], size=[], fx=[], dsize=[], M=[], bounds=[])
    frame = numpy.full((240, 320), 500., numpy.float32)
    for cam_fx, cam_fy in ((241.42, 241.42), (588.03, 587.07)):
        hd = mod.HandDetector(frame.copy(), cam_fx, cam_fy)
        for k in range(40):
            com = numpy.array([rng.uniform(20, 300), rng.uniform(20, 220), rng.uniform(250, 900)])
            size = tuple(float(v) for v in rng.choice([150., 200., 250., 300.], 1)) * 3
            if k % 5 == 0:
                size = (float(rng.choice([200., 250.])), float(rng.choice([230., 300.])), 250.)        # non-cubic: the wb != hb branches
            ds = (128, 128) if k % 7 else (96, 128)
            d['com'].append(com), d['size'].append(size), d['fx'].append((cam_fx, cam_fy)), d['dsize'].append(ds)
            d['M'].append(hd.comToTransform(com, size, ds))
            d['bounds'].append(hd.comToBounds(com, size))
    out = {k: numpy.asarray(v, numpy.float64) for k, v in d.items()}
    numpy.savez_compressed(os.path.join(HERE, 'transform.npz'), **out)
    return out


# ---- a Python-2 cPickle protocol-2 stream, opcode by opcode ---------------------------------------------------------------
def _str(s):
    b = s if isinstance(s, bytes) else s.encode('latin1')
    return (b'U' + struct.pack('<B', len(b)) + b) if len(b) < 256 else (b'T' + struct.pack('<i', len(b)) + b)


def _int(i):
    if 0 <= i < 256:
        return b'K' + struct.pack('<B', i)
    if 0 <= i < 65536:
        return b'M' + struct.pack('<H', i)
    return b'J' + struct.pack('<i', i)


def _tuple(items):
    if len(items) <= 3:
        return b''.join(items) + {0: b')', 1: b'\x85', 2: b'\x86', 3: b'\x87'}[len(items)]
    return b'(' + b''.join(items) + b't'


def _ndarray(a):
    a = numpy.ascontiguousarray(a)
    code = {'float32': 'f4', 'float64': 'f8', 'int32': 'i4'}[a.dtype.name]
    dtype = b'cnumpy\ndtype\n' + _tuple([_str(code), _int(0), _int(1)]) + b'R' + \
        _tuple([_int(3), _str('<'), b'N', b'N', b'N', b'J\xff\xff\xff\xff', b'J\xff\xff\xff\xff', _int(0)]) + b'b'
    head = b'cnumpy.core.multiarray\n_reconstruct\n' + _tuple([b'cnumpy\nndarray\n', _tuple([_int(0)]), _str('b')]) + b'R'
    state = _tuple([_int(1), _tuple([_int(int(s)) for s in a.shape]), dtype, b'\x89', _str(a.tobytes())])
    return head + state + b'b'


def make_py2_pickle():
    rng = numpy.random.RandomState(41)
    # the state NetBase.save writes for a two-layer net: conv-pool (W, b) and a BatchNorm (beta, gamma | mean, inv_std)
    arrays = {'0-values': [rng.normal(0, 0.2, (8, 1, 5, 5)).astype('float32'), rng.normal(0, 0.1, 8).astype('float32')],
              '1-values': [rng.normal(0, 0.1, 8).astype('float32'), rng.uniform(0.5, 1.5, 8).astype('float32'),
                           rng.normal(0, 0.3, 8).astype('float32'), rng.uniform(0.5, 2.0, 8).astype('float32')]}
    network = 'Network:\n  ConvPoolLayer 0 ...\n  BatchNormLayer 1 ...'
    body = b'\x80\x02}(' + _str('class') + _str('PoseRegNet') + _str('network') + _str(network)
    for key in ('0-values', '1-values'):
        body += _str(key) + b']' + b'(' + b''.join(_ndarray(a) for a in arrays[key]) + b'e'
    body += b'u.'
    with open(os.path.join(HERE, 'net_py2.pkl'), 'wb') as f:
        f.write(body)
    json.dump({'class': 'PoseRegNet', 'network': network,
               'arrays': {k: [dict(shape=list(a.shape), dtype=a.dtype.name, data=a.astype('float64').ravel().tolist()) for a in v] for k, v in arrays.items()}},
              open(os.path.join(HERE, 'net_py2.json'), 'w'))
    # self-check: Python 3 reads it the way NetBase.load does; without latin1 the str values would be rejected / bytes
    import pickle
    import pickletools
    pickletools.dis(body, out=open(os.devnull, 'w'))            # a well-formed opcode stream
    got = pickle.loads(body, encoding='latin1')
    assert got['class'] == 'PoseRegNet' and got['network'] == network
    for k, v in arrays.items():
        assert all(numpy.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(got[k], v))
    ops = set(op.name for op, _, _ in pickletools.genops(body))
    assert 'SHORT_BINSTRING' in ops and 'BINUNICODE' not in ops and 'SHORT_BINBYTES' not in ops
    return got


if __name__ == '__main__':
    t = make_transform()
    print('transform:', t['M'].shape, 'non-square cases', int((t['size'][:, 0] != t['size'][:, 1]).sum()))
    p = make_py2_pickle()
    print('net_py2.pkl:', sorted(p.keys()))
