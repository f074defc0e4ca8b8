"""
hipdp.ckpt_writer -- writes a checkpoint from a raw dump of the flat parameter buffers, in a process of its own.

NetBase.saveAsync (net/netbase.py) hands over (meta file, raw float32 file, target file): the per-epoch `net_last.pkl` of the epoch loop
(/root/reference/src/trainer/nettrainer.py:816-820).  The protocol-2 pickle the reference's checkpoints use encodes every array's bytes
through a latin-1 / UTF-8 round trip under Python 3 -- 0.6 s of interpreter time for the 75 MB of the 128x128 ResNet, all of it holding the
GIL -- so a writer THREAD stalls the training thread (measured: 0.5 s gaps); a writer PROCESS does not.  Imports NumPy and hipdp.layout
only (no torch, no device).  The state dictionary and the pickle call are the ones NetBase.save uses: the bytes are identical.
    python -m hipdp.ckpt_writer <meta.pkl> <raw.f32> <out.pkl>
"""
import gzip
import os
import pickle
import sys

import numpy

from hipdp import layout


def build_state(meta, flat):
    bulk = {}
    for (name, space, off, size, kind, info, shape) in meta['slots']:
        base = flat[space]
        bulk[name] = layout.from_kernel(kind, info, tuple(shape), base[off:off + size])
    state = dict([('class', meta['class']), ('network', meta['network'])])
    for num, names in meta['layers']:
        state['{}-values'.format(num)] = [numpy.array(bulk[n]) for n in names]
    return state


def write(state, filename):
    opener = gzip.open if filename.lower().endswith('.gz') else open
    tmp = filename + '.part'
    with opener(tmp, 'wb') as handle:
        pickle.dump(state, handle, 2)          # protocol 2 = what cPickle wrote; readable by the reference
    os.replace(tmp, filename)                  # a reader never sees a half-written net_last.pkl


def main(argv):
    meta_path, raw_path, out_path = argv
    with open(meta_path, 'rb') as fh:
        meta = pickle.load(fh)
    raw = numpy.fromfile(raw_path, dtype=numpy.float32)
    n_w, n_nt = meta['n_w'], meta['n_nt']
    assert raw.size == n_w + n_nt, (raw.size, n_w, n_nt)
    write(build_state(meta, {'w': raw[:n_w], 'nt': raw[n_w:]}), out_path)
    for f in (meta_path, raw_path):
        try:
            os.remove(f)
        except OSError:
            pass


if __name__ == '__main__':
    main(sys.argv[1:])
