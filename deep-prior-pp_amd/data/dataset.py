"""
Training stacks of cropped sequences (class / method names of /root/reference/src/data/dataset.py:38-185).

`imgStackDepthOnly(name)` turns a `NamedImgSequence` of cropped frames into the `(N, 1, H, W)` float32 tensor the nets
consume and the `(N, J, 3)` labels: undefined depth (0) becomes the far plane of the frame's cube, then
`(d - com_z) / (cube_z / 2)` (or the [0, 1] form), labels `gt3Dcrop / (cube_z / 2)` (dataset.py:91-106).  The whole
sequence is normalised as one float32 array expression (the device form of the same arithmetic is `crop_frames(...,
normalize=True)`, csrc/augment.hip, which the importers' cascade path uses).
Pinned by tests/golden/dataset.npz (the reference's own output on seeded sequences).
"""
import numpy


def _normalised_stack(seq, zero_one):
    frames = seq.data
    half = seq.config['cube'][2] / 2.
    depth = numpy.stack([numpy.asarray(f.dpt, dtype=numpy.float32) for f in frames])
    # float32 throughout, as with the importers' float32 `com` (jointImgTo3D): one rounding per operation
    com_z = numpy.asarray([f.com[2] for f in frames], dtype=numpy.float32).reshape(-1, 1, 1)
    far = (com_z.astype(numpy.float64) + half).astype(numpy.float32)
    depth = numpy.where(depth == 0, far, depth)
    if zero_one:
        depth -= (com_z.astype(numpy.float64) - half).astype(numpy.float32)
        depth /= numpy.float32(2. * half)
    else:
        depth -= com_z
        depth /= numpy.float32(half)
    labels = numpy.stack([numpy.asarray(f.gt3Dcrop, dtype=numpy.float32) for f in frames]) / numpy.float32(half)
    return depth[:, None], labels


class Dataset(object):
    def __init__(self, imgSeqs=None, localCache=True):
        self.localCache = localCache
        self._imgSeqs = list(imgSeqs) if imgSeqs is not None and not isinstance(imgSeqs, list) else (imgSeqs or [])
        self._imgStacks, self._labelStacks = {}, {}

    imgSeqs = property(lambda self: self._imgSeqs)

    @imgSeqs.setter
    def imgSeqs(self, value):
        self._imgSeqs = value
        self._imgStacks = {}

    def imgSeq(self, seqName):
        return next((s for s in self._imgSeqs if s.name == seqName), [])

    def imgStackDepthOnly(self, seqName, normZeroOne=False):
        seq = self.imgSeq(seqName)
        if seq == []:
            return []
        if seqName in self._imgStacks:
            return self._imgStacks[seqName], self._labelStacks[seqName]
        stacks = _normalised_stack(seq, normZeroOne)
        if self.localCache:
            self._imgStacks[seqName], self._labelStacks[seqName] = stacks
        return stacks


def _with_importer(importer_name, default_basepath):
    """The per-dataset subclasses differ only in the importer they hold as `.lmi` (dataset.py:137-185)."""
    class _Named(Dataset):
        def __init__(self, imgSeqs=None, basepath=None, localCache=True):
            Dataset.__init__(self, imgSeqs, localCache)
            from data import importers
            self.lmi = getattr(importers, importer_name)(default_basepath if basepath is None else basepath)
    return _Named


ICVLDataset = type('ICVLDataset', (_with_importer('ICVLImporter', '../../data/ICVL/'),), {})
MSRA15Dataset = type('MSRA15Dataset', (_with_importer('MSRA15Importer', '../../data/MSRA15/'),), {})
NYUDataset = type('NYUDataset', (_with_importer('NYUImporter', '../../data/NYU/'),), {})
