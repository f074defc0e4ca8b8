"""
Dataset (API of /root/reference/src/data/dataset.py:38-185): stacks the cropped depth frames of a sequence into the
NCHW float32 tensor the nets consume -- background (0) set to the far plane, then (d - com_z) / (cube_z / 2) -- and the
joint labels / (cube_z / 2).
"""
import numpy


class Dataset(object):
    def __init__(self, imgSeqs=None, localCache=True):
        self.localCache = localCache
        self._imgSeqs = [] if imgSeqs is None else imgSeqs
        self._imgStacks = {}
        self._labelStacks = {}

    @property
    def imgSeqs(self):
        return self._imgSeqs

    @imgSeqs.setter
    def imgSeqs(self, value):
        self._imgSeqs = value
        self._imgStacks = {}

    def imgSeq(self, seqName):
        for seq in self._imgSeqs:
            if seq.name == seqName:
                return seq
        return []

    def imgStackDepthOnly(self, seqName, normZeroOne=False):
        imgSeq = None
        for seq in self._imgSeqs:
            if seq.name == seqName:
                imgSeq = seq
                break
        if imgSeq is None:
            return []
        if seqName not in self._imgStacks:
            n = len(imgSeq.data)
            h, w = numpy.asarray(imgSeq.data[0].dpt).shape
            j, d = numpy.asarray(imgSeq.data[0].gtorig).shape
            imgStack = numpy.zeros((n, 1, h, w), dtype='float32')
            labelStack = numpy.zeros((n, j, d), dtype='float32')
            cz = imgSeq.config['cube'][2]
            for i in range(n):
                imgD = numpy.asarray(imgSeq.data[i].dpt.copy(), 'float32')
                imgD[imgD == 0] = imgSeq.data[i].com[2] + (cz / 2.)
                if normZeroOne:
                    imgD -= (imgSeq.data[i].com[2] - (cz / 2.))
                    imgD /= cz
                else:
                    imgD -= imgSeq.data[i].com[2]
                    imgD /= (cz / 2.)
                imgStack[i] = imgD
                labelStack[i] = numpy.asarray(imgSeq.data[i].gt3Dcrop, dtype='float32') / (cz / 2.)
            if self.localCache:
                self._imgStacks[seqName] = imgStack
                self._labelStacks[seqName] = labelStack
            else:
                return imgStack, labelStack
        return self._imgStacks[seqName], self._labelStacks[seqName]


class ICVLDataset(Dataset):
    pass


class MSRA15Dataset(Dataset):
    pass


class NYUDataset(Dataset):
    pass
