"""PoolLayer parametrisation (API of /root/reference/src/net/poollayer.py:39-157).  Stand-alone pooling layers are
not instantiated by the hot-path nets (pooling is fused in ConvPoolLayer); only the Params class is kept."""
from net.layerparams import LayerParams


class PoolLayerParams(LayerParams):
    def __init__(self, inputDim=None, poolsize=None, redDim=None, outputDim=None, activation=None, poolType=0):
        super(PoolLayerParams, self).__init__(inputDim, outputDim)
        self._poolsize, self._redDim, self._activation, self._poolType = poolsize, redDim, activation, poolType
        self.update()

    poolsize = property(lambda self: self._poolsize)
    activation = property(lambda self: self._activation)
    poolType = property(lambda self: self._poolType)

    def update(self):
        i = self._inputDim
        ch = self._redDim * i[1] if self._redDim is not None else i[1]
        self._outputDim = (i[0], ch, i[2] // self._poolsize[0], i[3] // self._poolsize[1])
        if self._poolsize[0] == 1 and self._poolsize[1] == 1:
            self._poolType = -1


class PoolLayer(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("stand-alone PoolLayer is outside the DeepPrior++ hot path (SURVEY.md section 8)")
