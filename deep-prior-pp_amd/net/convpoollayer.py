"""ConvPoolLayer (API of /root/reference/src/net/convpoollayer.py:39-305): convolution -> max-pool
(ignore_border) -> bias AFTER the pool -> activation."""
import numpy

from hipdp.graph import SharedParam, Var
from net.convlayer import ConvLayerParams
from net.layer import Layer, floatX


class ConvPoolLayerParams(ConvLayerParams):
    def __init__(self, inputDim=None, nFilters=None, filterDim=None, activation=None, poolsize=(1, 1), poolType=0,
                 filter_shape=None, image_shape=None, outputDim=None, stride=(1, 1), border_mode='valid', hasBias=True,
                 init_method=None):
        self._poolType = poolType
        self._poolsize = poolsize
        super(ConvPoolLayerParams, self).__init__(inputDim=inputDim, nFilters=nFilters, filterDim=filterDim,
                                                  activation=activation, hasBias=hasBias, filter_shape=filter_shape,
                                                  image_shape=image_shape, outputDim=outputDim, stride=stride,
                                                  border_mode=border_mode, init_method=init_method)

    def update(self):
        super(ConvPoolLayerParams, self).update()
        if self._poolsize[0] == 1 and self._poolsize[1] == 1:
            self._poolType = -1          # no pooling required

    @property
    def poolsize(self):
        return self._poolsize

    @poolsize.setter
    def poolsize(self, value):
        self._poolsize = value
        self.update()

    @property
    def poolType(self):
        return self._poolType


class ConvPoolLayer(Layer):
    def __init__(self, rng, inputVar, cfgParams, copyLayer=None, layerNum=None):
        super(ConvPoolLayer, self).__init__(rng)
        assert isinstance(cfgParams, ConvPoolLayerParams)
        if cfgParams.poolType not in (0, -1):
            raise NotImplementedError("only max pooling / no pooling is on the hot path")
        self.cfgParams, self.layerNum, self.inputVar = cfgParams, layerNum, inputVar
        fs = cfgParams.filter_shape
        assert cfgParams.image_shape[1] == fs[1]
        if copyLayer is not None:
            self.W = copyLayer.W
        else:
            w0 = self.getInitVals(fs, 'conv', act_fn=cfgParams.activation_str, orthogonal=False, method=cfgParams._init_method)
            self.W = SharedParam(w0, name='convW{}'.format(layerNum))
        if cfgParams.hasBias is True:
            self.b = copyLayer.b if copyLayer is not None else SharedParam(numpy.zeros((fs[0],), dtype=floatX),
                                                                           name='convB{}'.format(layerNum))
        self.output_pre_act = Var('layer', (inputVar,), layer=self, shape=cfgParams.outputDim)
        act = cfgParams.activation
        self.output = self.output_pre_act if act is None else act(self.output_pre_act)
        self.output.name = 'output_layer_{}'.format(self.layerNum)
        self.params = [self.W, self.b] if cfgParams.hasBias else [self.W]
        self.weights = [self.W]

    def __str__(self):
        c = self.cfgParams
        return "inputDim {}, outputDim {}, filterDim {}, nFilters {}, activation {}, stride {}, border_mode {}, " \
               "hasBias {}, pool_type {}, pool_size {}".format(c.inputDim, c.outputDim, c.filterDim, c.nFilters, c.activation_str,
                                                               c.stride, c.border_mode, c.hasBias, c.poolType, c.poolsize)
