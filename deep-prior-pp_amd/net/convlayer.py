"""ConvLayer (API of /root/reference/src/net/convlayer.py:39-266): Theano true convolution + bias + activation,
executed by the MFMA GEMM / conv3x3 kernels of libdpp_hip.so."""
import numpy

from hipdp.graph import SharedParam, Var
from net.layer import Layer, floatX
from net.layerparams import LayerParams, conv_output_dim


class ConvLayerParams(LayerParams):
    def __init__(self, inputDim=None, nFilters=None, filterDim=None, activation=None, hasBias=True, filter_shape=None,
                 image_shape=None, outputDim=None, stride=(1, 1), border_mode='valid', init_method=None):
        super(ConvLayerParams, self).__init__(inputDim, outputDim)
        self._nFilters, self._filterDim = nFilters, filterDim
        self._filter_shape, self._image_shape = filter_shape, image_shape
        self._activation, self._hasbias = activation, hasBias
        self._stride = stride
        self._border_mode = 'half' if border_mode == 'same' else border_mode
        self._init_method = init_method
        self._poolsize = getattr(self, '_poolsize', (1, 1))     # ConvPoolLayerParams sets it before chaining up
        self.update()

    def update(self):
        self._filter_shape = (self._nFilters, self._inputDim[1], self._filterDim[0], self._filterDim[1])
        self._image_shape = self._inputDim
        self._outputDim = conv_output_dim(self._inputDim, self._nFilters, self._filterDim, self._stride, self._border_mode,
                                          self._poolsize)

    def _rw(name, refresh=True):          # noqa: N805 - tiny property factory
        def fget(self):
            return getattr(self, name)

        def fset(self, value):
            if name == '_border_mode' and value == 'same':
                value = 'half'
            setattr(self, name, value)
            if refresh:
                self.update()
        return property(fget, fset)

    filter_shape = property(lambda self: self._filter_shape)
    image_shape = property(lambda self: self._image_shape)
    stride = _rw('_stride')
    border_mode = _rw('_border_mode')
    nFilters = _rw('_nFilters')
    filterDim = _rw('_filterDim')
    activation = _rw('_activation', refresh=False)
    hasBias = _rw('_hasbias', refresh=False)
    del _rw

    def getMemoryRequirement(self):
        return (numpy.prod(self.filter_shape) + self.filter_shape[0]) * 4


class ConvLayer(Layer):
    def __init__(self, rng, inputVar, cfgParams, copyLayer=None, layerNum=None):
        super(ConvLayer, self).__init__(rng)
        assert isinstance(cfgParams, ConvLayerParams)
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
               "hasBias {}".format(c.inputDim, c.outputDim, c.filterDim, c.nFilters, c.activation_str, c.stride, c.border_mode,
                                   c.hasBias)
