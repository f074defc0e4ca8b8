"""HiddenLayer (API of /root/reference/src/net/hiddenlayer.py:40-169): activation(x . W + b), W of shape
(n_in, n_out)."""
import numpy

from hipdp.graph import SharedParam, Var
from net.layer import Layer, floatX
from net.layerparams import LayerParams


class HiddenLayerParams(LayerParams):
    def __init__(self, inputDim=None, outputDim=None, activation=None, hasBias=True, init_method=None):
        super(HiddenLayerParams, self).__init__(inputDim, outputDim)
        self.activation, self.hasBias, self._init_method = activation, hasBias, init_method

    def getMemoryRequirement(self):
        return ((self.inputDim[1] * self.outputDim[1]) + self.outputDim[1]) * 4


class HiddenLayer(Layer):
    def __init__(self, rng, inputVar, cfgParams, copyLayer=None, layerNum=None):
        super(HiddenLayer, self).__init__(rng)
        assert isinstance(cfgParams, HiddenLayerParams)
        self.inputVar, self.cfgParams, self.layerNum = inputVar, cfgParams, layerNum
        n_in, n_out = cfgParams.inputDim[1], cfgParams.outputDim[1]
        if copyLayer is None:
            w0 = self.getInitVals((n_in, n_out), 'fc', act_fn=cfgParams.activation_str, method=cfgParams._init_method)
            self.W = SharedParam(w0, name='W{}'.format(layerNum))
            if cfgParams.hasBias is True:
                self.b = SharedParam(numpy.zeros((n_out,), dtype=floatX), name='b{}'.format(layerNum))
        else:
            self.W = copyLayer.W
            if cfgParams.hasBias is True:
                self.b = copyLayer.b
        if not cfgParams.hasBias:
            raise NotImplementedError("bias-free hidden layers are not used on the hot path")
        self.output_pre_act = Var('layer', (inputVar,), layer=self, shape=cfgParams.outputDim)
        act = cfgParams.activation
        self.output = self.output_pre_act if act is None else act(self.output_pre_act)
        self.output.name = 'output_layer_{}'.format(self.layerNum)
        self.params = [self.W, self.b]
        self.weights = [self.W]

    def __str__(self):
        c = self.cfgParams
        return "inputDim {}, outputDim {}, activiation {}, hasBias {}".format(c.inputDim, c.outputDim, c.activation_str, c.hasBias)
