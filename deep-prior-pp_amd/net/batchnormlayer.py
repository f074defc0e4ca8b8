"""BatchNormLayer (API of /root/reference/src/net/batchnormlayer.py:40-222).  Parameters: beta, gamma (trained),
mean, inv_std (running statistics, not trained).  flag_on = 1: batch statistics (training), 0: stored ones."""
import numpy

from hipdp.graph import SharedParam, Var
from net.layer import Layer, floatX
from net.layerparams import LayerParams


class BatchNormLayerParams(LayerParams):
    def __init__(self, inputDim=None, outputDim=None, epsilon=1e-4, alpha=0.1, mode='low_mem', learn_beta=True,
                 learn_gamma=True):
        super(BatchNormLayerParams, self).__init__(inputDim, outputDim)
        self._learn_beta, self._learn_gamma = learn_beta, learn_gamma
        self.epsilon, self.alpha, self.mode = epsilon, alpha, mode
        self._outputDim = self._inputDim


class BatchNormLayer(Layer):
    def __init__(self, rng, inputVar, cfgParams, copyLayer=None, layerNum=None):
        super(BatchNormLayer, self).__init__(rng)
        self.cfgParams, self.layerNum, self.inputVar = cfgParams, layerNum, inputVar
        if not (cfgParams._learn_beta and cfgParams._learn_gamma):
            raise NotImplementedError("fixed beta/gamma are not used on the hot path")
        inputDim = cfgParams.inputDim
        shape = (inputDim[1],)            # statistics over every axis but the channel axis
        self._flag_on = 1.0
        if copyLayer is not None:
            assert copyLayer.beta.get_value().shape == shape and copyLayer.gamma.get_value().shape == shape
            self.beta, self.gamma = copyLayer.beta, copyLayer.gamma
        else:
            self.beta = SharedParam(numpy.zeros(shape, dtype=floatX), name='beta{}'.format(layerNum))
            self.gamma = SharedParam(numpy.ones(shape, dtype=floatX), name='gamma{}'.format(layerNum))
        self.mean = SharedParam(numpy.zeros(shape, dtype=floatX), name='mean{}'.format(layerNum))
        self.inv_std = SharedParam(numpy.ones(shape, dtype=floatX), name='inv_std{}'.format(layerNum))
        if copyLayer is not None:
            self.mean.set_value(copyLayer.mean.get_value())
            self.inv_std.set_value(copyLayer.inv_std.get_value())
        self.weights = []
        self.params = [self.beta, self.gamma]
        self.params_nontrained = [self.mean, self.inv_std]
        self.output = Var('layer', (inputVar,), layer=self, shape=cfgParams.outputDim)
        self.output.name = 'output_layer_{}'.format(self.layerNum)
        self.output_pre_act = self.output

    def unsetDeterministic(self):
        self._flag_on = 1.0

    def setDeterministic(self):
        self._flag_on = 0.0

    def isDeterministic(self):
        return bool(numpy.allclose(self._flag_on, 0.0))

    def __str__(self):
        return "epsilon {}, alpha {}".format(self.cfgParams.epsilon, self.cfgParams.alpha)
