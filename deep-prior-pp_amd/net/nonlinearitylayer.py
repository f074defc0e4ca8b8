"""NonlinearityLayer (API of /root/reference/src/net/nonlinearitylayer.py:42-133): activation(x).  Never
materialised: the engine folds it into the consumer's operand prologue."""
from hipdp.graph import Var
from net.layer import Layer
from net.layerparams import LayerParams


class NonlinearityLayerParams(LayerParams):
    def __init__(self, inputDim=None, outputDim=None, activation=None):
        super(NonlinearityLayerParams, self).__init__(inputDim, outputDim)
        self._outputDim = self._inputDim
        self.activation = activation

    def getMemoryRequirement(self):
        return 0


class NonlinearityLayer(Layer):
    def __init__(self, rng, inputVar, cfgParams, copyLayer=None, layerNum=None):
        super(NonlinearityLayer, self).__init__(rng)
        assert isinstance(cfgParams, NonlinearityLayerParams)
        self.inputVar, self.cfgParams, self.layerNum = inputVar, cfgParams, layerNum
        self.output_pre_act = inputVar
        # (activation None = the identity: the engine reads the function off cfgParams when it folds the node into its consumer)
        self.output = Var('layer', (inputVar,), layer=self, shape=cfgParams.outputDim)
        self.output.name = 'output_layer_{}'.format(self.layerNum)
        self.params = []
        self.weights = []

    def __str__(self):
        c = self.cfgParams
        return "inputDim {}, outputDim {}, activation {}".format(c.inputDim, c.outputDim, c.activation_str)
