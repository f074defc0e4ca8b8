"""DropoutLayer (API of /root/reference/src/net/dropoutlayer.py:39-138): NON-inverted dropout.  Training:
mask * x with mask ~ Bernoulli(1 - p); deterministic: (1 - p) * x."""
import numpy

from hipdp.graph import Var
from net.layer import Layer
from net.layerparams import LayerParams


class DropoutLayerParams(LayerParams):
    def __init__(self, inputDim=None, outputDim=None, p=0.3):
        super(DropoutLayerParams, self).__init__(inputDim=inputDim, outputDim=outputDim)
        self.p = p


class DropoutLayer(Layer):
    def __init__(self, rng, inputVar, cfgParams, copyLayer=None, layerNum=None):
        super(DropoutLayer, self).__init__(rng)
        self.inputVar, self.cfgParams, self.layerNum = inputVar, cfgParams, layerNum
        assert 0. < cfgParams.p < 1.
        self.prob_drop = cfgParams.p
        self.prob_keep = 1.0 - cfgParams.p
        self._flag_on = 1.0
        # the reference seeds its MRG stream from rng.randint(999999) (dropoutlayer.py:98); the draw is consumed so
        # that later layers see the same rng state; masks come from a counter-based device generator instead
        self.mask_seed = int(rng.randint(999999)) if copyLayer is None else copyLayer.mask_seed
        self.output = Var('layer', (inputVar,), layer=self, shape=cfgParams.outputDim)
        self.output.name = 'output_layer_{}'.format(self.layerNum)
        self.output_pre_act = self.output
        self.params = []
        self.weights = []

    def unsetDeterministic(self):
        self._flag_on = 1.0

    def setDeterministic(self):
        self._flag_on = 0.0

    def isDeterministic(self):
        return bool(numpy.allclose(self._flag_on, 0.0))

    def __str__(self):
        c = self.cfgParams
        return "inputDim {}, outputDim {}, p {}".format(c.inputDim, c.outputDim, c.p)
