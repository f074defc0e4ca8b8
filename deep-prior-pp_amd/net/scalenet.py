"""ScaleNet / ScaleNetParams (API of /root/reference/src/net/scalenet.py:33-195): the multi-scale CoM-refinement regressor of
main_nyu_com_refine.py -- three conv-pool towers on the crop and on its 1/2 and 1/4 centre crops, their flattened outputs
concatenated into FC 1024 - dropout - FC 1024 - dropout - FC (numJoints * nDims)."""
from hipdp.graph import concatenate, tensor4
from net.convpoollayer import ConvPoolLayer, ConvPoolLayerParams
from net.dropoutlayer import DropoutLayer, DropoutLayerParams
from net.hiddenlayer import HiddenLayer, HiddenLayerParams
from net.netbase import NetBase, NetBaseParams
from util.theano_helpers import ReLU

_LAYER_CLASSES = {'ConvPool': ConvPoolLayer, 'Hidden': HiddenLayer, 'Dropout': DropoutLayer}


class ScaleNetParams(NetBaseParams):
    def __init__(self, type=0, nChan=1, wIn=128, hIn=128, batchSize=128, numJoints=16, nDims=3, resizeFactor=2, shared_conv=False):
        super(ScaleNetParams, self).__init__()
        self.batch_size = batchSize
        self.numJoints = numJoints
        self.nDims = nDims
        self.shared_conv = shared_conv
        if type != 1:
            raise NotImplementedError("not implemented")
        self.type = type
        self.numInputs = 3
        self.inpConv = 3
        self.inputDim = [(batchSize, nChan, hIn, wIn), (batchSize, nChan, hIn // resizeFactor, wIn // resizeFactor),
                         (batchSize, nChan, hIn // resizeFactor ** 2, wIn // resizeFactor ** 2)]
        L = self.layers
        towers = ((((5, 5), (4, 4)), ((5, 5), (2, 2)), ((3, 3), (1, 1))),        # scalenet.py:55-71
                  (((5, 5), (2, 2)), ((5, 5), (2, 2)), ((3, 3), (1, 1))),        # :73-89
                  (((5, 5), (2, 2)), ((5, 5), (1, 1)), ((3, 3), (1, 1))))        # :91-107
        for t, tower in enumerate(towers):
            dim = self.inputDim[t]
            for fd, pool in tower:
                L.append(ConvPoolLayerParams(inputDim=dim, nFilters=8, filterDim=fd, poolsize=pool, activation=ReLU))
                dim = L[-1].outputDim
        lout = 0
        for j in range(self.numInputs):
            od = L[(j + 1) * self.inpConv - 1].outputDim
            lout += od[1] * od[2] * od[3]
        L.append(HiddenLayerParams(inputDim=(batchSize, lout), outputDim=(batchSize, 1024), activation=ReLU))
        L.append(DropoutLayerParams(inputDim=L[-1].outputDim, outputDim=L[-1].outputDim))
        L.append(HiddenLayerParams(inputDim=L[-1].outputDim, outputDim=(batchSize, 1024), activation=ReLU))
        L.append(DropoutLayerParams(inputDim=L[-1].outputDim, outputDim=L[-1].outputDim))
        L.append(HiddenLayerParams(inputDim=L[-1].outputDim, outputDim=(batchSize, numJoints * nDims), activation=None))
        self.outputDim = L[-1].outputDim


class ScaleNet(NetBase):
    def __init__(self, rng, inputVar=None, cfgParams=None, twin=None):
        if cfgParams is None:
            raise Exception("Cannot create a Net without config parameters (ie. cfgParams==None)")
        if inputVar is not None:
            raise Exception("Do not give inputVar, created inline")
        # twin: a second net instance on ANOTHER net's parameters (scalenet.py:178: every layer is built with copyLayer =
        # twin.layers[i]) -- e.g. the same regressor at another batch size.  Both instances then run on one device parameter store
        # (hipdp.engine.get_store follows `_twin`).
        self._twin = twin
        self._params_filter = []
        self._weights_filter = []
        self.inputVar = [tensor4('x{}'.format(i)) for i in range(cfgParams.numInputs)]
        self.cfgParams = cfgParams
        self.rng = rng
        self.layers = []
        nConv = cfgParams.numInputs * cfgParams.inpConv
        for i, layerParam in enumerate(cfgParams.layers):
            if i % cfgParams.inpConv == 0 and i < nConv:
                inp = self.inputVar[i // cfgParams.inpConv]                      # a tower starts on its own input
            elif i == nConv:
                inp = concatenate([self.layers[(j + 1) * cfgParams.inpConv - 1].output.flatten(2)
                                   for j in range(cfgParams.numInputs)], axis=1)  # scalenet.py:167-171
            else:
                inp = self.layers[-1].output
            ctor = _LAYER_CLASSES[layerParam.__class__.__name__[:-11]]           # '<X>LayerParams' -> '<X>'
            # shared_conv: towers 2 and 3 run on the FIRST tower's filters and biases (scalenet.py:177-178) -- same shapes, the
            # towers differ in pooling only
            cl = None if twin is None else twin.layers[i]
            if cl is None and cfgParams.shared_conv is True and cfgParams.inpConv - 1 < i < nConv:
                cl = self.layers[i % cfgParams.inpConv]
            self.layers.append(ctor(rng, inputVar=inp, cfgParams=layerParam, copyLayer=cl, layerNum=len(self.layers)))
        self.output = self.layers[-1].output
        self.load(self.cfgParams.loadFile)
