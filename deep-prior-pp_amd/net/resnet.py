"""ResNet / ResNetParams (API of /root/reference/src/net/resnet.py:45-414): the pre-activation bottleneck
ResNet-47 pose regressor (4 stages x 5 blocks, widths 32-64-128-256-256).  Types 0 and 1 (type 1 = 30-D
bottleneck + prior layer) are on the benchmarked hot path; types 2-4 (dropout variants) build the same graph
with DropoutLayers."""
import numpy

from hipdp.graph import tensor4
from net.batchnormlayer import BatchNormLayer, BatchNormLayerParams
from net.convlayer import ConvLayer, ConvLayerParams
from net.convpoollayer import ConvPoolLayer, ConvPoolLayerParams
from net.dropoutlayer import DropoutLayer, DropoutLayerParams
from net.hiddenlayer import HiddenLayer, HiddenLayerParams
from net.netbase import NetBase, NetBaseParams
from net.nonlinearitylayer import NonlinearityLayer, NonlinearityLayerParams
from util.theano_helpers import ReLU


class ResNetParams(NetBaseParams):
    def __init__(self, type=0, nChan=1, wIn=128, hIn=128, batchSize=128, numJoints=16, nDims=3):
        super(ResNetParams, self).__init__()
        if type not in (0, 1, 2, 3, 4):
            raise NotImplementedError("not implemented")
        self.batch_size = batchSize
        self.numJoints = numJoints
        self.nDims = nDims
        self.numInputs = 1
        self.numOutputs = 1
        self.inputDim = (batchSize, nChan, hIn, wIn)
        self.outputDim = (batchSize, numJoints * nDims)
        self.type = type


# per type: (stage widths, dropout after the 1024-wide layers, 30-D bottleneck)   resnet.py:120-336
_VARIANTS = {0: ([32, 64, 128, 256, 256], False, False), 1: ([32, 64, 128, 256, 256], False, True),
             2: ([32, 64, 128, 256, 256], True, False), 3: ([32, 64, 128, 128, 128], True, False),
             4: ([32, 64, 128, 256, 256], True, True)}


class ResNet(NetBase):
    def __init__(self, rng, inputVar=None, cfgParams=None):
        self._params_filter = []
        self._weights_filter = []
        if cfgParams is None:
            raise Exception("Cannot create a Net without config parameters (ie. cfgParams==None)")
        if inputVar is None:
            inputVar = tensor4('x')
        elif isinstance(inputVar, str):
            raise NotImplementedError()
        self.inputVar = inputVar
        self.cfgParams = cfgParams
        self.rng = rng
        self.layers = []
        if cfgParams.type not in _VARIANTS:
            raise NotImplementedError()
        nStages, dropout, bottleneck = _VARIANTS[cfgParams.type]
        depth = 47
        assert (depth - 2) % 9 == 0, 'depth should be 9n+2 (e.g., 164 or 1001)'
        n = (depth - 2) // 9
        batchSize = cfgParams.batch_size
        L = self.layers
        L.append(ConvPoolLayer(rng, self.inputVar,
                               ConvPoolLayerParams(inputDim=cfgParams.inputDim, nFilters=nStages[0], filterDim=(5, 5),
                                                   stride=(1, 1), poolsize=(2, 2), border_mode='same', activation=None,
                                                   init_method='He'), layerNum=len(L)))
        rout = L[-1].output
        for s in range(1, 5):
            rout = self.add_res_layers(rng, rout, L[-1].cfgParams.outputDim, nStages[s], n, 2)
        L.append(BatchNormLayer(rng, rout, BatchNormLayerParams(inputDim=L[-1].cfgParams.outputDim), layerNum=len(L)))
        L.append(NonlinearityLayer(rng, L[-1].output, NonlinearityLayerParams(inputDim=L[-1].cfgParams.outputDim, activation=ReLU),
                                   layerNum=len(L)))
        od = L[-1].cfgParams.outputDim
        inp, dim = L[-1].output.flatten(2), (od[0], int(numpy.prod(od[1:])))
        for _ in range(2):
            L.append(HiddenLayer(rng, inp, HiddenLayerParams(inputDim=dim, outputDim=(batchSize, 1024), activation=ReLU),
                                 layerNum=len(L)))
            inp, dim = L[-1].output, L[-1].cfgParams.outputDim
            if dropout:
                L.append(DropoutLayer(rng, inp, DropoutLayerParams(inputDim=dim, outputDim=dim), layerNum=len(L)))
                inp = L[-1].output
        if bottleneck:
            L.append(HiddenLayer(rng, inp, HiddenLayerParams(inputDim=dim, outputDim=(batchSize, 30), activation=None),
                                 layerNum=len(L)))
            inp, dim = L[-1].output, L[-1].cfgParams.outputDim
        L.append(HiddenLayer(rng, inp, HiddenLayerParams(inputDim=dim, outputDim=(batchSize, cfgParams.numJoints * cfgParams.nDims),
                                                         activation=None), layerNum=len(L)))
        self.output = L[-1].output
        self.load(self.cfgParams.loadFile)

    def add_res_layers(self, rng, inputVar, inputDim, outputFilters, count, stride):
        rout = res_block(self.layers, rng, inputVar, inputDim, outputFilters, stride)
        for _ in range(1, count):
            rout = res_block(self.layers, rng, rout, self.layers[-1].cfgParams.outputDim, outputFilters, 1)
        return rout


def _bn_relu(layers, rng, inp, dim):
    layers.append(BatchNormLayer(rng, inp, BatchNormLayerParams(inputDim=dim), layerNum=len(layers)))
    layers.append(NonlinearityLayer(rng, layers[-1].output, NonlinearityLayerParams(inputDim=dim, activation=ReLU),
                                    layerNum=len(layers)))
    return layers[-1].output


def _conv(layers, rng, inp, dim, nf, k, stride=(1, 1)):
    layers.append(ConvLayer(rng, inp, ConvLayerParams(inputDim=dim, nFilters=nf, filterDim=k, stride=stride, border_mode='same',
                                                      activation=None, init_method='He'), layerNum=len(layers)))
    return layers[-1].output, layers[-1].cfgParams.outputDim


def res_block(layers, rng, inputVar, inputDim, outputFilters, stride, nBottleneckFilters=None):
    """Pre-activation bottleneck block (/root/reference/src/net/resnet.py:349-414).  Identity block when the channel
    count is unchanged (the stride argument is then ignored, as in the reference); otherwise a projection block whose
    1x1/stride shortcut reads the block's first BN+ReLU output."""
    nb = outputFilters // 4 if nBottleneckFilters is None else nBottleneckFilters
    identity = inputDim[1] == outputFilters
    s = (1, 1) if identity else (stride, stride)
    h = _bn_relu(layers, rng, inputVar, inputDim)
    c, d = _conv(layers, rng, h, inputDim, nb, (1, 1), s)
    c, d = _conv(layers, rng, _bn_relu(layers, rng, c, d), d, nb, (3, 3))
    c, d = _conv(layers, rng, _bn_relu(layers, rng, c, d), d, outputFilters, (1, 1))
    if identity:
        return inputVar + c
    sc, _ = _conv(layers, rng, h, inputDim, outputFilters, (1, 1), s)
    return c + sc
