"""PoseRegNet / PoseRegNetParams (API of /root/reference/src/net/poseregnet.py:44-165): the DeepPose-style
regressor the main_*_posereg_embedding.py scripts build (type 0; type 11 adds the 30-D bottleneck)."""
from hipdp.graph import tensor4
from net.convpoollayer import ConvPoolLayerParams
from net.dropoutlayer import DropoutLayerParams
from net.hiddenlayer import HiddenLayerParams
from net.netbase import NetBase, NetBaseParams
from util.theano_helpers import ReLU


class PoseRegNetParams(NetBaseParams):
    def __init__(self, type=0, nChan=1, wIn=128, hIn=128, batchSize=128, numJoints=16, nDims=3):
        super(PoseRegNetParams, self).__init__()
        self.batch_size = batchSize
        self.numJoints = numJoints
        self.nDims = nDims
        self.inputDim = (batchSize, nChan, hIn, wIn)
        self.type = type
        if type not in (0, 11):
            raise NotImplementedError("not implemented")
        L = self.layers
        dim = self.inputDim
        for fd, pool in (((5, 5), (4, 4)), ((5, 5), (2, 2)), ((3, 3), (1, 1))):     # poseregnet.py:62-78
            L.append(ConvPoolLayerParams(inputDim=dim, nFilters=8, filterDim=fd, poolsize=pool, activation=ReLU))
            dim = L[-1].outputDim
        dim = (dim[0], dim[1] * dim[2] * dim[3])
        for _ in range(2):                                                          # poseregnet.py:80-93
            L.append(HiddenLayerParams(inputDim=dim, outputDim=(batchSize, 1024), activation=ReLU))
            dim = L[-1].outputDim
            L.append(DropoutLayerParams(inputDim=dim, outputDim=dim))
        if type == 11:
            L.append(HiddenLayerParams(inputDim=dim, outputDim=(batchSize, 30), activation=None))
            dim = L[-1].outputDim
        L.append(HiddenLayerParams(inputDim=dim, outputDim=(batchSize, numJoints * nDims), activation=None))
        self.outputDim = L[-1].outputDim


class PoseRegNet(NetBase):
    def __init__(self, rng, inputVar=None, cfgParams=None):
        if cfgParams is None:
            raise Exception("Cannot create a Net without config parameters (ie. cfgParams==None)")
        if inputVar is None:
            inputVar = tensor4('x')
        elif isinstance(inputVar, str):
            inputVar = tensor4(inputVar)
        super(PoseRegNet, self).__init__(rng, inputVar, cfgParams)
