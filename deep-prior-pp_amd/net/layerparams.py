"""Layer parametrisation base (API of /root/reference/src/net/layerparams.py:35-104)."""
import inspect

import numpy


class LayerParams(object):
    def __init__(self, inputDim, outputDim):
        self._inputDim = inputDim
        self._outputDim = outputDim

    def update(self):
        """Recompute dependent shapes; overridden by layers that derive their output shape."""
        return None

    def _get_in(self):
        return self._inputDim

    def _set_in(self, value):
        self._inputDim = value
        self.update()

    def _get_out(self):
        return self._outputDim

    def _set_out(self, value):
        self._outputDim = value
        self.update()

    inputDim = property(_get_in, _set_in)
    outputDim = property(_get_out, _set_out)

    @property
    def activation_str(self):
        if not hasattr(self, 'activation'):
            return ''
        act = self.activation
        if act is None:
            return str(None)
        if inspect.isclass(act):
            return act.__class__.__name__
        if inspect.isfunction(act):
            return act.__name__
        return str(act)

    def getOutputRange(self):
        rng = {'tanh': [-1, 1], 'sigmoid': [0, 1], 'ReLU': [0, numpy.inf]}
        if not hasattr(self, 'activation'):
            return [-numpy.inf, numpy.inf]
        return rng.get(self.activation_str, [-numpy.inf, numpy.inf])


def conv_output_dim(inputDim, nFilters, filterDim, stride, border_mode, poolsize=(1, 1)):
    """Shape rule shared by ConvLayerParams / ConvPoolLayerParams
    (/root/reference/src/net/convlayer.py:131-163, convpoollayer.py:145-181): 'valid' H-k+1, 'full' H+k-1,
    'half' H; then ceil(/stride); conv-pool additionally floor-divides by the pool size."""
    H, W = inputDim[2], inputDim[3]
    if border_mode == 'valid':
        oh, ow = H - filterDim[0] + 1, W - filterDim[1] + 1
    elif border_mode == 'full':
        oh, ow = H + filterDim[0] - 1, W + filterDim[1] - 1
    elif border_mode == 'half':
        oh, ow = H, W
    else:
        raise ValueError("Unknown border mode")
    oh = int(numpy.ceil(oh / float(stride[0]))) // poolsize[0]
    ow = int(numpy.ceil(ow / float(stride[1]))) // poolsize[1]
    return (inputDim[0], nFilters, oh, ow)
