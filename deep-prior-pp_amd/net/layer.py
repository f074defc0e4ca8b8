"""Layer base: weight initialisation (API of /root/reference/src/net/layer.py:36-124)."""
import numpy

from util.theano_helpers import ReLU, sigmoid, tanh

floatX = 'float32'


class Layer(object):
    def __init__(self, rng):
        self.weights = []
        self.params = []
        self.params_nontrained = []
        self.rng = rng

    def getOptimalInitMethod(self, act_str):
        if act_str == ReLU.__name__:
            return 'He'
        if act_str == sigmoid.__name__:
            return 'sigmoid'
        if act_str in tanh.__name__:
            return 'tanh'
        if act_str is None or str(act_str) == 'None':
            return None
        raise NotImplementedError("Unknown activation function: {}".format(act_str))

    def orthogonalize(self, init_vals):
        flat = numpy.reshape(init_vals, (init_vals.shape[0], -1))
        U = numpy.linalg.svd(flat.T)[0]
        return numpy.reshape(U.T[0:init_vals.shape[0]].T.swapaxes(0, 1), init_vals.shape)

    def getInitVals(self, shape, mode, act_fn=None, method=None, orthogonal=False):
        """Same draws from self.rng, in the same order, as the reference (layer.py:70-124): 'He' = normal with
        std sqrt(2/fan_in) for conv and std 0.01 for fc; 'Xavier'/'sigmoid'/'tanh'(default) = the uniform rules."""
        if act_fn is None and method is None:
            raise UserWarning("act_fn and method not defined! At least one must be specified.")
        if act_fn is not None and method is None:
            method = self.getOptimalInitMethod(act_fn)
        if mode not in ('conv', 'fc'):
            raise NotImplementedError()
        fan_in = numpy.prod(shape[1:])
        if method == 'He':
            std = numpy.sqrt(2. / fan_in) if mode == 'conv' else 0.01
            vals = self.rng.normal(loc=0.0, scale=std, size=shape)
        elif method == 'Xavier':
            b = numpy.sqrt(3. / fan_in) if mode == 'conv' else numpy.sqrt(1. / shape[0])
            vals = self.rng.uniform(low=-b, high=b, size=shape)
        elif method == 'sigmoid':
            if mode == 'conv':
                b = 4. * numpy.sqrt(6. / (fan_in + (shape[0] * numpy.prod(shape[2:]))))
                vals = self.rng.uniform(low=-b, high=b, size=shape)
            else:
                b = numpy.sqrt(6. / numpy.sum(shape))
                vals = 4. * numpy.asarray(self.rng.uniform(low=-b, high=b, size=shape), dtype=floatX)
        elif method == 'tanh' or method is None:
            b = 1. / (fan_in + (shape[0] * numpy.prod(shape[2:]))) if mode == 'conv' else numpy.sqrt(6. / numpy.sum(shape))
            vals = self.rng.uniform(low=-b, high=b, size=shape)
        else:
            raise NotImplementedError("Unknown method!")
        vals = numpy.asarray(vals, dtype=floatX)
        return vals if orthogonal is False else self.orthogonalize(vals)
