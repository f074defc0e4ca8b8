"""Activation identifiers (API of /root/reference/src/util/theano_helpers.py:35-69).  In the reference these
are Theano expressions; here they are tags the engine maps onto kernel prologues/epilogues -- calling one on a
symbolic output returns a symbolic activation node."""
import numpy

EPS = numpy.float32(3. * numpy.finfo(numpy.float32).eps)
PI = numpy.float32(numpy.pi)


def sigmoid(x):
    raise NotImplementedError("sigmoid is not on the DeepPrior++ hot path")


def tanh(x):
    raise NotImplementedError("tanh is not on the DeepPrior++ hot path")


def ReLU(x):
    """max(x, 0); its gradient passes where x >= 0 (Theano's T.maximum)."""
    from hipdp.graph import Var
    return Var('relu', (x,), shape=getattr(x, 'shape', None))
