"""
The minimal symbolic layer the reference's scripts rely on (they pass `layer.output` handles around, flatten
them, add them, and re-point `net.output`).  A `Var` records how a value is produced; nothing is computed until
hipdp.engine compiles the graph reachable from `net.output` into kernel launches.
"""


class Var(object):
    """Symbolic value: kind in {'input', 'layer', 'add', 'flatten', 'reshape', 'relu', 'concat'}."""

    def __init__(self, kind, inputs=(), layer=None, shape=None, name=None):
        self.kind = kind
        self.inputs = tuple(inputs)
        self.layer = layer
        self.shape = tuple(shape) if shape is not None else None
        self.name = name

    def flatten(self, ndim=2):
        if ndim != 2:
            raise NotImplementedError("only flatten(2) is used by the reference nets")
        shp = None
        if self.shape is not None:
            n = 1
            for s in self.shape[1:]:
                n *= s
            shp = (self.shape[0], n)
        return Var('flatten', (self,), shape=shp)

    def reshape(self, shape, ndim=None):
        return Var('reshape', (self,), shape=tuple(shape))

    def __add__(self, other):
        if not isinstance(other, Var):
            raise TypeError("can only add symbolic outputs")
        return Var('add', (self, other), shape=self.shape)

    def __repr__(self):
        return "Var(%s%s)" % (self.kind, '' if self.name is None else ', ' + self.name)


def tensor4(name='x'):
    return Var('input', name=name)


def concatenate(vars_, axis=1):
    """T.concatenate of flattened (2-D) values along the feature axis (scalenet.py:167-171)."""
    if axis != 1:
        raise NotImplementedError("only axis=1 concatenation is used by the reference nets")
    shp = None
    if all(v.shape is not None for v in vars_):
        shp = (vars_[0].shape[0], sum(v.shape[1] for v in vars_))
    return Var('concat', tuple(vars_), shape=shp)


class SharedParam(object):
    """Stand-in for a Theano shared variable: get_value()/set_value() in the REFERENCE layout.  Before a net is
    compiled the value lives on the host; afterwards the engine binds it to its slice of the flat device
    parameter buffer (kernel layout) and get/set go through the device."""
    _counter = [0]

    def __init__(self, value, name=None):
        self._host = value
        self.name = name
        self._binding = None           # (engine, slot)
        SharedParam._counter[0] += 1
        self.auto_name = 'auto_%d' % SharedParam._counter[0]

    def get_value(self, borrow=False):
        if self._binding is not None:
            eng, slot = self._binding
            return eng.read_param(slot)
        return self._host

    def set_value(self, value, borrow=False):
        import numpy
        value = numpy.asarray(value)
        if tuple(value.shape) != tuple(self._host.shape):
            raise ValueError("shape mismatch for %s: %s vs %s" % (self.name, value.shape, self._host.shape))
        self._host = numpy.asarray(value, dtype=self._host.dtype)
        if self._binding is not None:
            eng, slot = self._binding
            eng.write_param(slot, self._host)

    @property
    def shape(self):
        return self._host.shape
