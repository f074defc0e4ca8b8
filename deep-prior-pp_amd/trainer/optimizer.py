"""Optimizer (API of /root/reference/src/trainer/optimizer.py:38-116).  In the reference this builds Theano update
expressions; here it records the update rule and its constants for the trainer's compiled step: `rule` goes to
hipdp.engine.CompiledNet(optimizer=), which writes it into the device-resident hyper-parameter block that the fused update kernel
(csrc/elementwise.hip: adam_kernel, one launch over the flat parameter buffer) reads."""


class Optimizer(object):
    def __init__(self, grads, params):
        self.grads = grads
        self.params = params
        self.updates = []
        self.shared = []
        self.rule = None
        if len(grads) != len(params):
            print("Warning: Size of gradients ({}) does not fit size of parameters ({})!".format(len(grads), len(params)))

    def ADAM(self, learning_rate=0.0002, beta1=0.9, beta2=0.999, epsilon=1e-8, gamma=1 - 1e-8):
        """Adam (Kingma & Ba) with momentum decay, optimizer.py:58-90: t starts at 1; beta1_t = beta1*gamma^(t-1);
        m = beta1_t*m + (1-beta1_t)*g; v = beta2*v + (1-beta2)*g^2; w -= lr*(m/(1-beta1^t)) / (sqrt(v/(1-beta2^t)) + eps)."""
        self.rule = dict(name='ADAM', learning_rate=learning_rate, beta1=beta1, beta2=beta2, epsilon=epsilon, gamma=gamma)
        self.shared = [(k, p) for p in self.params for k in ('m', 'v')]          # 1st / 2nd moment per parameter
        self.updates = [('adam', p) for p in self.params]
        return self.updates

    def RMSProp(self, learning_rate=0.01, decay=0.9, epsilon=1.0 / 100.):
        """RMSProp of Tieleman et al., optimizer.py:92-116: msg = decay*msg + (1-decay)*g^2; w += -lr * g / max(sqrt(msg), epsilon)."""
        self.rule = dict(name='RMSProp', learning_rate=learning_rate, decay=decay, epsilon=epsilon)
        self.shared = [('msg', p) for p in self.params]
        self.updates = [('rmsprop', p) for p in self.params]
        return self.updates
