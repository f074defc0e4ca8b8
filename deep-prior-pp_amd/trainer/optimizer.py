"""Optimizer (API of /root/reference/src/trainer/optimizer.py:38-116).  In the reference this builds Theano update
expressions; here it records which update rule the trainer's compiled step applies.  ADAM (with the gamma-decayed
beta1) is executed by the fused dpp_adam kernel over the flat parameter buffer; see hipdp.engine."""


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
        if (beta1, beta2, epsilon) != (0.9, 0.999, 1e-8):
            raise NotImplementedError("the fused ADAM kernel is configured with the reference's defaults")
        self.updates = [('adam', p) for p in self.params]
        return self.updates

    def RMSProp(self, learning_rate=0.01, decay=0.9, epsilon=1.0 / 100.):
        raise NotImplementedError("RMSProp is never selected by the reference's trainers (poseregnettrainer.py:147-149)")
