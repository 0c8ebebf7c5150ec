"""Objective plug-in interface (reference: pyNeuralEMPC/objective/base.py:4-37).

Call-site signature (optimizer/ipopt.py:33,40,56,71): forward/gradient/hessian(states, u, p=, tvp=)
and hessianstructure(H, model)."""


class ObjectiveFunc:
    def __init__(self):
        pass

    def forward(self, states, u, p=None, tvp=None):
        raise NotImplementedError("")

    def gradient(self, states, u, p=None, tvp=None):
        raise NotImplementedError("")

    def hessian(self, states, u, p=None, tvp=None):
        raise NotImplementedError("")

    def hessianstructure(self, H, model):
        raise NotImplementedError("")


class ManualObjectifFunc(ObjectiveFunc):
    """User-supplied host callables func / grad_func / hessian_func (states, u, p, tvp).
    (The reference's constructor of this class raises TypeError, objective/base.py:22-26; this one
    stores the three callables as its name promises.)"""

    def __init__(self, func, grad_func, hessian_func, hessianstructure_func=None):
        super().__init__()
        self.func = func
        self.grad_func = grad_func
        self.hessian_func = hessian_func
        self.hessianstructure_func = hessianstructure_func

    def forward(self, states, u, p=None, tvp=None):
        return self.func(states, u, p, tvp)

    def gradient(self, states, u, p=None, tvp=None):
        return self.grad_func(states, u, p, tvp)

    def hessian(self, states, u, p=None, tvp=None):
        return self.hessian_func(states, u, p, tvp)

    def hessianstructure(self, H, model):
        if self.hessianstructure_func is None:
            raise NotImplementedError("")
        return self.hessianstructure_func(H, model)
