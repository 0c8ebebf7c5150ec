"""Objective plug-in interface (reference: pyNeuralEMPC/objective/base.py:4-37).

The solver glue calls ``forward / gradient / hessian(states, u, p=, tvp=)`` and ``hessianstructure(H, model)``
(optimizer/ipopt.py:33,40,56,71).  Three implementations ship: ``QuadraticObjective`` (evaluated inside the fused
HIP callback), ``TorchObjectifFunc`` (any differentiable torch callable) and ``ManualObjectifFunc`` below."""


class ObjectiveFunc:
    """Abstract cost: scalar value, gradient over [states.ravel() | u.ravel()], dense (n, n) Hessian and its 0/1
    structure map."""

    def __init__(self):
        pass

    def _missing(self, *_args, **_kwargs):
        raise NotImplementedError("")

    forward = gradient = hessian = hessianstructure = _missing


class ManualObjectifFunc(ObjectiveFunc):
    """Host callables supplied by the user, each called as ``f(states, u, p, tvp)``.

    The reference's constructor of this class cannot run (malformed ``super`` call, and ``self.func`` is never set:
    objective/base.py:22-26); this one keeps the callables its name promises, plus an optional structure callable
    ``hessianstructure_func(H, model)``."""

    def __init__(self, func, grad_func, hessian_func, hessianstructure_func=None):
        ObjectiveFunc.__init__(self)
        self._callables = {"forward": func, "gradient": grad_func, "hessian": hessian_func}
        self.func, self.grad_func, self.hessian_func = func, grad_func, hessian_func
        self.hessianstructure_func = hessianstructure_func

    def forward(self, states, u, p=None, tvp=None):
        return self._callables["forward"](states, u, p, tvp)

    def gradient(self, states, u, p=None, tvp=None):
        return self._callables["gradient"](states, u, p, tvp)

    def hessian(self, states, u, p=None, tvp=None):
        return self._callables["hessian"](states, u, p, tvp)

    def hessianstructure(self, H, model):
        if self.hessianstructure_func is None:
            raise NotImplementedError("")
        return self.hessianstructure_func(H, model)
