"""Name kept for scripts written against the reference (objective/jax.py:16): ``JAXObjectifFunc`` wraps an arbitrary
JAX callable there.  JAX is not part of this build and a Python callable cannot be compiled into the HIP callback, so
constructing it here explains the two replacements instead of failing with an AttributeError."""
from .base import ObjectiveFunc


class JAXObjectifFunc(ObjectiveFunc):
    def __init__(self, func):
        raise NotImplementedError(
            "JAXObjectifFunc needs JAX, which this build does not use.  Use objective.QuadraticObjective for costs of "
            "the form sum (x-xref)'Q(x-xref) + (u-uref)'R(u-uref) + cx.x + cu.u (evaluated inside the fused HIP callback; "
            "covers the reference's sum(u*c) and sum((u-2)^2)), or objective.TorchObjectifFunc(func) for any "
            "differentiable torch callable func(states, u, p, tvp).")
