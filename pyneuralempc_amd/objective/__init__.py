# Objective functions (reference: pyNeuralEMPC/objective/__init__.py).
from . import base
from . import quadratic
from . import autodiff
from . import jax
from .base import ObjectiveFunc, ManualObjectifFunc
from .quadratic import QuadraticObjective
from .autodiff import TorchObjectifFunc
