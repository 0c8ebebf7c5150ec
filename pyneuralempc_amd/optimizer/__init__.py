# Solver interfaces (reference: pyNeuralEMPC/optimizer/__init__.py).
from .ipopt import Ipopt
from .base import Optimizer
from .slsqp import Slsqp
