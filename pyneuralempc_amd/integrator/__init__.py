# Derivation and integration models (reference: pyNeuralEMPC/integrator/__init__.py).
from . import base
from . import discret
from . import rk4
from . import unity
from .base import Integrator
from .discret import DiscretIntegrator
from .unity import UnityIntegrator
from .rk4 import RK4Integrator
