"""pyneuralempc_amd -- MI355X-native NMPC callback engine behind pyNeuralEMPC's plug-in surface.

Drop-in for the hot path of Enderdead/pyNeuralEMPC (per-iterate f, grad f, g, jac g, Lagrangian
Hessian of a neural-network NMPC problem), evaluated by hand-written HIP kernels for gfx950 through
the C ABI in include/nempc.h.  There is no CPU fallback.
"""
__version__ = "0.1"

from . import model
from . import objective
from . import integrator
from . import optimizer
from . import constraints
from . import controller
from . import parallel
from .engine import CallbackEngine
