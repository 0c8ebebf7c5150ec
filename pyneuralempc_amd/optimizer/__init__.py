"""Solver front ends (reference: pyNeuralEMPC/optimizer/__init__.py): ``Optimizer`` is the interface NMPC talks to,
``Ipopt`` (cyipopt, optional dependency) and ``Slsqp`` (SciPy) drive the device callbacks one problem at a time; the
batched on-device solver is reached through ``NMPC.next_batch`` / ``CallbackEngine.solve``."""
from . import base, ipopt, slsqp

Optimizer = base.Optimizer
Ipopt = ipopt.Ipopt
Slsqp = slsqp.Slsqp

__all__ = ["Optimizer", "Ipopt", "Slsqp"]
