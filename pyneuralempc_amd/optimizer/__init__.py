"""Solver front ends (reference: pyNeuralEMPC/optimizer/__init__.py): ``Optimizer`` is the interface NMPC talks to,
``Ipopt`` (cyipopt, optional dependency) and ``Slsqp`` (SciPy) drive the device callbacks one problem at a time; the
batched on-device solver is reached through ``NMPC.next_batch`` / ``CallbackEngine.solve``, and through ``DeviceSqp``
for ``NMPC.next`` on one problem."""
from . import base, device, ipopt, slsqp

Optimizer = base.Optimizer
Ipopt = ipopt.Ipopt
Slsqp = slsqp.Slsqp
DeviceSqp = device.DeviceSqp

__all__ = ["Optimizer", "Ipopt", "Slsqp", "DeviceSqp"]
