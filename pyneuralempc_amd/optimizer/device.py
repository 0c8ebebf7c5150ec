"""``DeviceSqp``: the batched on-device solver (csrc/solver.hip, ``CallbackEngine.solve``) behind the reference's
``Optimizer`` interface, for ``NMPC.next`` on ONE problem.  The reference has no counterpart -- its optimizers are Ipopt
(optimizer/ipopt.py:138-195) and SciPy SLSQP (optimizer/slsqp.py:143-197), both on the CPU, calling back into the model
once per iterate; here the whole iteration (callbacks, Riccati solve, line search) stays on the GPU and the host gets the
solution.  Same plumbing as ``Slsqp``: problem object from the factory, cold start / shifted warm start
(``init_with_last_result``), ``prev_result`` for ``NMPC.next`` to split.

Needs the fused device path: a device integrator, a ``QuadraticObjective`` and no extra constraint rows other than one
``BoxStateConstraint`` (which the solver turns into state bounds); anything else raises ``NotImplementedError`` -- there
is no CPU fallback."""
import numpy as np

from .base import Optimizer
from .slsqp import Slsqp, SlsqpProblemFactory


class DeviceSqp(Slsqp):
    def __init__(self, max_iteration=200, tolerance=None, verbose=0, init_with_last_result=False, warm_mu=1e-4,
                 **solver_opts):
        """tolerance: max |defect| and relative step at convergence (tol_constraint = tol_step); None = what an iterate
        of the model's precision can reach (CallbackEngine.solve: 1e-8 for fp64, 1e-4 for fp32 -- an fp32 iterate stalls
        near 1e-5 and would never report success against 1e-8); warm_mu: initial barrier
        parameter of a warm-started solve (a cold start uses the solver's default of 0.1); solver_opts: further keyword
        arguments of CallbackEngine.solve (linesearch, lq_kernel, mu_min, ...)."""
        super().__init__(max_iteration=max_iteration, tolerance=tolerance, verbose=verbose,
                         init_with_last_result=init_with_last_result)
        self.warm_mu = warm_mu
        self.solver_opts = dict(solver_opts)
        self.last_iterations = None

    def get_factory(self):
        return SlsqpProblemFactory()      # the same problem object: x0, parameters, fused evaluator, initial values

    def solve(self, problem, domain_constraint):
        fused = getattr(problem, "_fused", None)
        if fused is None:
            raise NotImplementedError("DeviceSqp needs the fused device path: a device integrator, a QuadraticObjective and "
                                      "no extra constraint rows other than one BoxStateConstraint")
        eng = fused.engine
        H = problem.integrator.H
        warm = (problem.get_init_variables()[0] is not None) or (self.init_with_last_result and self.prev_result is not None)
        z0 = np.asarray(self.initial_point(problem), dtype=np.float64).reshape(1, -1)
        lb = np.asarray(domain_constraint.get_lower_bounds(H), dtype=np.float64)
        ub = np.asarray(domain_constraint.get_upper_bounds(H), dtype=np.float64)
        opts = dict(max_iter=self.max_iteration)
        if self.tolerance is not None:
            opts.update(tol_constraint=self.tolerance, tol_step=self.tolerance)
        if warm:
            opts["mu_init"] = self.warm_mu
        opts.update(self.solver_opts)
        X0 = eng.to_device(np.asarray(problem.get_init_value(), dtype=np.float64).reshape(1, -1))
        Z, status, iters = eng.solve(X0, eng.to_device(z0), lb, ub, **opts)
        self.last_iterations = iters
        if int(status[0].item()) != 0:
            return Optimizer.FAIL
        self.prev_result = Z[0].to("cpu").double().numpy()
        return Optimizer.SUCCESS
