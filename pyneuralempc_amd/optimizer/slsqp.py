"""SciPy SLSQP glue (reference: pyNeuralEMPC/optimizer/slsqp.py:10-197)."""
import warnings

import numpy as np
from scipy.optimize import Bounds, minimize

from ..constraints import Constraint
from .base import Optimizer, ProblemFactory, ProblemInterface, _CallbackGlue, cold_start, warm_start_shift


class SlsqpProblem(_CallbackGlue, ProblemInterface):
    def __init__(self, x0, objective_func, constraints, integrator, p=None, tvp=None, init_x=None, init_u=None):
        ProblemInterface.__init__(self, False)
        self._setup(x0, objective_func, constraints, integrator, p, tvp)
        self.debug_mode = False
        self.debug_x, self.debug_u = list(), list()
        self.init_x, self.init_u = init_x, init_u

    def set_debug(self, debug_mode):
        self.debug_mode = debug_mode

    def objective(self, x):
        if self.debug_mode:
            states, u, _, _ = self._split(x)
            self.debug_x.append(states.copy())
            self.debug_u.append(u.copy())
        return self._objective(x)

    def gradient(self, x):
        return self._gradient(x)

    def _row_groups(self):
        """(start, stop, type, lo, hi) of every extra constraint inside the stacked g."""
        groups, start = [], self.integrator.nb_contraints
        for c in self.constraints_list:
            k = int(c.get_dim(self.H))
            groups.append((start, start + k, c.get_type(self.H), np.asarray(c.get_lower_bounds(self.H)),
                           np.asarray(c.get_upper_bounds(self.H))))
            start += k
        return groups

    def constraints(self, x, eq=True):
        """eq=True: integrator defects + EQ rows (== 0);  eq=False: INEQ rows (>= 0) and INTER rows
        twice, (g - lo, hi - g) (slsqp.py:54-72; the reference's INTER branch calls two accessors that
        do not exist, slsqp.py:67-68 -- the plural ones are meant)."""
        g = self._all_constraints(x)
        nint = self.integrator.nb_contraints
        parts = [g[:nint]] if eq else []
        for a, b, kind, lo, hi in self._row_groups():
            if eq and kind == Constraint.EQ_TYPE:
                parts.append(g[a:b])
            elif not eq and kind == Constraint.INEQ_TYPE:
                parts.append(g[a:b])
            elif not eq and kind == Constraint.INTER_TYPE:
                parts.append(g[a:b] - lo)
                parts.append(hi - g[a:b])
        return np.concatenate(parts, axis=0)

    def jacobian(self, x, eq=True):
        J = self._all_jacobian(x)
        nint = self.integrator.nb_contraints
        parts = [J[:nint]] if eq else []
        for a, b, kind, _, _ in self._row_groups():
            if eq and kind == Constraint.EQ_TYPE:
                parts.append(J[a:b])
            elif not eq and kind == Constraint.INEQ_TYPE:
                parts.append(J[a:b])
            elif not eq and kind == Constraint.INTER_TYPE:
                parts.append(J[a:b])
                parts.append(-J[a:b])
        return np.concatenate(parts, axis=0)

    def hessianstructure(self):
        raise NotImplementedError("Not needed")

    def hessian(self, x, lagrange, obj_factor):
        raise NotImplementedError("Not needed")

    def get_constraints_dict(self):
        result = [{"type": "eq", "fun": lambda x: self.constraints(x, eq=True),
                   "jac": lambda x: self.jacobian(x, eq=True)}]
        if any(c.get_type(self.H) in (Constraint.INEQ_TYPE, Constraint.INTER_TYPE) for c in self.constraints_list):
            result.append({"type": "ineq", "fun": lambda x: self.constraints(x, eq=False),
                           "jac": lambda x: self.jacobian(x, eq=False)})
        return result


class SlsqpProblemFactory(ProblemFactory):
    def _process(self):
        return SlsqpProblem(self.x0, self.objective, self.constraints, self.integrator, p=self.p, tvp=self.tvp,
                            init_x=self.init_x, init_u=self.init_u)


class Slsqp(Optimizer):
    def __init__(self, max_iteration=200, tolerance=0.5e-6, verbose=1, init_with_last_result=False, nb_max_try=15,
                 debug=False):
        super().__init__()
        self.max_iteration = max_iteration
        self.verbose = verbose
        self.tolerance = tolerance
        self.init_with_last_result = init_with_last_result
        self.prev_result = None
        self.nb_max_try = nb_max_try
        self.debug = debug

    def get_factory(self):
        return SlsqpProblemFactory()

    def initial_point(self, problem):
        H = problem.integrator.H
        model = problem.integrator.model
        init_x, init_u = problem.get_init_variables()
        if init_x is not None and init_u is not None:
            assert init_u.shape[0] == H, ("The init u values is not compliant with the MPC horizon size "
                                          f"(receive ={init_u.shape[0]}, expected={H})")
            assert init_x.shape[0] == H, ("The init x values is not compliant with the MPC horizon size "
                                          f"(receive ={init_x.shape[0]}, expected={H})")
            return np.concatenate([np.asarray(init_x, dtype=np.float64).reshape(-1),
                                   np.asarray(init_u, dtype=np.float64).reshape(-1)])
        if self.init_with_last_result and self.prev_result is not None:
            return warm_start_shift(self.prev_result, H, model.x_dim, model.u_dim)
        return cold_start(problem.get_init_value(), H, model.u_dim)

    def _minimize(self, problem, x_init, bounds, ftol):
        return minimize(problem.objective, x_init, method="SLSQP", jac=problem.gradient,
                        constraints=problem.get_constraints_dict(), bounds=bounds,
                        options={"maxiter": self.max_iteration, "ftol": ftol, "disp": self.verbose > 0,
                                 "iprint": self.verbose})

    def solve(self, problem, domain_constraint):
        problem.set_debug(self.debug)
        H = problem.integrator.H
        x_init = self.initial_point(problem)
        bounds = Bounds(domain_constraint.get_lower_bounds(H), domain_constraint.get_upper_bounds(H))
        res = self._minimize(problem, x_init, bounds, self.tolerance)
        if self.debug:
            self.constraints_val = problem.constraints(res.x)
            self.debug_x, self.debug_u = problem.debug_x, problem.debug_u
        if not res.success:
            warnings.warn("Process do not converge ! ")
            if self.debug:
                return Optimizer.FAIL
            # retry ladder of the reference (slsqp.py:184-194): cold start, tolerance doubled each try
            if np.max(problem.constraints(res.x)) > 1e-5:
                cold = cold_start(problem.get_init_value(), H, problem.integrator.model.u_dim)
                for i in range(self.nb_max_try):
                    print("RETRY SQP optimization")
                    res = self._minimize(problem, cold, bounds, self.tolerance * (2.0 ** i))
                    if np.max(problem.constraints(res.x)) < 1e-5 or res.success:
                        break
            if not res.success and (np.max(problem.constraints(res.x)) > 1e-5):
                return Optimizer.FAIL
        self.prev_result = res.x
        return Optimizer.SUCCESS
