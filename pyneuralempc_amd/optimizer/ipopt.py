"""Ipopt glue (reference: pyNeuralEMPC/optimizer/ipopt.py:7-195).  Ipopt itself (C++, via cyipopt)
is third-party and stays on the CPU; it is imported lazily in Ipopt.solve."""
import numpy as np

from .base import (Optimizer, ProblemFactory, ProblemInterface, ProblemInterfaceHessianFree, _CallbackGlue,
                   cold_start, warm_start_shift)


class IpoptProblem(_CallbackGlue, ProblemInterface):
    def __init__(self, x0, objective_func, constraints, integrator, p=None, tvp=None, use_hessian=True, init_x=None,
                 init_u=None):
        ProblemInterface.__init__(self, use_hessian)
        self._setup(x0, objective_func, constraints, integrator, p, tvp)
        self.init_x, self.init_u = None, None  # the reference's Ipopt path ignores user guesses (ipopt.py:18)

    def objective(self, x):
        return self._objective(x)

    def gradient(self, x):
        return self._gradient(x)

    def constraints(self, x):
        return self._all_constraints(x)

    def jacobian(self, x):
        return self._all_jacobian(x)

    def hessianstructure(self):
        if self._fused is not None:
            rows, cols = self._fused.engine.hess_structure()
            return rows.astype(np.int64), cols.astype(np.int64)
        pattern = (self.objective_func.hessianstructure(self.H, self.integrator.model)
                   + self.integrator.hessianstructure()) != 0
        return np.nonzero(np.tril(pattern))

    def hessian(self, x, lagrange, obj_factor):
        if self._fused is not None:
            return self._fused.hessian_values(x, self.x0, lagrange, obj_factor)
        states, u, tvp, p = self._split(x)
        total = obj_factor * np.asarray(self.objective_func.hessian(states, u, p=p, tvp=tvp), dtype=np.float64)
        blocks = [self.integrator.hessian(states, u, self.x0, p=p, tvp=tvp)]
        blocks += [c.hessian(states, u, p=p, tvp=tvp) for c in self.constraints_list]
        total = total + np.tensordot(np.asarray(lagrange, dtype=np.float64), np.concatenate(blocks, axis=0), axes=1)
        rows, cols = self.hessianstructure()
        return total[rows, cols]

    # sparse-Jacobian extension: exact band pattern instead of dense (m,n)
    def jacobianstructure(self):
        if self._fused is None:
            raise NotImplementedError("sparse Jacobian needs the fused device path")
        rows, cols = self._fused.engine.jac_structure()
        return rows.astype(np.int64), cols.astype(np.int64)


class _SparseJacobianView:
    """Problem view that advertises jacobianstructure() and returns values in that order."""

    def __init__(self, core, with_hessian):
        self.core = core
        if getattr(core, "_fused", None) is not None:
            self.objective = lambda x: float(self._ev(x)["f"])
            self.gradient = lambda x: self._ev(x)["grad"].copy()
            self.constraints = lambda x: self._ev(x)["g"].copy()
        else:
            self.objective, self.gradient, self.constraints = core.objective, core.gradient, core.constraints
        self.rows, self.cols = core.jacobianstructure()
        if with_hessian:
            self.hessian, self.hessianstructure = core.hessian, core.hessianstructure

    def jacobianstructure(self):
        return self.rows, self.cols

    def jacobian(self, x):
        # the band values come from the device in jac_structure() order (fused sparse launch): the dense (m,n) matrix
        # is neither assembled nor copied
        fused = getattr(self.core, "_fused", None)
        if fused is not None:
            return fused.evaluate(x, self.core.x0, sparse=True)["jac_sparse"].copy()
        return self.core.jacobian(x)[self.rows, self.cols]

    # with the sparse view every callback of an iterate shares the sparse evaluation (one device call per iterate)
    def _ev(self, x):
        return self.core._fused.evaluate(x, self.core.x0, sparse=True)


class IpoptProblemFactory(ProblemFactory):
    def _process(self):
        return IpoptProblem(self.x0, self.objective, self.constraints, self.integrator, p=self.p, tvp=self.tvp,
                            use_hessian=self.use_hessian)


class Ipopt(Optimizer):
    def __init__(self, max_iteration=500, init_with_last_result=False, mu_strategy="monotone", mu_target=0,
                 mu_linear_decrease_factor=0.2, alpha_for_y="primal", obj_scaling_factor=1,
                 nlp_scaling_max_gradient=100.0, tol=1e-1, acceptable_tol=1e-4, sparse_jacobian=False):
        super().__init__()
        self.max_iteration = max_iteration
        # stored like the reference does; the reference never passes them to Ipopt (ipopt.py:173-183)
        self.mu_strategy = mu_strategy
        self.mu_target = mu_target
        self.mu_linear_decrease_factor = mu_linear_decrease_factor
        self.alpha_for_y = alpha_for_y
        self.obj_scaling_factor = obj_scaling_factor
        self.nlp_scaling_max_gradient = nlp_scaling_max_gradient
        self.tol, self.acceptable_tol = tol, acceptable_tol  # reference hard-wires 1e-1 / 1e-4 (ipopt.py:184-185)
        self.sparse_jacobian = sparse_jacobian
        self.init_with_last_result = init_with_last_result
        self.prev_result = None

    def get_factory(self):
        return IpoptProblemFactory()

    def initial_point(self, problem):
        model = problem.integrator.model
        H = problem.integrator.H
        if self.init_with_last_result and self.prev_result is not None:
            return warm_start_shift(self.prev_result, H, model.x_dim, model.u_dim)
        return cold_start(problem.get_init_value(), H, model.u_dim)

    def solve(self, problem, domain_constraint):
        try:
            import cyipopt
        except ImportError as e:  # third-party solver, not part of this build
            raise ImportError("Ipopt.solve needs the `cyipopt` package (Ipopt bindings); it is not installed. "
                              "Use optimizer.Slsqp() or install cyipopt.") from e
        x_init = self.initial_point(problem)
        H = problem.integrator.H
        lb = domain_constraint.get_lower_bounds(H)
        ub = domain_constraint.get_upper_bounds(H)
        cl = problem.get_constraint_lower_bounds()
        cu = problem.get_constraint_upper_bounds()
        if self.sparse_jacobian:
            view = _SparseJacobianView(problem, problem.use_hessian)
        elif not problem.use_hessian:
            view = ProblemInterfaceHessianFree(problem)
        else:
            view = problem
        nlp = cyipopt.Problem(n=len(x_init), m=len(cl), problem_obj=view, lb=lb, ub=ub, cl=cl, cu=cu)
        add = getattr(nlp, "add_option", None) or nlp.addOption
        add("max_iter", self.max_iteration)
        add("tol", self.tol)
        add("acceptable_tol", self.acceptable_tol)
        add("print_level", 0)
        x, info = nlp.solve(x_init)
        self.prev_result = x
        return Optimizer.SUCCESS if info["status"] in (0, 1) else Optimizer.FAIL
