"""Variable bounds and extra constraint rows (reference: pyNeuralEMPC/constraints.py:3-96)."""
import numpy as np


class DomainConstraint:
    """Box bounds on the decision variables, consumed by the optimizer only (no rows in g)."""

    def __init__(self, states_constraint: list, control_constraint: list):
        if len(states_constraint) == 0:
            raise ValueError("States constraint empty !")
        if len(control_constraint) == 0:
            raise ValueError("Control constraint empty !")
        for name, seq in (("states", states_constraint), ("control", control_constraint)):
            if any(len(c) != 2 for c in seq):
                raise ValueError(f"Your {name} constraint must be a list of bound couple  ! "
                                 "[(lower_bound, upper_bound), ...]")
        self.states_constraint = states_constraint
        self.control_constraint = control_constraint

    def get_dim(self, H):
        return len(self.states_constraint), len(self.control_constraint)

    def get_lower_bounds(self, H):
        return [c[0] for c in self.states_constraint] * H + [c[0] for c in self.control_constraint] * H

    def get_upper_bounds(self, H):
        return [c[1] for c in self.states_constraint] * H + [c[1] for c in self.control_constraint] * H

    def get_type(self):
        return Constraint.EQ_TYPE


class Constraint:
    """Extra rows appended after the integrator defects (optimizer/ipopt.py:47-52,91-96)."""
    EQ_TYPE = 0
    INEQ_TYPE = 1
    INTER_TYPE = 2

    def forward(self, x, u, p=None, tvp=None):
        pass

    def jacobian(self, x, u, p=None, tvp=None):
        pass

    def hessian(self, x, u, p=None, tvp=None):
        """(k, n, n); called by the solver glue (ipopt.py:75). Linear rows: zeros."""
        H = np.asarray(x).shape[0]
        n = H * (np.asarray(x).shape[1] + np.asarray(u).shape[1])
        return np.zeros((int(self.get_dim(H)), n, n))

    def get_dim(self, H):
        raise NotImplementedError()

    def get_lower_bounds(self, H):
        raise NotImplementedError()

    def get_upper_bounds(self, H):
        raise NotImplementedError()

    def get_type(self, H=None):
        lo, hi = np.asarray(self.get_lower_bounds(H)), np.asarray(self.get_upper_bounds(H))
        if (hi == lo).all() and (lo == 0).all():
            return Constraint.EQ_TYPE
        if (hi == np.inf).all() and (lo == 0).all():
            return Constraint.INEQ_TYPE
        return Constraint.INTER_TYPE


class EqualityConstraint(Constraint):
    def forward(self, x, u, p=None, tvp=None):
        raise NotImplementedError()

    def jacobian(self, x, u, p=None, tvp=None):
        raise NotImplementedError()

    def get_lower_bounds(self, H):
        return np.zeros(int(self.get_dim(H)))

    def get_upper_bounds(self, H):
        return np.zeros(int(self.get_dim(H)))


class InequalityConstraint(Constraint):
    def forward(self, x, u, p=None, tvp=None):
        raise NotImplementedError()

    def jacobian(self, x, u, p=None, tvp=None):
        raise NotImplementedError()

    def get_lower_bounds(self, H):
        return np.zeros(int(self.get_dim(H)))

    def get_upper_bounds(self, H):
        return np.ones(int(self.get_dim(H))) * np.inf


class BoxStateConstraint(Constraint):
    """rows = states.ravel() kept in [lo, hi] (BASELINE config 5: box state constraints as rows of g,
    dense selector Jacobian).  The reference declares the interface but ships no concrete row
    constraint; this one is recognised by the solver glue and evaluated inside the fused device call.
    The NumPy methods below are the interface's contract for third-party glue."""

    def __init__(self, lo, hi, x_dim=None):
        lo, hi = np.atleast_1d(np.asarray(lo, dtype=np.float64)), np.atleast_1d(np.asarray(hi, dtype=np.float64))
        if x_dim is not None:
            lo, hi = np.broadcast_to(lo, (x_dim,)).copy(), np.broadcast_to(hi, (x_dim,)).copy()
        elif lo.size == 1 or hi.size == 1:
            raise ValueError("give one (lo, hi) per state, or scalars together with x_dim")
        if lo.shape != hi.shape:
            raise ValueError("lo and hi must have the same shape")
        self.lo, self.hi = lo, hi

    def _bounds(self, nx):
        return np.broadcast_to(self.lo, (nx,)), np.broadcast_to(self.hi, (nx,))

    def forward(self, x, u, p=None, tvp=None):
        return np.asarray(x, dtype=np.float64).reshape(-1).copy()

    def jacobian(self, x, u, p=None, tvp=None):
        H, nx = np.asarray(x).shape
        nu = np.asarray(u).shape[1]
        return np.concatenate([np.eye(H * nx), np.zeros((H * nx, H * nu))], axis=1)

    def get_dim(self, H):
        return H * len(self.lo)

    def get_lower_bounds(self, H):
        return np.tile(self.lo, H)

    def get_upper_bounds(self, H):
        return np.tile(self.hi, H)
