"""Variable bounds and extra constraint rows (reference: pyNeuralEMPC/constraints.py:3-96)."""
import numpy as np


def _bound_column(pairs, which, H):
    """[b[which] for every (lower, upper) pair] repeated for the H steps of the horizon"""
    return [pair[which] for pair in pairs] * H


class DomainConstraint:
    """Box bounds on the decision variables z = [states | controls]; consumed by the optimizer as lb / ub
    (optimizer/ipopt.py:153-154), they add no rows to g."""

    def __init__(self, states_constraint: list, control_constraint: list):
        for pairs, empty_msg, label in ((states_constraint, "States constraint empty !", "states"),
                                        (control_constraint, "Control constraint empty !", "control")):
            if len(pairs) == 0:
                raise ValueError(empty_msg)
            if any(len(pair) != 2 for pair in pairs):
                raise ValueError(f"Your {label} constraint must be a list of bound couple  ! "
                                 "[(lower_bound, upper_bound), ...]")
        self.states_constraint, self.control_constraint = states_constraint, control_constraint

    def get_dim(self, H):
        return len(self.states_constraint), len(self.control_constraint)

    def get_lower_bounds(self, H):
        return _bound_column(self.states_constraint, 0, H) + _bound_column(self.control_constraint, 0, H)

    def get_upper_bounds(self, H):
        return _bound_column(self.states_constraint, 1, H) + _bound_column(self.control_constraint, 1, H)

    def get_type(self):
        return Constraint.EQ_TYPE     # meaningless for bounds and unused, as in the reference (constraints.py:32-33)


class Constraint:
    """Extra rows appended to g after the integrator defects (optimizer/ipopt.py:47-52,91-96).  A subclass gives
    forward (k,), jacobian (k, n), get_dim(H) and the two bound vectors; the row type follows from the bounds."""
    EQ_TYPE, INEQ_TYPE, INTER_TYPE = 0, 1, 2

    def forward(self, x, u, p=None, tvp=None):
        pass

    def jacobian(self, x, u, p=None, tvp=None):
        pass

    def hessian(self, x, u, p=None, tvp=None):
        """(k, n, n); called by the solver glue (ipopt.py:75) although the reference's base class does not declare
        it.  Default: linear rows, all zeros."""
        H, nx = np.asarray(x).shape
        n = H * (nx + np.asarray(u).shape[1])
        return np.zeros((int(self.get_dim(H)), n, n))

    def get_dim(self, H):
        raise NotImplementedError()

    def get_lower_bounds(self, H):
        raise NotImplementedError()

    def get_upper_bounds(self, H):
        raise NotImplementedError()

    def get_type(self, H=None):
        lo, hi = (np.asarray(v) for v in (self.get_lower_bounds(H), self.get_upper_bounds(H)))
        if not (lo == 0).all():
            return Constraint.INTER_TYPE
        if (hi == 0).all():
            return Constraint.EQ_TYPE
        return Constraint.INEQ_TYPE if (hi == np.inf).all() else Constraint.INTER_TYPE


class _ZeroLowerBound(Constraint):
    """rows bounded below by zero; the two public flavours differ in the upper bound only"""
    _upper = 0.0

    def forward(self, x, u, p=None, tvp=None):
        raise NotImplementedError()

    def jacobian(self, x, u, p=None, tvp=None):
        raise NotImplementedError()

    def get_lower_bounds(self, H):
        return np.zeros(int(self.get_dim(H)))

    def get_upper_bounds(self, H):
        return np.full(int(self.get_dim(H)), self._upper)


class EqualityConstraint(_ZeroLowerBound):       # g(z) = 0
    _upper = 0.0


class InequalityConstraint(_ZeroLowerBound):     # g(z) >= 0
    _upper = np.inf


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
