"""NMPC facade (reference: pyNeuralEMPC/controller.py:7-113): argument checks, problem assembly
through the optimizer's factory, solve, reshape."""
import numpy as np

from .constraints import DomainConstraint
from .optimizer import Ipopt, Optimizer


class NMPC:
    def __init__(self, integrator, objective_func, constraint_list, H, DT, optimizer=None, use_hessian=True):
        self.integrator = integrator
        self.objective_func = objective_func
        self.constraint_list = constraint_list
        domains = [c for c in constraint_list if isinstance(c, DomainConstraint)]
        if not domains:
            raise IndexError("constraint_list must contain a DomainConstraint")
        # like the reference (controller.py:13-14) the first DomainConstraint is taken OUT of the caller's list
        self.domain_constraint = domains[0]
        self.constraint_list.remove(self.domain_constraint)
        self.H = H
        self.DT = DT
        # the reference's default argument `optimizer=Ipopt()` is one instance shared by every NMPC
        # (controller.py:8); a fresh one per controller is used here
        self.optimizer = Ipopt() if optimizer is None else optimizer
        # stored and, as in the reference, NOT forwarded to the problem factory (controller.py:18,
        # optimizer/base.py:78): the solve runs hessian-free unless the optimizer opts in
        self.use_hessian = use_hessian

    def _check(self, x0, p, tvp, init_x, init_u):
        model = self.integrator.model
        assert len(x0.shape) == 1, "x0 must be a vector"
        assert x0.shape[0] == model.x_dim, "x0 dim must set according to your model !"
        if p is not None:
            assert len(p.shape) == 1, "p must be a vector"
            assert p.shape[0] == model.p_dim, "p dim must set according to your model !"
        if tvp is not None:
            assert len(tvp.shape) == 2, "tvp must be a vector"
            assert tvp.shape[1] == model.tvp_dim, "tvp dim must set according to your model !"
            assert tvp.shape[0] == self.H, "tvp first dim must set according to the horizon size !"
        assert (init_x is None) == (init_u is None), "you must give both init values"
        if init_x is not None:
            assert init_x.shape[1] == model.x_dim, ("init_x dim must have the good feature size "
                                                    f"(expected {model.x_dim})")
            assert init_u.shape[1] == model.u_dim, ("init_u dim mist have the good feature size "
                                                    f"(expected {model.u_dim})")

    def get_pb(self, x0: np.array, p=None, tvp=None, init_x=None, init_u=None):
        self._check(x0, p, tvp, init_x, init_u)
        factory = self.optimizer.get_factory()
        factory.set_x0(x0)
        factory.set_objective(self.objective_func)
        factory.set_integrator(self.integrator)
        factory.set_constraints(self.constraint_list)
        if init_x is not None:
            factory.set_init_values(init_x, init_u)
        if tvp is not None:
            factory.set_tvp(tvp)
        if p is not None:
            factory.set_p(p)
        if getattr(self.optimizer, "exact_hessian", False):
            factory.set_use_hessian(bool(self.use_hessian))
        return factory.getProblemInterface()

    def next(self, x0: np.array, p=None, tvp=None, init_x=None, init_u=None):
        """Solve one MPC problem; returns (states (H,nx), u (H,nu)) or (None, None) on solver failure."""
        pb = self.get_pb(x0, p=p, tvp=tvp, init_x=init_x, init_u=init_u)
        status = self.optimizer.solve(pb, self.domain_constraint)
        if status != Optimizer.SUCCESS:
            return None, None
        model = self.integrator.model
        z = self.optimizer.prev_result
        nxh = model.x_dim * self.integrator.H
        return z[:nxh].reshape(self.integrator.H, -1), z[nxh:].reshape(self.integrator.H, -1)


    def next_batch(self, X0, init_z=None, p=None, tvp=None, prev_x=None, prev_u=None, prev_tvp=None, **solver_opts):
        """Solve B MPC problems at once on the device (no reference counterpart).  X0 (B,nx) NumPy
        array or device tensor.  Needs the fused device path (device integrator + QuadraticObjective; the only
        extra rows accepted are BoxStateConstraint, which become bounds on the state variables).  Models with
        parameters take p (p_dim,) or (B,p_dim) and tvp (H,tvp_dim) or (B,H,tvp_dim) -- one set for all problems or one
        per problem (the batched form of NMPC.next's p / tvp, controller.py:65).  Rolling-window models take the histories
        prev_x (w-1, nx) or (B, w-1, nx), prev_u, prev_tvp -- the batched form of set_prev_data (model/tensorflow.py:178-189);
        left out, what set_prev_data stored serves every problem.  Returns (states (B,H,nx), u (B,H,nu),
        status (B,) with Optimizer.SUCCESS / FAIL per problem) as NumPy arrays."""
        import torch
        from .optimizer.base import fused_evaluator
        from .objective.quadratic import QuadraticObjective
        from .integrator.base import DeviceIntegrator
        from .constraints import BoxStateConstraint
        boxes = [c for c in self.constraint_list if isinstance(c, BoxStateConstraint)]
        if not (isinstance(self.integrator, DeviceIntegrator) and self.integrator.on_device and isinstance(self.objective_func, QuadraticObjective)
                and len(boxes) == len(self.constraint_list)):
            raise NotImplementedError("next_batch needs a device integrator, a QuadraticObjective and no extra "
                                      "constraint rows other than BoxStateConstraint")
        eng = fused_evaluator(self.integrator, self.objective_func, None).engine
        H = self.integrator.H
        X0t = X0 if isinstance(X0, torch.Tensor) else eng.to_device(np.atleast_2d(np.asarray(X0, dtype=np.float64)))
        B = int(X0t.shape[0])
        model = self.integrator.model
        w = int(getattr(model, "rolling_window", 1))

        def history(given, stored, d, name):
            v = stored if given is None else given
            if v is None:
                raise ValueError(f"rolling-window model: pass {name} or call set_prev_data first")
            v = np.asarray(v, dtype=np.float64)
            if v.shape[-2:] != (w - 1, d) or v.ndim not in (2, 3) or (v.ndim == 3 and v.shape[0] != B):
                raise ValueError(f"{name} must have shape {(w - 1, d)} or {(B, w - 1, d)} (received : {v.shape})")
            return np.ascontiguousarray(np.broadcast_to(v if v.ndim == 3 else v[None], (B, w - 1, d)))

        if w > 1:
            eng.bind_history(eng.to_device(history(prev_x, model.prev_x, model.x_dim, "prev_x")),
                             eng.to_device(history(prev_u, model.prev_u, model.u_dim, "prev_u")))
        elif prev_x is not None or prev_u is not None or prev_tvp is not None:
            raise ValueError("this model has no rolling window: it takes no prev_x / prev_u / prev_tvp")
        if model.p_dim + model.tvp_dim:
            # extra network inputs of every problem: (B, H, tvp_dim + p_dim) = [tvp_t | p], the reference's
            # concatenation order (model/tensorflow.py:39-47); a (1, ...) binding left by NMPC.next never serves B > 1
            parts = []
            if model.tvp_dim:
                if tvp is None:
                    raise ValueError("this model has tvp_dim > 0: pass tvp (H, tvp_dim) or (B, H, tvp_dim)")
                tv = np.asarray(tvp, dtype=np.float64)
                tv = np.broadcast_to(tv if tv.ndim == 3 else tv[None], (B, H, model.tvp_dim))
                if w > 1:      # rolled like the states: (B, H, w * tvp_dim)  (_gather_input_V2, model/tensorflow.py:218-233)
                    ptv = history(prev_tvp, model.prev_tvp, model.tvp_dim, "prev_tvp")
                    tv = np.stack([model._roll(ptv[b], tv[b]) for b in range(B)], axis=0)
                parts.append(tv)
            if model.p_dim:
                if p is None:
                    raise ValueError("this model has p_dim > 0: pass p (p_dim,) or (B, p_dim)")
                pv = np.asarray(p, dtype=np.float64)
                pv = np.broadcast_to(pv if pv.ndim == 2 else pv[None], (B, model.p_dim))
                parts.append(np.broadcast_to(pv[:, None, :], (B, H, model.p_dim)))
            eng.bind_extra(eng.to_device(np.concatenate(parts, axis=2)))
        elif p is not None or tvp is not None:
            raise ValueError("this model takes no p / tvp")
        Zi = None if init_z is None else (init_z if isinstance(init_z, torch.Tensor) else eng.to_device(init_z))
        lb = np.asarray(self.domain_constraint.get_lower_bounds(H), dtype=np.float64).copy()
        ub = np.asarray(self.domain_constraint.get_upper_bounds(H), dtype=np.float64).copy()
        # box rows on the states are bounds on the state variables: intersect them with the domain
        for box in boxes:
            lo, hi = box._bounds(eng.nx)
            lb[:H * eng.nx] = np.maximum(lb[:H * eng.nx], np.tile(lo, H))
            ub[:H * eng.nx] = np.minimum(ub[:H * eng.nx], np.tile(hi, H))
        Z, status, iters = eng.solve(X0t.contiguous(), Zi, lb, ub, **solver_opts)
        self.last_batch_iterations = iters
        z = Z.to("cpu", torch.float64).numpy()
        B, nx, nu = z.shape[0], eng.nx, eng.nu
        return z[:, :H * nx].reshape(B, H, nx), z[:, H * nx:].reshape(B, H, nu), status.cpu().numpy()


MPC = NMPC  # BASELINE.json's north_star calls the entry point controller.MPC
