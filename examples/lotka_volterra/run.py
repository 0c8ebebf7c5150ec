#!/usr/bin/env python3
"""Lotka-Volterra economic NMPC on the MI355X path -- the counterpart of the reference's
examples/lotka_volterra/run.py:64-98 (same plug-in calls, same normalisation, horizon, bounds and cost).

What differs, and why: the reference loads a Keras file (`nn_model.h5`, needs TensorFlow + h5py) and an external
simulator package; neither exists here, so the 3 -> 30 tanh -> 30 tanh -> 2 surrogate of the normalised vector field
is fitted in a few seconds with torch on the CPU and the plant is the analytic ODE integrated with RK4.  Everything
from `KerasTFModel`-style model construction to `NMPC.next` is the drop-in surface; the callbacks run in the HIP
kernels (needs a GPU; SLSQP drives them because cyipopt is not installed).

    python examples/lotka_volterra/run.py --steps 20            # closed loop, one plant
    python examples/lotka_volterra/run.py --steps 5 --batch 256 # 256 plants solved in lock step on the device
    python examples/lotka_volterra/run.py --steps 20 --device-solver   # one plant, the whole solve on the device
"""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pyneuralempc_amd as nEMPC  # noqa: E402

# run.py:38-52 -- everything is normalised: x = x/30 - 1, y = y/30 - 1, u = u/50 - 1
Hb, DT = 25, 0.05
X_MAX = 60 / 30.0 - 1
U_MAX, U_MIN = 60 / 50.0 - 1, 0 / 50.0 - 1


def vector_field(xn, un):
    """normalised d/dt of (x, y) under control u: run.py:66-68 in physical units, rescaled"""
    x, y, u = 30.0 * (xn[..., 0] + 1.0), 30.0 * (xn[..., 1] + 1.0), 50.0 * (un[..., 0] + 1.0)
    dx = 0.5 * x - 0.025 * x * y
    dy = -0.5 * y + u + 0.005 * x * y
    return np.stack([dx, dy], axis=-1) / 30.0


def fit_surrogate(iters=1500, seed=0):
    """3 -> 30 tanh -> 30 tanh -> 2, the architecture of the reference's nn_model.h5"""
    g = torch.Generator().manual_seed(seed)
    with torch.random.fork_rng(devices=[]):       # the layers draw their initial weights from the global generator
        torch.manual_seed(seed)
        net = torch.nn.Sequential(torch.nn.Linear(3, 30), torch.nn.Tanh(), torch.nn.Linear(30, 30), torch.nn.Tanh(),
                                  torch.nn.Linear(30, 2)).double()
    xi = torch.rand(4096, 3, generator=g, dtype=torch.float64) * 2.0 - 1.0
    xi[:, 2] = xi[:, 2] * 0.5 * (U_MAX - U_MIN) + 0.5 * (U_MAX + U_MIN)
    target = torch.from_numpy(vector_field(xi[:, :2].numpy(), xi[:, 2:].numpy()))
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    for _ in range(iters):
        opt.zero_grad()
        loss = torch.mean((net(xi) - target) ** 2)
        loss.backward()
        opt.step()
    return net.eval(), float(loss.detach())


def plant_step(x, u, dt):
    f = lambda s: vector_field(s, u)
    k1 = f(x); k2 = f(x + 0.5 * dt * k1); k3 = f(x + 0.5 * dt * k2); k4 = f(x + dt * k3)
    return x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)


def main(steps=20, batch=0, fit_iters=1500, device="cuda:0", verbose=True, device_solver=False):
    net, loss = fit_surrogate(fit_iters)
    if verbose:
        print(f"surrogate fitted, mse {loss:.2e}")
    H = Hb
    # run.py:73 hands the loaded Keras model to KerasTFModel; the torch module goes to its counterpart the same way
    model_nmpc = nEMPC.model.TorchMLPModel(net, x_dim=2, u_dim=1, device=device)
    constraints_nmpc = [nEMPC.constraints.DomainConstraint(                            # run.py:77-79
        states_constraint=[[-np.inf, X_MAX], [-np.inf, np.inf]], control_constraint=[[U_MIN, U_MAX]])]
    integrator = nEMPC.integrator.rk4.RK4Integrator(model_nmpc, H, 0.1, cache_mode=True)   # run.py:82
    # run.py:84-93: cost = sum(u * 1.1) -- the linear member of the quadratic family
    objective_func = nEMPC.objective.QuadraticObjective(Q=np.zeros((2, 2)), R=np.zeros((1, 1)), cu=np.full((H, 1), 1.1),
                                                        device=device)
    optimizer = (nEMPC.optimizer.DeviceSqp(max_iteration=200, tolerance=1e-8, init_with_last_result=True) if device_solver
                 else nEMPC.optimizer.Slsqp(max_iteration=200, tolerance=1e-8, verbose=0, init_with_last_result=True))
    MPC = nEMPC.controller.NMPC(integrator, objective_func, constraints_nmpc, H, DT, optimizer=optimizer)   # run.py:96

    if batch:
        rng = np.random.default_rng(0)
        X = np.stack([np.array([0.66, -0.9]) + 0.05 * rng.normal(size=2) for _ in range(batch)])
        for k in range(steps):
            # the batched solver keeps the iterates strictly inside their bounds: start from an interior control
            z0 = np.concatenate([np.tile(X, (1, H)), np.full((batch, H), 0.5 * (U_MIN + U_MAX))], axis=1)
            states, u, status = MPC.next_batch(X, init_z=z0, max_iter=60)
            X = plant_step(X, u[:, 0, :], DT)
            if verbose:
                print(f"step {k}: {int((status == 0).sum())}/{batch} solved, mean u0 {u[:, 0, 0].mean():+.3f}, "
                      f"max x {X[:, 0].max():+.3f} (limit {X_MAX:+.3f})")
        return X
    curr_x = np.array([0.66, -0.9])                                                     # run.py:55
    traj = [curr_x]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(steps):
            pred, u = MPC.next(curr_x)                                                  # run.py:98 (order: controller.py:110)
            if pred is None:
                raise RuntimeError("solver failed")
            curr_x = plant_step(curr_x, u[0], DT)
            traj.append(curr_x)
            if verbose:
                print(f"step {k}: u0 {u[0, 0]:+.3f}  x {curr_x[0]:+.3f} y {curr_x[1]:+.3f}  (x limit {X_MAX:+.3f})")
    return np.array(traj)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--fit-iters", type=int, default=1500)
    ap.add_argument("--device-solver", action="store_true", help="NMPC.next with optimizer.DeviceSqp instead of SLSQP")
    a = ap.parse_args()
    main(a.steps, a.batch, a.fit_iters, device_solver=a.device_solver)
