#!/usr/bin/env python3
"""Latency of the B=1 drop-in path (BASELINE configs[0] shape: H=10, 2/1 MLP 2x30, Discret, SLSQP on the CPU driving the
device callbacks): milliseconds per NMPC.next and microseconds per fused callback evaluation, host copies included."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyneuralempc_amd as nEMPC
from oracle import nempc_oracle as orc
H, nx, nu = 10, 2, 1
net = orc.MLP.random(nx + nu, [30, 30], nx, seed=0); net.W[-1] *= 0.2; net.b[-1] *= 0.2
model = nEMPC.model.MLPModel(net.W, net.b, nx, nu, device="cuda:0")
integ = nEMPC.integrator.discret.DiscretIntegrator(model, H)
obj = nEMPC.objective.QuadraticObjective(Q=np.eye(nx), R=0.1 * np.eye(nu), device="cuda:0")
dom = nEMPC.constraints.DomainConstraint([[-5.0, 5.0]] * nx, [[-1.0, 1.0]])
opt = nEMPC.optimizer.Slsqp(max_iteration=200, tolerance=1e-10, verbose=0, init_with_last_result=True)
mpc = nEMPC.controller.NMPC(integ, obj, [dom], H, 1.0, optimizer=opt)
x = np.array([0.7, -0.4])
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mpc.next(x)
    pb = mpc.get_pb(x)
    n0 = pb._fused.n_device_evals
    t0 = time.perf_counter(); steps = 20
    for _ in range(steps):
        states, u = mpc.next(x)
        x = states[0]
    dt = time.perf_counter() - t0
    z = np.concatenate([states.ravel(), u.ravel()])
    t1 = time.perf_counter()
    for i in range(200):
        pb._fused._key = None
        pb.constraints(z + 1e-9 * i)
    te = (time.perf_counter() - t1) / 200
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem, _SparseJacobianView
    ip = IpoptProblem(x, obj, [], integ)
    sv = _SparseJacobianView(ip, True)
    lam = np.ones(ip._fused.engine.m)
    for i in range(20):       # first calls create the pinned buffers and streams of these two paths: not a per-callback cost
        ip._fused._key = None
        sv.jacobian(z - 1e-9 * (i + 1))
        ip.hessian(z - 1e-9 * (i + 1), lam, 1.0)
    t2 = time.perf_counter()
    for i in range(200):
        ip._fused._key = None
        sv.jacobian(z + 1e-9 * i)
    ts = (time.perf_counter() - t2) / 200
    lam = np.ones(ip._fused.engine.m)
    t3 = time.perf_counter()
    for i in range(200):
        ip.hessian(z + 1e-9 * i, lam, 1.0)
    th = (time.perf_counter() - t3) / 200
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mpc_d = nEMPC.controller.NMPC(integ, obj, [nEMPC.constraints.DomainConstraint([[-5.0, 5.0]] * nx, [[-1.0, 1.0]])], H, 1.0,
                                  optimizer=nEMPC.optimizer.DeviceSqp(init_with_last_result=True))
    xd = np.array([0.7, -0.4])
    mpc_d.next(xd)
    t4 = time.perf_counter()
    for _ in range(steps):
        sd, ud = mpc_d.next(xd)
        xd = sd[0]
    td = (time.perf_counter() - t4) / steps
print(f"NMPC.next (DeviceSqp, warm-started, the whole solve on the device): {td * 1e3:.2f} ms per MPC step, "
      f"{mpc_d.optimizer.last_iterations} iterations in the last one")
print(f"sparse-Jacobian callback (band values from the device): {ts * 1e6:.0f} us; Lagrangian-Hessian callback: {th * 1e6:.0f} us")
print(f"NMPC.next (warm-started SLSQP): {dt / steps * 1e3:.2f} ms per MPC step; fused callback evaluation incl. host copies: {te * 1e6:.0f} us")
