#!/usr/bin/env python3
"""Stress of the wave-per-tile row kernel's instantiations: every (dtype, padded width, hidden layers) x activation x
transcription, repeated launches over a batch of 28 tiles, every launch checked against the oracle.
python tools/tile_stress.py [reps]"""
import os, sys, itertools, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
kinds = {"discret": orc.DISCRET, "rk4": orc.RK4}
nx, nu, H, B = 2, 1, 7, 64
summary = {}
for dt, width, depth in itertools.product((torch.float64, torch.float32), (24, 48, 96), (1, 2, 3)):
    for integ, act in itertools.product(("discret", "rk4"), ("tanh", "relu", "sigmoid", "softplus", "elu")):
        DT = 0.1 if integ == "rk4" else 1.0
        net = orc.MLP.random(nx + nu, [width] * depth, nx, seed=5, activations=act)
        Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
        prob = orc.Problem(net, H, nx, nu, kinds[integ], DT)
        f, grad, g, J = prob.eval_batch(Zh, X0h)
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT, dtype=dt, device="cuda:0", max_batch=B, kernel="mfma_tile", activations=act)
        tol = 2e-4 if dt == torch.float32 else 1e-10
        nbad = 0
        for r in range(reps):
            res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
            e = max(np.abs(res["g"] - g).max() / max(1, np.abs(g).max()), np.abs(res["jac_dense"] - J).max() / max(1, np.abs(J).max()))
            nbad += int(not e < tol)
        key = (str(dt)[6:], width, depth)
        summary.setdefault(key, []).append((integ, act, nbad))
        del eng
for key, rows in summary.items():
    bad = [(i, a, n) for i, a, n in rows if n]
    print(key, "OK" if not bad else f"BAD {bad}", flush=True)
