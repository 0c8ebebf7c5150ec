#!/usr/bin/env python3
"""Gauss-Newton callback timing probe: with / without row weights, B = 256 / 1024, direct library calls vs the engine wrapper."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
from tools.quick_bench import timed

for B in (256, 1024):
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    eng = CallbackEngine(net.W, net.b, 20, 2, 1, dtype=torch.float64, device="cuda:0", max_batch=B)
    Zh, X0h = orc.synthetic_inputs(B, 20, 2, 1, seed=1)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    rng = np.random.default_rng(7)
    sig = eng.to_device(rng.uniform(0.5, 1.5, size=B)); w = eng.to_device(rng.uniform(0.2, 1.5, size=(B, 40)))
    lam = eng.to_device(rng.normal(size=(B, eng.m)))
    print(B, "gn w=None", timed(lambda: eng.hess_gn(Z, X0, None, sig), 400), "gn w", timed(lambda: eng.hess_gn(Z, X0, w, sig), 400),
          "exact", timed(lambda: eng.hess(Z, X0, lam, sig), 400),
          "gn w unprimed", timed(lambda: eng.hess_gn(Z, X0, w, sig), 50, prime_ms=0), flush=True)
