import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
nx, nu, width, depth, B, H, DT = 2, 1, 128, 2, 4, 16, 0.1
for act in ("tanh", "relu", "sigmoid", "softplus", "elu"):
    for integ, kind in (("rk4", orc.RK4), ("discret", orc.DISCRET)):
        net = orc.MLP.random(nx + nu, [width] * depth, nx, seed=36, activations=act)
        Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=36)
        prob = orc.Problem(net, H, nx, nu, kind, DT if integ == "rk4" else 1.0)
        f, grad, g, J = prob.eval_batch(Zh, X0h)
        eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator=integ, DT=DT if integ == "rk4" else 1.0, dtype=torch.float32, device="cuda:0",
                             max_batch=B, kernel="mfma_tile", activations=act)
        res = eng.eval_numpy(Zh, X0h, want=("g", "jac_dense"))
        print(f"{act:9s} {integ:8s}: g err {np.abs(res['g'] - g).max():.2e} jac err {np.abs(res['jac_dense'] - J).max():.2e}", flush=True)
        del eng
