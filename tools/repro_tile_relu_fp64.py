import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
nx, nu, H, B = 2, 1, 7, 64
net = orc.MLP.random(nx + nu, [96] * 2, nx, seed=5, activations="relu")
Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=5)
prob = orc.Problem(net, H, nx, nu, orc.DISCRET, 1.0)
ref = np.stack([prob.tiles(Zh[i], X0h[i])[2] for i in range(B)])
f, grad, g, J = prob.eval_batch(Zh, X0h)
eng = CallbackEngine(net.W, net.b, H, nx, nu, integrator="discret", dtype=torch.float64, device="cuda:0", max_batch=B, kernel="mfma_tile", activations="relu")
Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
for want in (("g", "jac_tiles"), ("g", "jac_dense"), ("g", "jac_tiles", "jac_dense"), ("jac_tiles",), ("g", "jac_tiles")):
    nb = []
    for rep in range(10):
        res = eng.eval(Z, X0, want)
        torch.cuda.synchronize()
        bad = 0
        if "jac_tiles" in res: bad += int((np.abs(res["jac_tiles"].cpu().numpy() - ref) > 1e-9).sum())
        if "jac_dense" in res: bad += int((np.abs(res["jac_dense"].cpu().numpy() - J) > 1e-9).sum())
        nb.append(bad)
    print(want, "bad entries per launch:", nb, flush=True)
for mode in ("eval_numpy", "temporaries+eval", "persistent"):
    nb = []
    for rep in range(12):
        if mode == "eval_numpy":
            T = eng.eval_numpy(Zh, X0h, want=("g", "jac_tiles"))["jac_tiles"]
        elif mode == "temporaries+eval":
            res = eng.eval(eng.to_device(Zh), eng.to_device(X0h), ("g", "jac_tiles")); torch.cuda.synchronize(); T = res["jac_tiles"].cpu().numpy()
        else:
            res = eng.eval(Z, X0, ("g", "jac_tiles")); torch.cuda.synchronize(); T = res["jac_tiles"].cpu().numpy()
        nb.append(int((np.abs(T - ref) > 1e-9).sum()))
    print(mode, nb, flush=True)
