import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import nempc_oracle as orc
from pyneuralempc_amd import CallbackEngine
for (B, H, box) in ((4, 20, None), (3, 7, None), (37, 20, None), (5, 50, (-2.0, 2.0)), (1024, 20, None)):
    nx, nu = 2, 1
    net = orc.MLP.random(3, [64, 64], 2, seed=0)
    eng = CallbackEngine(net.W, net.b, H, nx, nu, dtype=torch.float64, device="cuda:0", max_batch=B)
    if box: eng.set_box_rows(*box)
    Zh, X0h = orc.synthetic_inputs(B, H, nx, nu, seed=1)
    Z, X0 = eng.to_device(Zh), eng.to_device(X0h)
    out = {"jac_sparse": torch.full((B, eng.nnz_jac), 777.0, dtype=torch.float64, device="cuda:0")}
    res = eng.eval(Z, X0, ("f", "grad", "g", "jac_sparse"), out=out)
    torch.cuda.synchronize()
    k1 = eng.last_row_kernel
    sp = res["jac_sparse"].cpu().numpy().copy()
    full = eng.eval(Z, X0, ("f", "grad", "g", "jac_dense", "jac_tiles", "jac_sparse"))
    torch.cuda.synchronize()
    ref = full["jac_sparse"].cpu().numpy()
    bad = np.argwhere(sp != ref)
    print(B, H, box, k1, eng.last_row_kernel, "nnz", eng.nnz_jac, "bad", len(bad), bad[:12].tolist(), [float(sp[tuple(b)]) for b in bad[:6]], flush=True)
