"""Shared test helpers: load a golden case and rebuild the matching oracle Problem."""
import os

import numpy as np

from oracle import nempc_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASE_NAMES = ["c1_discret", "c2_discret", "c2_unity", "c2_rk4", "c3_rk4", "c3_discret", "c5_box",
              "odd_dims", "h1", "tvp_p_discret", "tvp_p_rk4"]
# activation family: uniform hidden activation + linear output (every kernel family), and per-layer mixes incl. the
# output layer (generic kernel only)
ACT_UNIFORM_NAMES = [f"act_{a}_{c}" for a in ("relu", "sigmoid", "softplus", "elu") for c in ("c2", "c3")]
ACT_MIXED_NAMES = ["act_mixed_box", "act_mixed_rk4", "act_linear_hidden", "act_param_box", "act_selu_rk4"]
# per-layer mixes of the output-based family under a linear output layer, width <= 128, <= 3 hidden layers (4 up to width 64):
# the register-resident matrix-core kernels with run-time activation codes (round 5) -- and every other kernel family
ACT_RUNTIME_NAMES = ["act_mix3_c2", "act_mix2_rk4", "act_mix_c3_rk4", "deep4_mixed_c2", "act_linear_hidden", "act_selu_rk4",
                     "deep4_c2"]
# networks only the layer-at-a-time GEMM path (and the generic kernel) take: width > 128, more than three hidden layers
WIDE_DEEP_NAMES = ["wide256_c2", "deep4_c2", "deep5_mixed_rk4"]
ZBASED_NAMES = ["act_swish_gelu_box", "act_gelu_rk4", "act_zout_box", "act_zout_rk4"]     # swish / gelu ...: the layered path and the generic kernel
ROLLING_NAMES = ["roll2_discret", "roll3_unity_rev", "roll3_discret_rev", "roll4_wide", "roll2_tvp_p", "roll4_short"]


def case_extra(d):
    """(H, tvp_dim + p_dim) extra network inputs of a golden case, [tvp_t | p], or None.  Rolling cases roll the
    time-varying parameters like the states (model/tensorflow.py:218-228): (H, w*tvp_dim + p_dim)."""
    H = int(d["H"])
    parts = []
    if int(d.get("tvp_dim", 0)):
        w = int(d.get("window", 1))
        if w > 1:
            ext = np.concatenate([d["prev_tvp"], d["tvp"]], axis=0)
            order = slice(None) if int(d["forward_rolling"]) else slice(None, None, -1)
            parts.append(np.stack([ext[t:t + w][order].reshape(-1) for t in range(H)], axis=0))
        else:
            parts.append(d["tvp"])
    if int(d.get("p_dim", 0)):
        parts.append(np.tile(d["p"].reshape(1, -1), (H, 1)))
    return np.concatenate(parts, axis=1) if parts else None


def load_case(name):
    d = dict(np.load(os.path.join(GOLDEN, f"{name}.npz")))
    nl = sum(1 for k in d if k.startswith("W"))
    W = [d[f"W{i}"] for i in range(nl)]
    b = [d[f"b{i}"] for i in range(nl)]
    return d, W, b


def case_activations(d):
    """Per-layer activation names of a golden case (None = the default tanh ... linear stack)."""
    if "activations" not in d:
        return None
    names = [orc.ACTIVATIONS[int(c)] for c in d["activations"]]
    if "act_param" in d:            # alpha of the elu / leaky_relu layers ("name:value" specs)
        names = [f"{n}:{float(p)!r}" if n in orc.ACT_DEFAULT_PARAM else n for n, p in zip(names, d["act_param"])]
    return names


def oracle_problem(d, W, b, i=0):
    """Oracle Problem of a golden case; for rolling cases with the history of problem i of the batch."""
    net = orc.MLP(W, b, case_activations(d))
    box = (d["box_lo"], d["box_hi"]) if int(d["has_box"]) else None
    w = int(d.get("window", 1))
    roll = {} if w == 1 else dict(window=w, forward_rolling=bool(int(d["forward_rolling"])),
                                  hist_x=d["hist_x"][i], hist_u=d["hist_u"][i])
    return orc.Problem(net, int(d["H"]), int(d["nx"]), int(d["nu"]), int(d["kind"]), float(d["DT"]),
                       Q=d["Q"], R=d["R"], xref=d["xref"], uref=d["uref"], cu=d["cu"], box=box, extra=case_extra(d),
                       **roll)
