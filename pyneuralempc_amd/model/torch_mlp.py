"""A dense network given as a ``torch.nn`` module, evaluated by the HIP kernels -- the counterpart of ``KerasTFModel`` (reference:
model/tensorflow.py:8-109 wraps a live Keras model) for the framework this image does have.  The module is only *read*: the
``nn.Linear`` weights and the activation modules between them are copied once (``extract_linear_stack``); the network then runs
on the same kernels as ``MLPModel`` / ``KerasTFModel``, not through torch.  A module the kernels cannot express (convolutions,
recurrences, skip connections, custom ``forward`` code) belongs in ``TorchModel``, which differentiates any torch callable on the
host side.

    net = torch.nn.Sequential(torch.nn.Linear(3, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 2))
    model = TorchMLPModel(net, x_dim=2, u_dim=1)
"""
import numpy as np

from .mlp import MLPModel
from .rolling import MLPModelRollingInput
from .tensorflow import extract_dense_stack


# ---- duck-typed "Keras layers" for extract_dense_stack: the same walk, folding rules and refusals serve both adapters ----
class Dense:
    use_bias = True

    def __init__(self, W, b):
        self._p, self.activation = [W, b], "linear"

    def get_weights(self):
        return self._p


class Activation:
    def __init__(self, name):
        self.activation = name

    def get_weights(self):
        return []


class Dropout:
    def get_weights(self):
        return []


class BatchNormalization:
    def __init__(self, gamma, beta, mean, var, eps):
        self.scale, self.center, self.epsilon = gamma is not None, beta is not None, eps
        self._p = ([gamma] if gamma is not None else []) + ([beta] if beta is not None else []) + [mean, var]

    def get_weights(self):
        return self._p


def _np(t):
    return t.detach().to("cpu").double().numpy()


def _as_layers(module, path="module"):
    """torch.nn module tree -> the layer sequence extract_dense_stack walks; containers are flattened in order"""
    import torch.nn as nn
    simple = {nn.Tanh: "tanh", nn.ReLU: "relu", nn.Sigmoid: "sigmoid", nn.SELU: "selu", nn.SiLU: "swish", nn.Softsign: "softsign",
              nn.Mish: "mish", nn.ReLU6: "relu6"}
    if isinstance(module, nn.Sequential):
        out = []
        for name, child in module.named_children():
            out += _as_layers(child, f"{path}.{name}")
        return out
    if isinstance(module, nn.Linear):
        W = _np(module.weight).T.copy()                      # torch keeps (out, in); the kernels take Keras' (in, out)
        b = _np(module.bias) if module.bias is not None else np.zeros(W.shape[1])
        return [Dense(W, b)]
    for cls, name in simple.items():
        if type(module) is cls:
            return [Activation(name)]
    if isinstance(module, nn.Softplus):
        if float(module.beta) != 1.0:
            raise NotImplementedError(f"{path}: Softplus with beta != 1 is unsupported on the device path")
        return [Activation("softplus")]      # (torch switches to the identity above `threshold` = 20: a 2e-9 difference)
    if isinstance(module, nn.ELU):
        return [Activation("elu" if float(module.alpha) == 1.0 else f"elu:{float(module.alpha)!r}")]
    if isinstance(module, nn.LeakyReLU):
        return [Activation(f"leaky_relu:{float(module.negative_slope)!r}")]
    if isinstance(module, nn.GELU):
        if getattr(module, "approximate", "none") != "none":
            raise NotImplementedError(f"{path}: GELU(approximate='tanh') is unsupported on the device path (the kernels have the erf form)")
        return [Activation("gelu")]
    if isinstance(module, (nn.Identity, nn.Dropout, nn.AlphaDropout, nn.Flatten)):
        return [Dropout()]
    if isinstance(module, nn.BatchNorm1d):
        if module.running_mean is None or module.training:
            raise NotImplementedError(f"{path}: BatchNorm1d needs its running statistics and eval() mode to be folded into a Linear layer")
        return [BatchNormalization(_np(module.weight) if module.affine else None, _np(module.bias) if module.affine else None,
                                   _np(module.running_mean), _np(module.running_var), float(module.eps))]
    raise NotImplementedError(f"{path}: module '{type(module).__name__}' is unsupported on the device path (supported: Sequential, Linear, "
                              "the activation modules of the kernels' family, BatchNorm1d in eval mode, Identity / Dropout); wrap the "
                              "function in TorchModel instead")


class _Stack:
    def __init__(self, layers):
        self.layers = layers


def extract_linear_stack(module):
    """-> (weights (in, out), biases, activation names) of a torch.nn.Sequential-shaped dense network; activation modules fold
    into the Linear in front of them, BatchNorm1d (eval) into the neighbouring Linear, anything else is refused"""
    return extract_dense_stack(_Stack(_as_layers(module)))


def _check_dims(weights, x_dim, n_in):
    if weights[-1].shape[1] != x_dim:
        raise ValueError("Your torch module do not provide a suitable output dim ! \n It must get the same dim as the state dim.")
    if weights[0].shape[0] != n_in:
        raise ValueError("Your torch module do not provide a suitable input dim ! \n It must get the same dim as the sum of all "
                         "input vars (x, u, p, tvp).")


class TorchMLPModel(MLPModel):
    def __init__(self, module, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, **device_kwargs):
        weights, biases, activations = extract_linear_stack(module)
        _check_dims(weights, x_dim, sum((x_dim, u_dim, p_dim, tvp_dim)))
        super().__init__(weights, biases, x_dim, u_dim, p_dim, tvp_dim, activations=activations, **device_kwargs)


class TorchMLPModelRollingInput(MLPModelRollingInput):
    """The rolling-window counterpart (reference: KerasTFModelRollingInput, model/tensorflow.py:132-340): the module reads
    rolling_window * (x_dim + u_dim + tvp_dim) + p_dim inputs."""

    def __init__(self, module, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, rolling_window=2, forward_rolling=True, **device_kwargs):
        if not isinstance(rolling_window, int) or rolling_window < 1:
            raise ValueError("Your rolling windows need to be an integer gretter than 1.")
        weights, biases, activations = extract_linear_stack(module)
        if weights[-1].shape[1] != x_dim:
            raise ValueError("Your torch module do not provide a suitable output dim ! \n It must get the same dim as the state dim.")
        super().__init__(weights, biases, x_dim, u_dim, p_dim, tvp_dim, rolling_window=rolling_window,
                         forward_rolling=forward_rolling, activations=activations, **device_kwargs)
