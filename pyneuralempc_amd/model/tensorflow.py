"""KerasTFModel drop-in (reference: model/tensorflow.py:8-109) without importing TensorFlow.

The reference wraps a live Keras model -- any feed-forward one (model/tensorflow.py:8-29) -- and differentiates it with
tf.GradientTape on the CPU.  Here the Keras object is only *read*: its Dense kernels / biases / activation names are
copied once and the network is evaluated by the HIP kernels.  Dense stacks with any of the activations linear, tanh,
relu, sigmoid, softplus, elu (per layer, the output layer included) are taken; anything else -- other layer types,
other activations -- is rejected loudly."""
import numpy as np

from .mlp import MLPModel
from .rolling import MLPModelRollingInput


from .. import _lib


def _activation_name(layer):
    act = getattr(layer, "activation", None)
    if act is None:
        return "linear"
    if isinstance(act, str):
        return act
    return getattr(act, "__name__", None) or getattr(act, "name", None) or str(act)


def extract_dense_stack(keras_model):
    """-> (weights, biases, activations) from a duck-typed Keras Sequential/Functional model of Dense layers;
    activations holds one Keras activation name per dense layer (what Dense(..., activation=...) was given)."""
    layers = [l for l in getattr(keras_model, "layers", []) if len(l.get_weights()) > 0]
    if not layers:
        raise ValueError("The provided model has no parameterised layers")
    weights, biases, activations = [], [], []
    for i, layer in enumerate(layers):
        params = layer.get_weights()
        if len(params) != 2 or np.ndim(params[0]) != 2 or np.ndim(params[1]) != 1:
            raise NotImplementedError("Only Dense layers (kernel, bias) are supported on the device path")
        name = _activation_name(layer)
        if name not in _lib.ACTIVATION_IDS:
            raise NotImplementedError(f"layer {i}: activation '{name}' unsupported on the device path (supported: "
                                      f"{', '.join(_lib.ACTIVATION_IDS)})")
        if name == "elu" and float(getattr(layer.activation, "alpha", 1.0)) != 1.0:
            raise NotImplementedError(f"layer {i}: elu with alpha != 1 is unsupported on the device path")
        weights.append(np.asarray(params[0], dtype=np.float64))
        biases.append(np.asarray(params[1], dtype=np.float64))
        activations.append(name)
    return weights, biases, activations


class KerasTFModel(MLPModel):
    def __init__(self, model, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, standardScaler=None, **device_kwargs):
        if standardScaler is not None:
            raise NotImplementedError("This feature isn't supported yet !")
        if len(model.input_shape) != 2:
            raise NotImplementedError("Recurrent neural network are not supported atm.")
        if model.output_shape[-1] != x_dim:
            raise ValueError("Your Keras model do not provide a suitable output dim ! \n It must get the same dim "
                             "as the state dim.")
        if model.input_shape[-1] != sum((x_dim, u_dim, p_dim, tvp_dim)):
            raise ValueError("Your Keras model do not provide a suitable input dim ! \n It must get the same dim as "
                             "the sum of all input vars (x, u, p, tvp).")
        weights, biases, activations = extract_dense_stack(model)
        super().__init__(weights, biases, x_dim, u_dim, p_dim, tvp_dim, activations=activations, **device_kwargs)


class KerasTFModelRollingInput(MLPModelRollingInput):
    """Drop-in for model/tensorflow.py:132-340: the Keras model reads rolling_window * (x_dim + u_dim + tvp_dim)
    + p_dim inputs; set_prev_data supplies the history before each solve."""

    def __init__(self, model, x_dim: int, u_dim: int, p_dim=0, tvp_dim=0, rolling_window=2, forward_rolling=True,
                 standardScaler=None, **device_kwargs):
        if standardScaler is not None:
            raise NotImplementedError("This feature isn't supported yet !")
        if model.output_shape[-1] != x_dim:
            raise ValueError("Your Keras model do not provide a suitable output dim ! \n It must get the same dim "
                             "as the state dim.")
        if not isinstance(rolling_window, int) or rolling_window < 1:
            raise ValueError("Your rolling windows need to be an integer gretter than 1.")
        weights, biases, activations = extract_dense_stack(model)
        super().__init__(weights, biases, x_dim, u_dim, p_dim, tvp_dim, rolling_window=rolling_window,
                         forward_rolling=forward_rolling, activations=activations, **device_kwargs)
