"""KerasTFModel drop-in (reference: model/tensorflow.py:8-109) without importing TensorFlow.

The reference wraps a live Keras model -- any feed-forward one (model/tensorflow.py:8-29) -- and differentiates it with
tf.GradientTape on the CPU.  Here the Keras object is only *read*: its Dense kernels / biases / activation names are
copied once and the network is evaluated by the HIP kernels.  Dense stacks with any of the activations linear, tanh,
relu, sigmoid, softplus, elu(alpha), leaky_relu(alpha), selu (per layer, the output layer included), swish / silu, gelu,
softsign, mish, exponential and relu6 (hidden layers; these six run on the layer-at-a-time matrix-core path only) are taken, stand-alone Activation / ReLU /
LeakyReLU / ELU layers fold into the Dense in front of them; anything else -- other layer types, other activations -- is
rejected loudly.  BatchNormalization / Normalization / Rescaling layers are folded into the neighbouring Dense layer."""
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


# layers that are the identity at inference time (what the reference's model.predict evaluates, model/tensorflow.py:49-51)
_IDENTITY_LAYERS = {"InputLayer", "Dropout", "AlphaDropout", "GaussianDropout", "GaussianNoise", "ActivityRegularization",
                    "Flatten"}


def _standalone_activation(layer, i):
    """Keras activation name of a parameter-less activation layer (`Activation('tanh')`, `ReLU()`, `ELU()`, `Softplus`-like
    wrappers), or a loud refusal: such a layer must never be dropped silently -- the network would be evaluated as a
    different function."""
    kind = type(layer).__name__
    if kind == "Activation":
        return _activation_name(layer)
    if kind == "ReLU":
        slope = float(getattr(layer, "negative_slope", 0.0) or 0.0)
        mx = getattr(layer, "max_value", None)
        if float(getattr(layer, "threshold", 0.0) or 0.0) != 0.0 or (mx is not None and (float(mx) != 6.0 or slope != 0.0)):
            raise NotImplementedError(f"layer {i}: ReLU with a threshold, or a max_value other than 6, is unsupported on the device path")
        if mx is not None:
            return "relu6"
        return "relu" if slope == 0.0 else f"leaky_relu:{slope!r}"
    if kind == "ELU":
        alpha = float(getattr(layer, "alpha", 1.0))
        return "elu" if alpha == 1.0 else f"elu:{alpha!r}"
    if kind == "LeakyReLU":
        alpha = getattr(layer, "negative_slope", None)
        if alpha is None:
            alpha = getattr(layer, "alpha", 0.3)                     # (Keras 2 spelling; both default to 0.3)
        return f"leaky_relu:{float(alpha)!r}"
    raise NotImplementedError(f"layer {i}: parameter-less layer '{kind}' is unsupported on the device path (it would be "
                              "dropped from the network); supported: Dense, Activation, ReLU, LeakyReLU, ELU and the "
                              "inference-identity layers " + ", ".join(sorted(_IDENTITY_LAYERS)))


def _affine_layer(layer, i):
    """(scale, shift) of a layer that is an elementwise affine map at inference -- BatchNormalization (moving statistics),
    the Normalization / Rescaling preprocessing layers -- or None.  Such a layer folds into a neighbouring Dense layer exactly."""
    kind = type(layer).__name__
    if kind == "BatchNormalization":
        params = [np.asarray(p, dtype=np.float64) for p in layer.get_weights()]
        scale_on, center_on = bool(getattr(layer, "scale", True)), bool(getattr(layer, "center", True))
        if len(params) != 2 + int(scale_on) + int(center_on) or any(p.ndim != 1 for p in params):
            raise NotImplementedError(f"layer {i}: BatchNormalization over anything but the feature axis is unsupported on the device path")
        mean, var = params[-2], params[-1]
        gamma = params[0] if scale_on else np.ones_like(mean)
        beta = params[int(scale_on)] if center_on else np.zeros_like(mean)
        sc = gamma / np.sqrt(var + float(getattr(layer, "epsilon", 1e-3)))
        return sc, beta - mean * sc
    if kind == "Normalization":
        if getattr(layer, "invert", False):
            raise NotImplementedError(f"layer {i}: Normalization(invert=True) is unsupported on the device path")
        mean = np.asarray(layer.mean, dtype=np.float64).reshape(-1)
        var = np.asarray(layer.variance, dtype=np.float64).reshape(-1)
        sc = 1.0 / np.maximum(np.sqrt(var), 1e-7)           # (Keras: max(sqrt(var), backend.epsilon()))
        return sc, -mean * sc
    if kind == "Rescaling":
        sc = np.asarray(layer.scale, dtype=np.float64).reshape(-1)
        return sc, np.broadcast_to(np.asarray(getattr(layer, "offset", 0.0), dtype=np.float64).reshape(-1), sc.shape) if sc.size > 1 \
            else np.asarray(getattr(layer, "offset", 0.0), dtype=np.float64).reshape(-1)
    return None


def extract_dense_stack(keras_model):
    """-> (weights, biases, activations) from a duck-typed Keras Sequential/Functional model of Dense layers;
    activations holds one Keras activation name per dense layer.  EVERY layer is walked in order: a stand-alone activation
    layer (`Dense(64)` + `Activation('tanh')`, `ReLU()`, `ELU()`) is folded into the linear Dense in front of it, layers that
    are the identity at inference (InputLayer, Dropout, ...) are skipped, an elementwise affine layer (BatchNormalization with
    its moving statistics, Normalization, Rescaling) is folded exactly into the Dense behind it -- or, right behind a Dense
    that has not been given an activation yet, into that one -- and anything else is refused: the reference evaluates the
    Keras model as built (model/tensorflow.py:49-51), so no layer may be ignored."""
    def flat(layers):            # (a Sequential used as a layer of another model: its layers, in order)
        for l in layers:
            sub = getattr(l, "layers", None)
            if sub is not None and type(l).__name__ in ("Sequential", "Functional", "Model"):
                yield from flat(sub)
            else:
                yield l
    all_layers = list(flat(getattr(keras_model, "layers", [])))
    weights, biases, activations = [], [], []
    pre = None                     # (scale, shift) waiting for the next Dense: x -> scale * x + shift in front of it
    for i, layer in enumerate(all_layers):
        aff = _affine_layer(layer, i)
        if aff is not None:
            sc, sh = (np.asarray(v, dtype=np.float64) for v in aff)
            if weights and activations[-1] == "linear" and pre is None:
                # behind a linear Dense: y -> sc * (x W + b) + sh
                weights[-1] = weights[-1] * sc[None, :]
                biases[-1] = biases[-1] * sc + sh
            else:
                pre = (sc, sh) if pre is None else (pre[0] * sc, pre[1] * sc + sh)
            continue
        params = layer.get_weights()
        if len(params) == 0:
            if type(layer).__name__ in _IDENTITY_LAYERS:
                if type(layer).__name__ == "Flatten":
                    # the identity only on an input that is one vector per sample already
                    shp = getattr(layer, "input_shape", None)
                    if shp is None and getattr(layer, "input", None) is not None:
                        shp = getattr(layer.input, "shape", None)
                    if shp is not None and len(tuple(shp)) != 2:
                        raise NotImplementedError(f"layer {i}: Flatten of a {len(tuple(shp)) - 1}-D sample is not the identity; "
                                                  "unsupported on the device path")
                continue
            name = _standalone_activation(layer, i)
            if name == "linear":
                continue
            if not activations or pre is not None:
                raise NotImplementedError(f"layer {i}: an activation layer that does not follow a Dense layer directly is unsupported")
            if activations[-1] != "linear":
                raise NotImplementedError(f"layer {i}: activation '{name}' on top of a Dense layer that already applies "
                                          f"'{activations[-1]}' is unsupported on the device path")
            try:
                _lib.split_activation(name)
            except NotImplementedError as e:
                raise NotImplementedError(f"layer {i}: {e}")
            activations[-1] = name
            continue
        if len(params) == 1 and np.ndim(params[0]) == 2:
            # Dense(use_bias=False) -- and nothing else: an Embedding or a custom layer also holds ONE matrix, and no layer
            # may be evaluated as a different function
            if type(layer).__name__ != "Dense" or getattr(layer, "use_bias", True) is not False:
                raise NotImplementedError(f"layer {i}: '{type(layer).__name__}' holds a single matrix but is not Dense(use_bias=False); "
                                          "only Dense layers (kernel[, bias]) are supported on the device path")
            params = [params[0], np.zeros(np.shape(params[0])[1])]
        if len(params) != 2 or np.ndim(params[0]) != 2 or np.ndim(params[1]) != 1:
            raise NotImplementedError("Only Dense layers (kernel, bias) are supported on the device path")
        name = _activation_name(layer)
        if name == "elu" and float(getattr(layer.activation, "alpha", 1.0)) != 1.0:
            name = f"elu:{float(layer.activation.alpha)!r}"
        try:
            _lib.split_activation(name)
        except NotImplementedError as e:
            raise NotImplementedError(f"layer {i}: {e}")
        Wl, bl = np.asarray(params[0], dtype=np.float64), np.asarray(params[1], dtype=np.float64)
        if pre is not None:          # (scale * x + shift) W + b = x (diag(scale) W) + (shift W + b)
            sc, sh = (np.broadcast_to(v, (Wl.shape[0],)) for v in pre)
            bl = bl + sh @ Wl
            Wl = Wl * sc[:, None]
            pre = None
        weights.append(Wl)
        biases.append(bl)
        activations.append(name)
    if not weights:
        raise ValueError("The provided model has no parameterised layers")
    if pre is not None:
        raise NotImplementedError("an affine layer (BatchNormalization / Normalization / Rescaling) behind the last non-linear layer "
                                  "has no Dense layer to fold into")
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
