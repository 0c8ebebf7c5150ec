# Links between user models and the solving system (reference: pyNeuralEMPC/model/__init__.py).
from . import base
from . import mlp
from . import rolling
from . import jax
from . import tensorflow
from . import torch_model
from . import torch_mlp
from .base import Model
from .mlp import MLPModel
from .rolling import MLPModelRollingInput
from .tensorflow import KerasTFModel, KerasTFModelRollingInput
from .torch_model import TorchModel, TorchModelRollingWindow
from .torch_mlp import TorchMLPModel, TorchMLPModelRollingInput
