"""CPU: host-side logic of the plug-in surface (no device work)."""
import os
import pickle

import numpy as np
import pytest

from helpers import GOLDEN


@pytest.fixture(scope="module")
def misc():
    return dict(np.load(os.path.join(GOLDEN, "misc.npz")))


def test_start_vectors_match_reference(misc):
    from pyneuralempc_amd.optimizer.base import cold_start, warm_start_shift
    H = int(misc["nmpc_H"])
    np.testing.assert_array_equal(cold_start(misc["nmpc_x0"], H, 1), misc["cold_init"])
    np.testing.assert_array_equal(warm_start_shift(misc["nmpc_prev"], H, 2, 1), misc["warm_from_first"])


def test_domain_constraint_bounds_match_reference(misc):
    from pyneuralempc_amd.constraints import DomainConstraint
    dc = DomainConstraint(states_constraint=[[-np.inf, 1.0], [-2.0, np.inf]], control_constraint=[[-1.0, 0.2]])
    np.testing.assert_array_equal(np.array(dc.get_lower_bounds(5)), misc["dom_lb"])
    np.testing.assert_array_equal(np.array(dc.get_upper_bounds(5)), misc["dom_ub"])
    assert dc.get_dim(5) == (2, 1)
    with pytest.raises(ValueError):
        DomainConstraint([], [[0, 1]])
    with pytest.raises(ValueError):
        DomainConstraint([[0, 1]], [])
    with pytest.raises(ValueError):
        DomainConstraint([[0, 1, 2]], [[0, 1]])


def test_constraint_types():
    from pyneuralempc_amd.constraints import (BoxStateConstraint, Constraint, EqualityConstraint,
                                              InequalityConstraint)

    class E(EqualityConstraint):
        def get_dim(self, H): return 3 * H

    class I(InequalityConstraint):
        def get_dim(self, H): return H

    assert E().get_type(4) == Constraint.EQ_TYPE and len(E().get_lower_bounds(4)) == 12
    assert I().get_type(4) == Constraint.INEQ_TYPE and np.all(np.isinf(I().get_upper_bounds(4)))
    box = BoxStateConstraint(-2.0, 2.0, x_dim=2)
    assert box.get_type(5) == Constraint.INTER_TYPE and box.get_dim(5) == 10
    x, u = np.arange(10.0).reshape(5, 2), np.zeros((5, 1))
    np.testing.assert_array_equal(box.forward(x, u), x.ravel())
    J = box.jacobian(x, u)
    assert J.shape == (10, 15) and np.array_equal(J[:, :10], np.eye(10)) and not J[:, 10:].any()
    assert box.hessian(x, u).shape == (10, 15, 15)
    with pytest.raises(ValueError):
        BoxStateConstraint(-1.0, 1.0)


def test_factory_and_controller_contract():
    from pyneuralempc_amd.constraints import DomainConstraint
    from pyneuralempc_amd.controller import MPC, NMPC
    from pyneuralempc_amd.optimizer import Optimizer, Slsqp
    from pyneuralempc_amd.optimizer.base import ProblemFactory
    assert MPC is NMPC and Optimizer.SUCCESS == 0 and Optimizer.FAIL == 1
    f = ProblemFactory()
    with pytest.raises(RuntimeError, match="x0 is missing"):
        f.getProblemInterface()
    f.set_x0(np.zeros(2))
    with pytest.raises(RuntimeError, match="objective is missing"):
        f.getProblemInterface()

    class FakeModel:
        x_dim, u_dim, p_dim, tvp_dim = 2, 1, 0, 0

    class FakeIntegrator:
        model, H = FakeModel(), 3

    dom = DomainConstraint([[-1, 1]] * 2, [[-1, 1]])
    lst = [dom]
    mpc = NMPC(FakeIntegrator(), object(), lst, 3, 1.0, optimizer=Slsqp())
    assert lst == [] and mpc.domain_constraint is dom       # reference behaviour: list is mutated
    with pytest.raises(AssertionError, match="x0 must be a vector"):
        mpc.next(np.zeros((1, 2)))
    with pytest.raises(AssertionError, match="x0 dim"):
        mpc.next(np.zeros(3))
    with pytest.raises(AssertionError, match="both init values"):
        mpc.next(np.zeros(2), init_x=np.zeros((3, 2)))
    with pytest.raises(IndexError):
        NMPC(FakeIntegrator(), object(), [], 3, 1.0, optimizer=Slsqp())


def test_keras_adapter_folds_affine_layers_into_the_dense_stack():
    """BatchNormalization (moving statistics), Normalization and Rescaling are elementwise affine maps at inference: the
    adapter folds them exactly into the neighbouring Dense layer -- the extracted stack evaluates to what the layers
    evaluate to one by one (what the reference's model.predict would return, model/tensorflow.py:49-51) -- and refuses the
    placements that have no Dense to fold into."""
    from pyneuralempc_amd.model.tensorflow import extract_dense_stack
    rng = np.random.default_rng(0)

    class Layer:
        def __init__(self, W, b, act): self.W, self.b, self.activation = W, b, act
        def get_weights(self): return [self.W, self.b]

    class BatchNormalization:
        scale, center, epsilon = True, True, 1e-3
        def __init__(self, n, scale=True, center=True):
            self.scale, self.center = scale, center
            self.gamma, self.beta = rng.uniform(0.5, 1.5, n), rng.normal(size=n)
            self.mean, self.var = rng.normal(size=n), rng.uniform(0.5, 2.0, n)
        def get_weights(self):
            return ([self.gamma] if self.scale else []) + ([self.beta] if self.center else []) + [self.mean, self.var]
        def __call__(self, x):
            y = (x - self.mean) / np.sqrt(self.var + self.epsilon)
            return (self.gamma if self.scale else 1.0) * y + (self.beta if self.center else 0.0)

    class Normalization:
        invert = False
        def __init__(self, n): self.mean, self.variance = rng.normal(size=(1, n)), rng.uniform(0.5, 2.0, (1, n))
        def get_weights(self): return [self.mean, self.variance, np.array(0)]
        def __call__(self, x): return (x - self.mean.ravel()) / np.sqrt(self.variance.ravel())

    class Rescaling:
        scale, offset = 0.5, -1.0
        def get_weights(self): return []
        def __call__(self, x): return x * self.scale + self.offset

    class Activation:
        def __init__(self, act): self.activation = act
        def get_weights(self): return []

    class Fake:
        def __init__(self, layers): self.layers = layers

    L1, L2, L3 = (Layer(rng.normal(size=(3, 8)), rng.normal(size=8), "linear"), Layer(rng.normal(size=(8, 8)), rng.normal(size=8), "tanh"),
                  Layer(rng.normal(size=(8, 2)), rng.normal(size=2), "linear"))
    N0, R0, B1, B2, B3 = Normalization(3), Rescaling(), BatchNormalization(8), BatchNormalization(8, scale=False), BatchNormalization(2, center=False)
    W, b, acts = extract_dense_stack(Fake([N0, R0, L1, B1, Activation("tanh"), L2, B2, L3, B3]))
    assert acts == ["tanh", "tanh", "linear"]
    x = rng.normal(size=(6, 3))
    y = B3(B2(np.tanh(np.tanh(B1(R0(N0(x)) @ L1.W + L1.b)) @ L2.W + L2.b)) @ L3.W + L3.b)
    z = x
    for Wl, bl, a in zip(W, b, acts):
        z = z @ Wl + bl
        z = np.tanh(z) if a == "tanh" else z
    np.testing.assert_allclose(z, y, rtol=1e-13, atol=1e-13)
    # a Sequential nested as a layer contributes its layers in order
    class Sequential(Fake):
        pass
    W2, b2, acts2 = extract_dense_stack(Fake([N0, R0, Sequential([L1, B1, Activation("tanh")]), Sequential([L2, B2, Sequential([L3])]), B3]))
    assert acts2 == acts and all(np.array_equal(p, q) for p, q in zip(W + b, W2 + b2))
    with pytest.raises(NotImplementedError, match="no Dense layer to fold into"):
        extract_dense_stack(Fake([L1, Activation("tanh"), BatchNormalization(8)]))
    with pytest.raises(NotImplementedError, match="does not follow a Dense layer directly"):
        extract_dense_stack(Fake([L2, BatchNormalization(8), Activation("relu"), L3]))


def test_torch_module_adapter_extracts_the_dense_stack():
    """TorchMLPModel's reader: nn.Linear weights (transposed to (in, out)), activation modules folded into the Linear in front,
    BatchNorm1d (eval) folded into its neighbour, nested Sequential flattened -- the extracted stack evaluates to the module's
    own forward, with every activation module of the kernels' family; what the kernels cannot express is refused."""
    import torch
    from pyneuralempc_amd.model.torch_mlp import TorchMLPModel, extract_linear_stack
    from oracle import nempc_oracle as orc
    nn = torch.nn
    torch.manual_seed(0)
    bn = nn.BatchNorm1d(16)
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    net = nn.Sequential(nn.Linear(3, 16), bn, nn.SiLU(), nn.Sequential(nn.Linear(16, 12, bias=False), nn.ELU(0.7), nn.Dropout(0.1)),
                        nn.Linear(12, 8), nn.GELU(), nn.Linear(8, 2)).double().eval()
    W, b, acts = extract_linear_stack(net)
    assert acts == ["swish", "elu:0.7", "gelu", "linear"] and [w.shape for w in W] == [(3, 16), (16, 12), (12, 8), (8, 2)]
    x = np.random.default_rng(0).normal(size=(7, 3))
    np.testing.assert_allclose(orc.MLP(W, b, acts).forward(x), net(torch.tensor(x)).detach().numpy(), rtol=1e-13, atol=1e-14)
    for mod, name in ((nn.Tanh(), "tanh"), (nn.ReLU(), "relu"), (nn.Sigmoid(), "sigmoid"), (nn.Softplus(), "softplus"), (nn.SELU(), "selu"),
                      (nn.LeakyReLU(0.05), "leaky_relu:0.05"), (nn.Softsign(), "softsign"), (nn.Mish(), "mish"), (nn.ReLU6(), "relu6"),
                      (nn.ELU(), "elu")):
        m = nn.Sequential(nn.Linear(3, 5), mod, nn.Linear(5, 2)).double().eval()
        Wm, bm, am = extract_linear_stack(m)
        assert am == [name, "linear"]
        np.testing.assert_allclose(orc.MLP(Wm, bm, am).forward(x), m(torch.tensor(x)).detach().numpy(), rtol=1e-12, atol=1e-13)
    for bad, msg in ((nn.Sequential(nn.Linear(3, 4), nn.GELU(approximate="tanh"), nn.Linear(4, 2)), "approximate"),
                     (nn.Sequential(nn.Linear(3, 4), nn.Softplus(beta=2.0), nn.Linear(4, 2)), "beta"),
                     (nn.Sequential(nn.Linear(3, 4), nn.Hardtanh(), nn.Linear(4, 2)), "Hardtanh"),
                     (nn.Sequential(nn.Linear(3, 4), nn.BatchNorm1d(4), nn.Linear(4, 2)).train(), "eval"),
                     (nn.Sequential(nn.Conv1d(1, 1, 3)), "Conv1d")):
        with pytest.raises(NotImplementedError, match=msg):
            extract_linear_stack(bad)
    with pytest.raises(ValueError, match="output dim"):
        TorchMLPModel(net, x_dim=3, u_dim=0)
    with pytest.raises(ValueError, match="input dim"):
        TorchMLPModel(net, x_dim=2, u_dim=2)
    m = TorchMLPModel(net, x_dim=2, u_dim=1)
    assert m.activations == acts and (m.x_dim, m.u_dim) == (2, 1)
    # the rolling-window reader: rolling_window * (x + u) inputs (reference: KerasTFModelRollingInput, model/tensorflow.py:132-147)
    from pyneuralempc_amd.model.torch_mlp import TorchMLPModelRollingInput
    from pyneuralempc_amd.model.rolling import MLPModelRollingInput
    roll = nn.Sequential(nn.Linear(9, 12), nn.Tanh(), nn.Linear(12, 2)).double().eval()
    mr = TorchMLPModelRollingInput(roll, x_dim=2, u_dim=1, rolling_window=3)
    Wr, br, ar = extract_linear_stack(roll)
    ref = MLPModelRollingInput(Wr, br, 2, 1, rolling_window=3, activations=ar)
    assert mr.rolling_window == ref.rolling_window == 3 and mr.activations == ref.activations
    assert all(np.array_equal(p, q) for p, q in zip(mr.weights, ref.weights))
    with pytest.raises(ValueError, match="output dim"):
        TorchMLPModelRollingInput(roll, x_dim=3, u_dim=1, rolling_window=3)


def test_keras_adapter_validation_without_device():
    from pyneuralempc_amd.model.tensorflow import KerasTFModel, extract_dense_stack

    def tanh(x): return x

    def linear(x): return x

    def relu(x): return x

    class Layer:
        def __init__(self, w, b, act): self.w, self.b, self.activation = w, b, act
        def get_weights(self): return [self.w, self.b]

    class Fake:
        def __init__(self, layers, nin, nout):
            self.layers, self.input_shape, self.output_shape = layers, (None, nin), (None, nout)

    good = Fake([Layer(np.ones((3, 8)), np.zeros(8), tanh), Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2)
    W, b, acts = extract_dense_stack(good)
    assert [w.shape for w in W] == [(3, 8), (8, 2)] and acts == ["tanh", "linear"]
    m = KerasTFModel(good, x_dim=2, u_dim=1)
    assert (m.x_dim, m.u_dim, m.p_dim, m.tvp_dim) == (2, 1, 0, 0) and m.activations == ["tanh", "linear"]
    m2 = pickle.loads(pickle.dumps(m))
    assert m2._row_engine is None and np.array_equal(m2.weights[0], m.weights[0])
    with pytest.raises(ValueError, match="output dim"):
        KerasTFModel(good, x_dim=3, u_dim=1)
    with pytest.raises(ValueError, match="input dim"):
        KerasTFModel(good, x_dim=2, u_dim=2)
    with pytest.raises(NotImplementedError):
        KerasTFModel(good, x_dim=2, u_dim=1, standardScaler=object())
    # any Keras activation of the device family is taken, per layer, the output layer included (the reference wraps
    # ANY feed-forward Keras model, model/tensorflow.py:8-29); names also arrive as strings or objects with .name
    def softplus(x): return x

    class Named:
        name = "elu"
    mixed = Fake([Layer(np.ones((3, 8)), np.zeros(8), relu), Layer(np.ones((8, 8)), np.zeros(8), "sigmoid"),
                  Layer(np.ones((8, 8)), np.zeros(8), Named()), Layer(np.ones((8, 2)), np.zeros(2), softplus)], 3, 2)
    mm = KerasTFModel(mixed, x_dim=2, u_dim=1)
    assert mm.activations == ["relu", "sigmoid", "elu", "softplus"]
    assert pickle.loads(pickle.dumps(mm)).activations == mm.activations
    none_act = Fake([Layer(np.ones((3, 2)), np.zeros(2), None)], 3, 2)
    assert KerasTFModel(none_act, x_dim=2, u_dim=1).activations == ["linear"]

    # parameter-less layers are never dropped silently: Dense(8) + Activation('tanh') / ReLU() / ELU() fold into the linear
    # Dense in front of them, inference-identity layers are skipped, anything else is refused
    class Activation:
        def __init__(self, act): self.activation = act
        def get_weights(self): return []

    class ReLU:
        max_value, negative_slope, threshold = None, 0.0, 0.0
        def get_weights(self): return []

    class ELU:
        alpha = 1.0
        def get_weights(self): return []

    class Dropout:
        def get_weights(self): return []

    class InputLayer(Dropout):
        pass

    class LeakyReLU(Dropout):
        negative_slope = 0.3

    class Softmax(Dropout):
        pass

    split = Fake([InputLayer(), Layer(np.ones((3, 8)), np.zeros(8), linear), Activation(tanh), Dropout(),
                  Layer(np.ones((8, 8)), np.zeros(8), None), ReLU(), Layer(np.ones((8, 8)), np.zeros(8), "linear"), ELU(),
                  Layer(np.ones((8, 2)), np.zeros(2), linear), Activation("linear")], 3, 2)
    W, b, acts = extract_dense_stack(split)
    assert len(W) == 4 and acts == ["tanh", "relu", "elu", "linear"]
    assert KerasTFModel(split, x_dim=2, u_dim=1).activations == acts
    with pytest.raises(NotImplementedError, match="Softmax"):
        extract_dense_stack(Fake([Layer(np.ones((3, 8)), np.zeros(8), linear), Softmax(),
                                  Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))
    # the monotone parameterised family: LeakyReLU / ELU layers and ReLU(negative_slope) carry their alpha in the name
    half = ELU()
    half.alpha = 0.5
    W, b, acts = extract_dense_stack(Fake([Layer(np.ones((3, 8)), np.zeros(8), linear), LeakyReLU(),
                                           Layer(np.ones((8, 8)), np.zeros(8), linear), half,
                                           Layer(np.ones((8, 8)), np.zeros(8), "selu"),
                                           Layer(np.ones((8, 2)), np.zeros(2), "leaky_relu")], 3, 2))
    assert acts == ["leaky_relu:0.3", "elu:0.5", "selu", "leaky_relu"]
    # swish (= silu) and gelu are taken (layered path); an activation outside the family is refused by name
    assert extract_dense_stack(Fake([Layer(np.ones((3, 8)), np.zeros(8), "silu"), Layer(np.ones((8, 8)), np.zeros(8), "gelu"),
                                     Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))[2] == ["silu", "gelu", "linear"]
    with pytest.raises(NotImplementedError, match="hard_sigmoid"):
        extract_dense_stack(Fake([Layer(np.ones((3, 8)), np.zeros(8), "hard_sigmoid"), Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))
    with pytest.raises(NotImplementedError, match="already applies"):
        extract_dense_stack(Fake([Layer(np.ones((3, 8)), np.zeros(8), tanh), Activation(relu),
                                  Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))
    leaky = ReLU()
    leaky.negative_slope = 0.1
    assert extract_dense_stack(Fake([Layer(np.ones((3, 8)), np.zeros(8), linear), leaky,
                                     Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))[2] == ["leaky_relu:0.1", "linear"]
    capped = ReLU()
    capped.max_value = 6.0
    assert extract_dense_stack(Fake([Layer(np.ones((3, 8)), np.zeros(8), linear), capped,
                                     Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))[2] == ["relu6", "linear"]
    capped.max_value = 4.0
    with pytest.raises(NotImplementedError, match="max_value"):
        extract_dense_stack(Fake([Layer(np.ones((3, 8)), np.zeros(8), linear), capped,
                                  Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))

    # Dense(use_bias=False) -- by its class name and its use_bias flag, nothing else that happens to hold one matrix
    class Dense:
        activation, use_bias = "tanh", False
        def get_weights(self): return [np.ones((3, 8))]
    W, b, acts = extract_dense_stack(Fake([Dense(), Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))
    assert np.array_equal(b[0], np.zeros(8)) and acts == ["tanh", "linear"]

    class Embedding(Dense):
        pass
    with pytest.raises(NotImplementedError, match="Embedding"):
        extract_dense_stack(Fake([Embedding(), Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))
    biased = Dense()
    biased.use_bias = True
    with pytest.raises(NotImplementedError, match="single matrix"):
        extract_dense_stack(Fake([biased, Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2))

    # Flatten is the identity on one vector per sample only
    class Flatten(Dropout):
        input_shape = (None, 3)
    assert extract_dense_stack(Fake([Flatten(), Layer(np.ones((3, 2)), np.zeros(2), linear)], 3, 2))[2] == ["linear"]
    img = Flatten()
    img.input_shape = (None, 3, 4)
    with pytest.raises(NotImplementedError, match="Flatten"):
        extract_dense_stack(Fake([img, Layer(np.ones((12, 2)), np.zeros(2), linear)], 12, 2))

    def hard_sigmoid(x): return x
    bad = Fake([Layer(np.ones((3, 8)), np.zeros(8), hard_sigmoid), Layer(np.ones((8, 2)), np.zeros(2), linear)], 3, 2)
    with pytest.raises(NotImplementedError, match="activation 'hard_sigmoid'"):
        KerasTFModel(bad, x_dim=2, u_dim=1)

    class Elu2:
        name, alpha = "elu", 0.5
    assert KerasTFModel(Fake([Layer(np.ones((3, 2)), np.zeros(2), Elu2())], 3, 2), x_dim=2, u_dim=1).activations == ["elu:0.5"]
    rnn = Fake(good.layers, 3, 2)
    rnn.input_shape = (None, 5, 3)
    with pytest.raises(NotImplementedError, match="Recurrent"):
        KerasTFModel(rnn, x_dim=2, u_dim=1)


def test_integrator_rejects_foreign_models():
    from pyneuralempc_amd.integrator import DiscretIntegrator
    from pyneuralempc_amd.model.base import Model
    with pytest.raises(ValueError):
        DiscretIntegrator(object(), 5)
    # any Model plug-in is taken (host algebra over its forward / jacobian / hessian, as in the reference); the abstract
    # base itself has no forward, and says so when the integrator calls it
    integ = DiscretIntegrator(Model(2, 1), 5)
    assert not integ.on_device
    with pytest.raises(NotImplementedError):
        integ.forward(np.zeros((5, 2)), np.zeros((5, 1)), np.zeros(2))


def test_quadratic_objective_host_side():
    from pyneuralempc_amd.objective import ManualObjectifFunc, QuadraticObjective
    q = QuadraticObjective(Q=[[1.0, 0.5], [0.0, 2.0]], R=[[0.3]])

    class M:
        x_dim, u_dim = 2, 1
    Hm = q.hessian(np.zeros((3, 2)), np.zeros((3, 1)))
    assert Hm.shape == (9, 9) and np.allclose(Hm[:2, :2], [[2.0, 0.5], [0.5, 4.0]]) and np.isclose(Hm[6, 6], 0.6)
    S = q.hessianstructure(3, M())
    assert np.array_equal(S != 0, Hm != 0)
    # terminal weight: the last state block carries QT + QT^T, and a parameter edit changes the fingerprint
    fp = q.fingerprint(3, 2, 1)
    q.params["QT"] = [[5.0, 0.0], [1.0, 7.0]]
    Ht = q.hessian(np.zeros((3, 2)), np.zeros((3, 1)))
    assert np.allclose(Ht[4:6, 4:6], [[10.0, 1.0], [1.0, 14.0]]) and np.allclose(Ht[:4, :4], Hm[:4, :4])
    assert q.fingerprint(3, 2, 1) != fp
    man = ManualObjectifFunc(lambda s, u, p, t: 1.0, lambda s, u, p, t: np.zeros(3), lambda s, u, p, t: np.eye(3))
    assert man.forward(None, None) == 1.0 and man.hessian(None, None).shape == (3, 3)


def test_torch_objective_plugin_matches_closed_form_family():
    """TorchObjectifFunc (the JAXObjectifFunc counterpart, objective/jax.py:16-90) on the reference's two cost
    functions and on the general quadratic family, against the oracle's closed forms."""
    import torch
    from oracle import nempc_oracle as orc
    from pyneuralempc_amd.objective import TorchObjectifFunc
    H, nx, nu = 6, 2, 1
    rng = np.random.default_rng(0)
    Q = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx))
    R = np.array([[0.3]])
    xref, uref, cu = rng.normal(size=(H, nx)), rng.normal(size=(H, nu)), rng.normal(size=(H, nu))
    prob = orc.Problem(orc.MLP.random(nx + nu, [4], nx), H, nx, nu, Q=Q, R=R, xref=xref, uref=uref, cu=cu)
    Qt, Rt, xr, ur, ct = (torch.tensor(a) for a in (Q, R, xref, uref, cu))

    def cost(states, u, p=None, tvp=None):
        dx, du = states - xr, u - ur
        return torch.einsum("ti,ij,tj->", dx, Qt, dx) + torch.einsum("ti,ij,tj->", du, Rt, du) + torch.sum(ct * u)

    obj = TorchObjectifFunc(cost, device="cpu")
    states, u = rng.normal(size=(H, nx)), rng.normal(size=(H, nu))
    z = np.concatenate([states.ravel(), u.ravel()])
    np.testing.assert_allclose(obj.forward(states, u), prob.objective(z), rtol=1e-13)
    np.testing.assert_allclose(obj.gradient(states, u), prob.gradient(z), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(obj.hessian(states, u), prob.objective_hessian(), rtol=1e-12, atol=1e-13)

    class M:
        x_dim, u_dim, p_dim, tvp_dim = nx, nu, 0, 0
    m = M()
    S = obj.hessianstructure(H, m)
    assert np.array_equal(S != 0, prob.objective_hessian() != 0) and obj.hessianstructure(H, m) is S
    # the reference's own two costs: sum(u * c) (run.py:89-90, gradient does not touch the states) and sum((u-2)^2)
    lin = TorchObjectifFunc(lambda s, u, p=None, tvp=None: torch.sum(u.reshape(-1) * 1.1), device="cpu")
    np.testing.assert_array_equal(lin.gradient(states, u), np.concatenate([np.zeros(H * nx), np.full(H * nu, 1.1)]))
    assert not lin.hessian(states, u).any()
    sq = TorchObjectifFunc(lambda s, u, p=None, tvp=None: torch.sum((u.reshape(-1) - 2.0) ** 2), device="cpu")
    np.testing.assert_allclose(sq.gradient(states, u)[H * nx:], 2.0 * (u.ravel() - 2.0), rtol=1e-14)


def test_reference_names_without_a_device_counterpart_explain_themselves():
    import pyneuralempc_amd as nEMPC
    with pytest.raises(NotImplementedError, match="TorchObjectifFunc"):
        nEMPC.objective.jax.JAXObjectifFunc(lambda x, u, p=None, tvp=None: 0.0)
    with pytest.raises(NotImplementedError, match="MLPModel"):
        nEMPC.model.jax.DiffDiscretJaxModel(lambda x, u, p=None, tvp=None: x, 2, 1)
    with pytest.raises(NotImplementedError, match="MLPModelRollingInput"):
        nEMPC.model.jax.DiffDiscretJaxModelRollingWindow(lambda x, u, p=None, tvp=None: x, 2, 1, rolling_window=2)
    with pytest.raises(NotImplementedError):
        nEMPC.model.base.ReOrderProxyModel(None, [])


def test_device_optimizer_host_contract():
    """optimizer.DeviceSqp on the host side: it is an Optimizer with Slsqp's start-vector plumbing and the same factory,
    and a problem without the fused device path is refused (no CPU fallback behind it)."""
    import pyneuralempc_amd as nEMPC
    from pyneuralempc_amd.optimizer.slsqp import SlsqpProblemFactory
    opt = nEMPC.optimizer.DeviceSqp(max_iteration=50, tolerance=1e-7, init_with_last_result=True, linesearch="deferred")
    assert isinstance(opt, nEMPC.optimizer.Optimizer) and isinstance(opt.get_factory(), SlsqpProblemFactory)
    assert opt.prev_result is None and opt.solver_opts == {"linesearch": "deferred"}

    class NoFusedPath:
        _fused = None

    with pytest.raises(NotImplementedError, match="fused device path"):
        opt.solve(NoFusedPath(), None)


@pytest.mark.parametrize("kind,DT", [("discret", 1.0), ("unity", 1.0), ("rk4", 0.2)])
@pytest.mark.parametrize("vector_mode", [True, False])
def test_torch_model_through_the_host_integrators_matches_the_oracle(kind, DT, vector_mode):
    """model.TorchModel (the DiffDiscretJaxModel counterpart: dynamics as a differentiable callable, model/jax.py:32-88)
    under the three integrators' host algebra and the unfused Ipopt glue -- against the oracle, with the callable being
    the same tanh network the oracle evaluates analytically.  Runs on the CPU torch device: no kernel is involved."""
    import torch
    from oracle import nempc_oracle as orc
    from pyneuralempc_amd.model import TorchModel
    from pyneuralempc_amd.integrator import DiscretIntegrator, RK4Integrator, UnityIntegrator
    from pyneuralempc_amd.objective.autodiff import TorchObjectifFunc
    from pyneuralempc_amd.optimizer.ipopt import IpoptProblem
    H, nx, nu = 5, 2, 1
    net = orc.MLP.random(nx + nu, [12, 9], nx, seed=3)
    Wt = [torch.tensor(w) for w in net.W]
    bt = [torch.tensor(b) for b in net.b]

    def f(x, u, p=None, tvp=None):           # whole trajectory (H, .) or one row (.), the same expression
        a = torch.cat([x, u], dim=-1)
        for w, b in zip(Wt[:-1], bt[:-1]):
            a = torch.tanh(a @ w + b)
        return a @ Wt[-1] + bt[-1]
    model = TorchModel(f, nx, nu, vector_mode=vector_mode, device="cpu")
    integ = {"discret": lambda: DiscretIntegrator(model, H), "unity": lambda: UnityIntegrator(model, H),
             "rk4": lambda: RK4Integrator(model, H, DT)}[kind]()
    assert not integ.on_device
    okind = {"discret": orc.DISCRET, "unity": orc.UNITY, "rk4": orc.RK4}[kind]
    Q, R = np.diag([1.0, 2.0]), np.array([[0.3]])
    prob = orc.Problem(net, H, nx, nu, okind, DT, Q=Q, R=R)
    Z, X0 = orc.synthetic_inputs(2, H, nx, nu, seed=5)
    lam = np.random.default_rng(6).normal(size=H * nx)
    Qt, Rt = torch.tensor(Q), torch.tensor(R)
    obj = TorchObjectifFunc(lambda s, c, p, tvp: torch.einsum("ti,ij,tj->", s, Qt, s) + torch.einsum("ti,ij,tj->", c, Rt, c),
                            device="cpu")
    for z, x0 in zip(Z, X0):
        x, u = prob.split(z)
        np.testing.assert_allclose(integ.forward(x, u, x0), prob.constraints(z, x0), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(integ.jacobian(x, u, x0), prob.jacobian(z, x0), rtol=1e-11, atol=1e-12)
        Hc = np.tensordot(lam, integ.hessian(x, u, x0), axes=1)
        np.testing.assert_allclose(Hc, prob.lagrangian_hessian(z, x0, lam, 0.0), rtol=1e-9, atol=1e-11)
        pb = IpoptProblem(x0, obj, [], integ)
        assert pb._fused is None                                   # the unfused glue: one plug-in call per callback
        np.testing.assert_allclose(pb.objective(z), prob.objective(z), rtol=1e-12)
        np.testing.assert_allclose(pb.gradient(z), prob.gradient(z), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(pb.constraints(z), prob.constraints(z, x0), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(pb.jacobian(z), prob.jacobian(z, x0), rtol=1e-11, atol=1e-12)
        rows, cols = pb.hessianstructure()
        np.testing.assert_allclose(pb.hessian(z, lam, 0.7), prob.lagrangian_hessian(z, x0, lam, 0.7)[rows, cols],
                                   rtol=1e-9, atol=1e-11)
    # block layouts of the model itself (model/jax.py:52-88)
    x, u = prob.split(Z[0])
    fo, Jo, So = net.forward_jac_hess(np.concatenate([x, u], axis=1))
    np.testing.assert_allclose(model.forward(x, u), fo, rtol=1e-13, atol=1e-14)
    mj, mh = model.jacobian(x, u), model.hessian(x, u)
    assert mj.shape == (H * nx, H * (nx + nu)) and mh.shape == (H, nx, H * (nx + nu), H * (nx + nu))
    for t in range(H):
        np.testing.assert_allclose(mj[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx], Jo[t][:, :nx], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(mj[t * nx:(t + 1) * nx, H * nx + t * nu:H * nx + (t + 1) * nu], Jo[t][:, nx:], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(mh[t][:, t * nx:(t + 1) * nx, H * nx + t * nu:H * nx + (t + 1) * nu], So[t][:, :nx, nx:],
                                   rtol=1e-10, atol=1e-12)
    off = mj.copy()
    for t in range(H):
        off[t * nx:(t + 1) * nx, t * nx:(t + 1) * nx] = 0
        off[t * nx:(t + 1) * nx, H * nx + t * nu:H * nx + (t + 1) * nu] = 0
    assert np.all(off == 0)


def test_torch_model_validation_and_the_jax_names():
    import torch
    from pyneuralempc_amd.model import TorchModel
    from pyneuralempc_amd.model.jax import DiffDiscretJaxModel, DiffDiscretJaxModelRollingWindow
    from pyneuralempc_amd.integrator import DiscretIntegrator
    with pytest.raises(ValueError, match="not differentiable"):
        TorchModel(lambda x, u, p=None, tvp=None: torch.tensor(x.detach().numpy()), 2, 1, device="cpu")
    with pytest.raises(NotImplementedError, match="TorchModel"):
        DiffDiscretJaxModel(lambda x, u, p, tvp: x, 2, 1)
    with pytest.raises(NotImplementedError, match="MLPModelRollingInput"):
        DiffDiscretJaxModelRollingWindow(lambda x, u, p, tvp: x, 2, 1, rolling_window=2)
    # parameters reach the callable: p constant, tvp per step
    m = TorchModel(lambda x, u, p=None, tvp=None: x * p[0] + u * tvp, 1, 1, p_dim=1, tvp_dim=1, device="cpu")
    x, u = np.array([[1.0], [2.0]]), np.array([[0.5], [0.25]])
    np.testing.assert_allclose(m.forward(x, u, p=np.array([3.0]), tvp=np.array([[2.0], [4.0]])), [[4.0], [7.0]])
    np.testing.assert_allclose(m.jacobian(x, u, p=np.array([3.0]), tvp=np.array([[2.0], [4.0]])),
                               [[3.0, 0.0, 2.0, 0.0], [0.0, 3.0, 0.0, 4.0]])
    integ = DiscretIntegrator(m, 2)
    with pytest.raises(NotImplementedError, match="dense-network model"):
        integ.engine(4)


@pytest.mark.parametrize("name", ["roll2_discret", "roll4_wide", "roll4_short"])
def test_torch_rolling_window_model_matches_the_reference_made_goldens(name):
    """model.TorchModelRollingWindow (the DiffDiscretJaxModelRollingWindow counterpart, model/jax.py:93-259: a rolling-window
    callable differentiated by autodiff) through the Discret integrator's host algebra and the unfused glue -- against the
    rolling goldens, whose Jacobians the reference's own gen_jac_proj_mat projected (tests/golden/make_golden.py).  The
    callable is the fixture's network written with torch ops.  CPU torch device: no kernel is involved."""
    import torch
    from helpers import load_case, oracle_problem
    from pyneuralempc_amd.model import TorchModelRollingWindow
    from pyneuralempc_amd.integrator import DiscretIntegrator, RK4Integrator
    d, W, b = load_case(name)
    H, nx, nu, w = int(d["H"]), int(d["nx"]), int(d["nu"]), int(d["window"])
    Wt, bt = [torch.tensor(x) for x in W], [torch.tensor(x) for x in b]

    def f(xs, us, p=None, tvp=None):                 # (H, w nx), (H, w nu) -> (H, nx): [state window | control window]
        a = torch.cat([xs, us], dim=-1)
        for wt, bb in zip(Wt[:-1], bt[:-1]):
            a = torch.tanh(a @ wt + bb)
        return a @ Wt[-1] + bt[-1]
    model = TorchModelRollingWindow(f, nx, nu, rolling_window=w, device="cpu")
    integ = DiscretIntegrator(model, H)
    assert not integ.on_device
    with pytest.raises(NotImplementedError):
        RK4Integrator(model, H, 0.1)
    with pytest.raises(AssertionError, match="history window"):
        model.forward(np.zeros((H, nx)), np.zeros((H, nu)))
    for i in range(d["Z"].shape[0]):
        model.set_prev_data(d["hist_x"][i], d["hist_u"][i])
        z, x0 = d["Z"][i], d["X0"][i]
        x, u = z[:H * nx].reshape(H, nx), z[H * nx:].reshape(H, nu)
        np.testing.assert_allclose(integ.forward(x, u, x0), d["g"][i][:H * nx], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(integ.jacobian(x, u, x0), d["jac"][i][:H * nx], rtol=1e-11, atol=1e-12)
        prob = oracle_problem(d, W, b, i)
        Hc = np.tensordot(d["lam"][i][:H * nx], integ.hessian(x, u, x0), axes=1) + float(d["sigma"][i]) * prob.objective_hessian()
        np.testing.assert_allclose(Hc, d["hdense"][i], rtol=1e-9, atol=1e-10)
    with pytest.raises(NotImplementedError):
        TorchModelRollingWindow(f, nx, nu, rolling_window=w, forward_rolling=False, device="cpu")
