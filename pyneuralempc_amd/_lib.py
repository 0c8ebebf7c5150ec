"""ctypes binding of libnempc.so (include/nempc.h).  There is no CPU fallback: if the library is
missing or no HIP device is visible the callers fail with a clear error."""
import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# NEMPC_LIB: an alternative build of the same library (A/B kernel experiments, tools/build_variant.py)
LIB_PATH = os.environ.get("NEMPC_LIB") or os.path.join(PKG, "libnempc.so")

ABI_VERSION = 7
COMM_ID_BYTES = 128
MAX_LAYERS = 8
F64, F32 = 0, 1
DISCRET, UNITY, RK4 = 0, 1, 2
KERNEL_AUTO, KERNEL_VALU, KERNEL_MFMA, KERNEL_MFMA_TILE, KERNEL_LAYERED = 0, 1, 2, 3, 4
KERNEL_NAMES = {"auto": KERNEL_AUTO, "valu": KERNEL_VALU, "mfma": KERNEL_MFMA, "mfma_tile": KERNEL_MFMA_TILE,
                "layered": KERNEL_LAYERED}
INTEGRATOR_IDS = {"discret": DISCRET, "unity": UNITY, "rk4": RK4}
# NEMPC_ACT_*: the activation of a dense layer (names as Keras spells them)
ACTIVATION_IDS = {"linear": 0, "tanh": 1, "relu": 2, "sigmoid": 3, "softplus": 4, "elu": 5, "leaky_relu": 6, "selu": 7,
                  "swish": 8, "gelu": 9, "softsign": 10, "mish": 11, "exponential": 12, "relu6": 13}
ACTIVATION_ALIASES = {"silu": "swish"}       # (Keras: swish and silu are the same function)
# activations with a parameter (alpha), written "name:value" ("elu:0.5", "leaky_relu:0.1"); the bare name takes the default
# (elu: Keras' 1.0; leaky_relu: keras.activations.leaky_relu's 0.2 -- the LeakyReLU LAYER carries its own negative_slope)
ACTIVATION_DEFAULT_PARAM = {"elu": 1.0, "leaky_relu": 0.2}


def split_activation(spec):
    """"name" | "name:value" -> (name, parameter), validated like nempc_create does"""
    name, _, val = str(spec).partition(":")
    name = ACTIVATION_ALIASES.get(name, name)
    if name not in ACTIVATION_IDS:
        raise NotImplementedError(f"activation '{spec}' is not supported on the device path (supported: "
                                  f"{', '.join(ACTIVATION_IDS)})")
    if val and name not in ACTIVATION_DEFAULT_PARAM:
        raise ValueError(f"activation '{name}' takes no parameter ('{spec}')")
    par = float(val) if val else ACTIVATION_DEFAULT_PARAM.get(name, 0.0)
    if name == "elu" and not par > 0.0:
        raise ValueError("elu needs alpha > 0")
    if name == "leaky_relu" and not par >= 0.0:
        raise ValueError("leaky_relu needs alpha >= 0")
    return name, par

EXPORTS = ["nempc_create", "nempc_destroy", "nempc_reserve", "nempc_set_weights", "nempc_set_objective", "nempc_set_terminal_weight", "nempc_set_box_rows", "nempc_bind_extra", "nempc_bind_history",
           "nempc_dims", "nempc_constraint_bounds", "nempc_jac_structure", "nempc_hess_structure", "nempc_eval",
           "nempc_hess", "nempc_hess_gn", "nempc_solve", "nempc_sync", "nempc_kernel_variant", "nempc_last_row_kernel", "nempc_last_hess_kernel", "nempc_last_error", "nempc_abi_version",
           "nempc_plan_grid", "nempc_num_cus",
           "nempc_comm_unique_id", "nempc_comm_init", "nempc_allgather_u0", "nempc_comm_size", "nempc_comm_destroy"]


class NempcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libnempc error {code}: {message}")
        self.code = code


class NempcConfig(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int32), ("device", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("integrator", ctypes.c_int32), ("H", ctypes.c_int32), ("nx", ctypes.c_int32),
                ("nu", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("widths", ctypes.c_int32 * MAX_LAYERS),
                ("max_batch", ctypes.c_int32), ("kernel", ctypes.c_int32), ("n_extra", ctypes.c_int32),
                ("rolling_window", ctypes.c_int32), ("rolling_reverse", ctypes.c_int32), ("DT", ctypes.c_double),
                ("activations", ctypes.c_int32 * MAX_LAYERS), ("act_param", ctypes.c_double * MAX_LAYERS)]


class NempcSolverOpts(ctypes.Structure):
    _fields_ = [("max_iter", ctypes.c_int32), ("max_linesearch", ctypes.c_int32), ("check_every", ctypes.c_int32),
                ("lq_kernel", ctypes.c_int32), ("tol_constraint", ctypes.c_double), ("tol_step", ctypes.c_double),
                ("mu_init", ctypes.c_double), ("mu_min", ctypes.c_double), ("mu_factor", ctypes.c_double),
                ("reg", ctypes.c_double), ("compact", ctypes.c_int32), ("barrier", ctypes.c_int32),
                ("iters_out", ctypes.c_void_p), ("linesearch", ctypes.c_int32), ("lq_attempts", ctypes.c_int32)]


_lib = None


def load():
    """Load libnempc.so once and declare every prototype of include/nempc.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m pyneuralempc_amd._build` "
                           "(or __graft_entry__.build()); pyneuralempc_amd has no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, dp = ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int32)
    dpp = ctypes.POINTER(dp)
    lib.nempc_create.argtypes = [ctypes.POINTER(NempcConfig), ctypes.POINTER(vp)]
    lib.nempc_destroy.argtypes = [vp]
    lib.nempc_reserve.argtypes = [vp, i32]
    lib.nempc_set_weights.argtypes = [vp, dpp, dpp]
    lib.nempc_set_objective.argtypes = [vp, dp, dp, dp, dp, dp, dp]
    lib.nempc_set_terminal_weight.argtypes = [vp, dp]
    lib.nempc_set_box_rows.argtypes = [vp, ctypes.c_int, dp, dp]
    lib.nempc_bind_extra.argtypes = [vp, vp, i32]
    lib.nempc_bind_history.argtypes = [vp, vp, vp, i32]
    lib.nempc_dims.argtypes = [vp, ip, ip, ip, ip]
    lib.nempc_constraint_bounds.argtypes = [vp, dp, dp]
    lib.nempc_jac_structure.argtypes = [vp, ip, ip]
    lib.nempc_hess_structure.argtypes = [vp, ip, ip]
    lib.nempc_eval.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nempc_hess.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nempc_hess_gn.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.nempc_solve.argtypes = [vp, i32, vp, vp, dp, dp, ctypes.POINTER(NempcSolverOpts), vp, ip, vp]
    lib.nempc_sync.argtypes = [vp, vp]
    lib.nempc_kernel_variant.argtypes = [vp]
    lib.nempc_last_row_kernel.argtypes = [vp]
    lib.nempc_last_hess_kernel.argtypes = [vp]
    lib.nempc_plan_grid.argtypes = [i32, i32, i32, ip, ip, ip]
    lib.nempc_num_cus.argtypes = [vp]
    lib.nempc_comm_unique_id.argtypes = [vp]
    lib.nempc_comm_init.argtypes = [vp, i32, i32, vp]
    lib.nempc_allgather_u0.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    lib.nempc_comm_size.argtypes = [vp, ip, ip]
    lib.nempc_comm_destroy.argtypes = [vp]
    lib.nempc_last_error.argtypes = []
    lib.nempc_last_error.restype = ctypes.c_char_p
    lib.nempc_abi_version.argtypes = []
    for name in EXPORTS:
        if name != "nempc_last_error":
            getattr(lib, name).restype = ctypes.c_int
    if lib.nempc_abi_version() != ABI_VERSION:
        raise RuntimeError("libnempc.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise NempcError(rc, load().nempc_last_error().decode("utf-8", "replace"))
