"""ctypes binding of the C ABI in include/isokann_hip.h (the same entry points a Julia
host binds with `ccall`).  No compute happens in Python.

The library is built in-tree by `__graft_entry__.build()` into
`isokann.jl_amd/lib/libisokann_hip.so`.  `load()` fails loudly when it is missing:
there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ISK_MAX_LAYERS = 8
ISK_ABI_VERSION = 2

OK, EINVAL, EHIP, ENOMEM, ECONSTANT_CHI, ECOLLAPSED, ESINGULAR, EUNSUPPORTED, ESTATE, EPEER = range(10)
ACT = {"identity": 0, "sigmoid": 1, "tanh": 2, "relu": 3, "swish": 4}
FEAT_IDENTITY, FEAT_ALLPAIRS, FEAT_PAIRLIST = 0, 1, 2
OPT = {"adam": 0, "nesterov": 1}
TARGET_SHIFTSCALE, TARGET_ISA, TARGET_HOST = 0, 1, 2
LN_MODE = {"var_eps2": 0, "std_eps": 1, "var_eps": 2}
T_FEATURIZE, T_FWD_YS, T_FWD_XS, T_TARGET, T_TRAIN, T_ALLREDUCE, T_COUNT = 0, 1, 2, 3, 4, 5, 6
TIMER_NAMES = ("featurize", "fwd_ys", "fwd_xs", "target", "train", "allreduce")

LIB_PATH = os.environ.get("ISK_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libisokann_hip.so")


class IskError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"isokann_hip status {status}: {msg}")
        self.status = status


class DomainError(ValueError):
    """Julia's DomainError, re-raised for ISK_ECONSTANT_CHI / ISK_ECOLLAPSED / ISK_ESINGULAR
    (src/isotarget.jl:39, src/iso.jl:186-189, src/isotarget.jl:94-97 of the reference)."""
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32),
        ("n_layers", C.c_int32), ("widths", C.c_int32 * (ISK_MAX_LAYERS + 1)),
        ("activation", C.c_int32), ("lastactivation", C.c_int32),
        ("layernorm", C.c_int32), ("ln_mode", C.c_int32), ("ln_eps", C.c_float),
        ("opt_kind", C.c_int32), ("lr", C.c_float), ("reg", C.c_float), ("beta1", C.c_float),
        ("beta2", C.c_float), ("opt_eps", C.c_float), ("rho", C.c_float),
        ("featurizer", C.c_int32), ("n_atoms", C.c_int32), ("n_pairs", C.c_int32),
        ("pairs", C.POINTER(C.c_int32)),
        ("featurize_every_iteration", C.c_int32),
        ("target_kind", C.c_int32), ("isa_permute", C.c_int32), ("isa_whitening", C.c_int32),
        ("minibatch", C.c_int32), ("seed", C.c_uint64),
        ("forward_mode", C.c_int32),
        ("force_layered", C.c_int32),
    ]


_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_H = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/isokann_hip.h
PROTOTYPES = {
    "isk_abi_version": (C.c_int, []),
    "isk_device_count": (C.c_int, [_i32p]),
    "isk_last_error": (C.c_char_p, []),
    "isk_status_string": (C.c_char_p, [C.c_int]),
    "isk_config_default": (None, [C.POINTER(Config)]),
    "isk_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "isk_destroy": (None, [_H]),
    "isk_synchronize": (C.c_int, [_H]),
    "isk_set_stream": (C.c_int, [_H, C.c_void_p]),
    "isk_set_option": (C.c_int, [_H, C.c_char_p, C.c_int64]),
    "isk_get_option": (C.c_int, [_H, C.c_char_p, _i64p]),
    "isk_nparams": (C.c_int, [_H, _i64p]),
    "isk_set_params": (C.c_int, [_H, _f32p, C.c_int64]),
    "isk_get_params": (C.c_int, [_H, _f32p, C.c_int64]),
    "isk_set_opt_state": (C.c_int, [_H, _f32p, _f32p, C.c_int64, _f64p]),
    "isk_get_opt_state": (C.c_int, [_H, _f32p, _f32p, C.c_int64, _f64p]),
    "isk_reset_opt_state": (C.c_int, [_H]),
    "isk_set_data": (C.c_int, [_H, _f32p, _f32p, C.c_int64, C.c_int64, C.c_int64]),
    "isk_set_data_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "isk_append_data": (C.c_int, [_H, _f32p, _f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "isk_set_koopman_weights": (C.c_int, [_H, _f32p, C.c_int64, C.c_int64]),
    "isk_data_shape": (C.c_int, [_H, _i64p, _i64p, _i64p]),
    "isk_features": (C.c_int, [_H, _f32p, C.c_int64, C.c_int64, _f32p]),
    "isk_forward": (C.c_int, [_H, _f32p, C.c_int64, C.c_int64, _f32p]),
    "isk_chicoords": (C.c_int, [_H, _f32p, C.c_int64, C.c_int64, _f32p]),
    "isk_chis": (C.c_int, [_H, _f32p]),
    "isk_koopman": (C.c_int, [_H, _f32p]),
    "isk_koopman_f64": (C.c_int, [_H, _f64p]),
    "isk_koopman_coords": (C.c_int, [_H, _f32p, C.c_int64, C.c_int64, C.c_int64, _f32p]),
    "isk_validationloss": (C.c_int, [_H, _f32p, _f32p, C.c_int64, C.c_int64, C.c_int64, _f64p]),
    "isk_chicoords_pullback": (C.c_int, [_H, _f32p, C.c_int64, C.c_int64, _f32p, _f32p]),
    "isk_target": (C.c_int, [_H, _f32p]),
    "isk_train_batch": (C.c_int, [_H, _f32p, C.c_int32, _i32p, _f64p]),
    "isk_batch_grads": (C.c_int, [_H, _f32p, _i32p, C.c_int32, _f32p, _f64p]),
    "isk_run": (C.c_int, [_H, C.c_int32, C.c_int32, _i32p, _f64p]),
    "isk_set_minibatch": (C.c_int, [_H, C.c_int32]),
    "isk_dp_koopman_local": (C.c_int, [_H]),
    "isk_dp_minmax_local": (C.c_int, [_H, _f32p]),
    "isk_dp_shiftscale_apply": (C.c_int, [_H, C.c_float, C.c_float]),
    "isk_dp_get_ks_chi": (C.c_int, [_H, _f32p, _f32p]),
    "isk_dp_isa_apply": (C.c_int, [_H, _f64p, _i32p]),
    "isk_dp_set_loss_weights": (C.c_int, [_H, _f32p]),
    "isk_dp_ks_chi_dev": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i64p]),
    "isk_dp_isa_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "isk_dp_check_target": (C.c_int, [_H]),
    "isk_dp_target_moments": (C.c_int, [_H, _f64p, _f64p]),
    "isk_isa_solve": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _f64p, _i32p]),
    "isk_dp_grad_buffer": (C.c_int, [_H, C.POINTER(C.c_void_p), _i64p]),
    "isk_dp_epoch_begin": (C.c_int, [_H, _i32p, C.c_int32]),
    "isk_dp_local_grads": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int64]),
    "isk_dp_apply": (C.c_int, [_H]),
    "isk_dp_epoch_end": (C.c_int, [_H, _f64p]),
    "isk_dp_comm_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "isk_dp_comm_init": (C.c_int, [_H, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
    "isk_dp_run": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _i32p, _f64p]),
    "isk_dp_create_all": (C.c_int, [C.POINTER(Config), _i32p, C.c_int32, C.POINTER(_H)]),
    "isk_dp_set_params_all": (C.c_int, [C.POINTER(_H), C.c_int32, _f32p, C.c_int64]),
    "isk_dp_set_data_all": (C.c_int, [C.POINTER(_H), C.c_int32, _f32p, _f32p, C.c_int64, C.c_int64, C.c_int64]),
    "isk_dp_run_all": (C.c_int, [C.POINTER(_H), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_i32p), _f64p]),
    "isk_dp_destroy_all": (None, [C.POINTER(_H), C.c_int32]),
    "isk_timers_enable": (C.c_int, [_H, C.c_int32]),
    "isk_timers_read": (C.c_int, [_H, _f64p, _i64p]),
    "isk_timers_reset": (C.c_int, [_H]),
}

_lib = None


def load():
    """dlopen the HIP library and bind every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built (run __graft_entry__.build()). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.isk_abi_version() != ISK_ABI_VERSION:
        raise RuntimeError("isokann_hip ABI version mismatch")
    _lib = lib
    return lib


def device_count() -> int:
    n = C.c_int32(0)
    st = load().isk_device_count(C.byref(n))
    return int(n.value) if st == OK else 0


def check(status: int):
    if status == OK:
        return
    lib = load()
    msg = (lib.isk_last_error() or b"").decode() or lib.isk_status_string(status).decode()
    if status in (ECONSTANT_CHI, ECOLLAPSED, ESINGULAR):
        raise DomainError(status, msg)
    raise IskError(status, msg)


def f32(a) -> np.ndarray:
    """Julia-layout (column-major) float32 buffer of `a`, as `ccall` would pass it"""
    return np.asfortranarray(a, dtype=np.float32)


def ptr_f32(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def ptr_i32(a: np.ndarray):
    return a.ctypes.data_as(_i32p)


def ptr_f64(a: np.ndarray):
    return a.ctypes.data_as(_f64p)
