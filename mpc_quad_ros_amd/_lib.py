"""Loader of libmpcq.so (the HIP/gfx950 engine behind include/mpcq.h).

There is no CPU implementation of the engine in this package: if the shared library is missing
or no MI355X is visible, construction fails loudly (``MpcqError``)."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libmpcq.so")

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p

# every symbol declared in include/mpcq.h: (name, restype, argtypes)
SYMBOLS = [
    ("mpcq_last_error", ctypes.c_char_p, []),
    ("mpcq_version", ctypes.c_char_p, []),
    ("mpcq_create", ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    ("mpcq_create_sized", ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.POINTER(_vp)]),
    ("mpcq_destroy", ctypes.c_int, [_vp]),
    ("mpcq_reset", ctypes.c_int, [_vp]),
    ("mpcq_set_trajectories", ctypes.c_int, [_vp, _dp, _ip, ctypes.c_int32]),
    ("mpcq_set_reference", ctypes.c_int, [_vp, _dp, _dp]),
    ("mpcq_set_params", ctypes.c_int, [_vp, _dp]),
    ("mpcq_solve", ctypes.c_int, [_vp, _dp]),
    ("mpcq_get_x", ctypes.c_int, [_vp, ctypes.c_int32, _dp]),
    ("mpcq_get_u", ctypes.c_int, [_vp, ctypes.c_int32, _dp]),
    ("mpcq_get_cost", ctypes.c_int, [_vp, _dp]),
    ("mpcq_get_status", ctypes.c_int, [_vp, _ip]),
    ("mpcq_get_qp_iter", ctypes.c_int, [_vp, _ip]),
    ("mpcq_get_qp_work", ctypes.c_int, [_vp, _ip]),
    ("mpcq_get_stats", ctypes.c_int, [_vp, _dp]),
    ("mpcq_predict_nominal", ctypes.c_int, [_vp, _dp, _dp, ctypes.c_double, _dp]),
    ("mpcq_rgp_regress", ctypes.c_int, [_vp, _dp, _dp]),
    ("mpcq_get_rgp", ctypes.c_int, [_vp, _dp, _dp]),
    ("mpcq_step", ctypes.c_int, [_vp, _dp, _dp, _dp]),
    ("mpcq_step_device_async", ctypes.c_int, [_vp, _vp, _vp]),
    ("mpcq_synchronize", ctypes.c_int, [_vp]),
    ("mpcq_stream", _vp, [_vp]),
    ("mpcq_get_command", ctypes.c_int, [_vp, _dp, _dp, _dp]),
    ("mpcq_get_finished", ctypes.c_int, [_vp, _ip]),
    ("mpcq_get_reference_chunk", ctypes.c_int, [_vp, _dp]),
    ("mpcq_plant_substeps", ctypes.c_int, [ctypes.c_double, ctypes.c_double]),
    ("mpcq_sim_plant_period", ctypes.c_int, [_vp, _dp, ctypes.c_double, ctypes.c_double, _ip]),
    ("mpcq_sim_control_periods", ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_double, ctypes.c_double, _ip]),
    ("mpcq_sim_reset", ctypes.c_int, [_vp, _dp]),
    ("mpcq_sim_steps", ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_double]),
    ("mpcq_sim_run", ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_double]),
    ("mpcq_sim_get_state", ctypes.c_int, [_vp, _dp, _dp]),
    ("mpcq_get_kernel_time", ctypes.c_int, [_vp, _dp, _ip]),
    ("mpcq_get_kernel_time_minmax", ctypes.c_int, [_vp, _dp, _dp]),
    ("mpcq_debug_profile", ctypes.c_int, [_vp, _vp]),
    ("mpcq_get_block_order", ctypes.c_int, [_vp, _ip]),
    ("mpcq_get_groups", ctypes.c_int, [_vp, _ip]),
    ("mpcq_get_tracking_stats", ctypes.c_int, [_vp, _dp]),
    ("mpcq_comm_unique_id", ctypes.c_int, [_vp]),
    ("mpcq_comm_init", ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, _vp]),
    ("mpcq_comm_share", ctypes.c_int, [_vp, _vp]),
    ("mpcq_allreduce_tracking_stats", ctypes.c_int, [_vp, _dp]),
    ("mpcq_get_state", ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _ip, _ip]),
    ("mpcq_set_state", ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _ip, _ip]),
    ("mpcq_get_solver_state", ctypes.c_int, [_vp, _ip, _dp, _ip]),
    ("mpcq_set_solver_state", ctypes.c_int, [_vp, _ip, _dp, _ip]),
    ("mpcq_learn_last_error", ctypes.c_char_p, []),
    ("mpcq_learn_create", ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, _dp, _dp, ctypes.c_int32, ctypes.POINTER(_vp)]),
    ("mpcq_learn_destroy", ctypes.c_int, [_vp]),
    ("mpcq_learn_step", ctypes.c_int, [_vp, _dp, _dp]),
    ("mpcq_learn_get", ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp]),
]


class MpcqError(RuntimeError):
    pass


_cache = {}


def load(path: str | None = None):
    path = os.path.abspath(path or os.environ.get("MPCQ_LIB") or DEFAULT_LIB)   # MPCQ_LIB: experiment builds (csrc/Makefile `variant`)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise MpcqError(
            f"{path} not found: build it with `make -C mpc_quad_ros_amd/csrc` (hipcc, gfx950). "
            "This package has no CPU implementation of the control step.")
    lib = ctypes.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _cache[path] = lib
    return lib


def d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def i(a):
    return None if a is None else a.ctypes.data_as(_ip)
