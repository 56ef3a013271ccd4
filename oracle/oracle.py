"""ctypes front-end of the CPU fp64 oracle (oracle/mpcq_oracle.cpp).

TEST INFRASTRUCTURE — NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module.  The method names mirror
``mpc_quad_ros_amd.engine.Engine`` so parity tests can drive both side by side.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NX, NU, NY = 13, 4, 17

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def build(native: bool = False) -> str:
    name = "libmpcq_oracle_native.so" if native else "libmpcq_oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "mpcq_oracle.cpp")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        args = ["make", "-C", _HERE] + (["NATIVE=1"] if native else [])
        subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return path


_libs = {}


def load(native: bool = False):
    if native not in _libs:
        lib = ctypes.CDLL(build(native))
        lib.orc_create.restype = ctypes.c_void_p
        lib.orc_create.argtypes = [ctypes.c_void_p]
        for name in ("orc_destroy", "orc_reset"):
            getattr(lib, name).argtypes = [ctypes.c_void_p]
            getattr(lib, name).restype = None
        lib.orc_plant_control_period.restype = ctypes.c_int
        lib.orc_set_threads.restype = ctypes.c_int
        lib.orc_set_threads.argtypes = [ctypes.c_int]
        lib.orc_learn_create.restype = ctypes.c_void_p
        lib.orc_learn_create.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp]
        lib.orc_learn_destroy.argtypes = [ctypes.c_void_p]
        lib.orc_learn_step.argtypes = [ctypes.c_void_p, _dp, _dp]
        lib.orc_learn_get.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, _dp, _dp]
        lib.orc_set_static_gp.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _libs[native] = lib
    return _libs[native]


class OracleEngine:
    """Batched fp64 CPU engine with the same surface as the HIP ``Engine``."""

    def __init__(self, cfg, native: bool = False):
        self.cfg = cfg
        self.lib = load(native)
        self._c = cfg.to_c()
        self.h = ctypes.c_void_p(self.lib.orc_create(ctypes.byref(self._c)))
        self.B, self.N, self.nb = cfg.batch, cfg.N, cfg.nb

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state
    def reset(self):
        self.lib.orc_reset(self.h)

    def set_trajectories(self, traj, lengths=None):
        traj = np.ascontiguousarray(traj, dtype=np.float64)
        assert traj.ndim == 3 and traj.shape[0] == self.B and traj.shape[2] == NX
        if lengths is None:
            lengths = np.full(self.B, traj.shape[1])
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        self.lib.orc_set_trajectories(self.h, _d(traj), _i(lengths), ctypes.c_int(traj.shape[1]))

    def set_reference(self, yref, yrefN):
        yref = np.ascontiguousarray(yref, dtype=np.float64).reshape(self.B, self.N, NY)
        yrefN = np.ascontiguousarray(yrefN, dtype=np.float64).reshape(self.B, NX)
        self.lib.orc_set_reference(self.h, _d(yref), _d(yrefN))

    def set_params(self, mu):
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(self.B, 3 * self.nb)
        self.lib.orc_set_params(self.h, _d(mu))

    def get_state(self):
        B, N, nb = self.B, self.N, self.nb
        s = dict(X=np.zeros((B, N + 1, NX)), U=np.zeros((B, N, NU)), mu=np.zeros((B, 3, nb)),
                 C=np.zeros((B, 3, nb, nb)), x_pred_prev=np.zeros((B, NX)),
                 has_prev=np.zeros(B, np.int32), idx=np.zeros(B, np.int32))
        self.lib.orc_get_state(self.h, _d(s["X"]), _d(s["U"]), _d(s["mu"]), _d(s["C"]), _d(s["x_pred_prev"]),
                               _i(s["has_prev"]), _i(s["idx"]))
        return s

    def set_state(self, X=None, U=None, mu=None, C=None, x_pred_prev=None, has_prev=None, idx=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        g = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        X, U, mu, C, xp, hp, ix = f(X), f(U), f(mu), f(C), f(x_pred_prev), g(has_prev), g(idx)
        self.lib.orc_set_state(self.h, _d(X), _d(U), _d(mu), _d(C), _d(xp), _i(hp), _i(ix))

    # ---- solve path
    def solve(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.B, NX)
        self.lib.orc_solve(self.h, _d(x0))

    def get_x(self, stage):
        out = np.zeros((self.B, NX))
        self.lib.orc_get_x(self.h, ctypes.c_int(stage), _d(out))
        return out

    def get_u(self, stage):
        out = np.zeros((self.B, NU))
        self.lib.orc_get_u(self.h, ctypes.c_int(stage), _d(out))
        return out

    def get_cost(self):
        out = np.zeros(self.B)
        self.lib.orc_get_cost(self.h, _d(out))
        return out

    def get_status(self):
        out = np.zeros(self.B, np.int32)
        self.lib.orc_get_status(self.h, _i(out))
        return out

    def get_qp_iter(self):
        out = np.zeros(self.B, np.int32)
        self.lib.orc_get_qp_iter(self.h, _i(out))
        return out

    def get_kkt(self):
        out = np.zeros(self.B)
        self.lib.orc_get_kkt(self.h, _d(out))
        return out

    def predict_nominal(self, x, u, dt):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.B, NX)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.B, NU)
        out = np.zeros((self.B, NX))
        self.lib.orc_predict_nominal(self.h, _d(x), _d(u), ctypes.c_double(dt), _d(out))
        return out

    def rgp_regress(self, v_body, a_drag):
        vb = np.ascontiguousarray(v_body, dtype=np.float64).reshape(self.B, 3)
        ad = np.ascontiguousarray(a_drag, dtype=np.float64).reshape(self.B, 3)
        self.lib.orc_rgp_regress(self.h, _d(vb), _d(ad))

    def get_rgp(self):
        mu = np.zeros((self.B, 3, self.nb))
        C = np.zeros((self.B, 3, self.nb, self.nb))
        self.lib.orc_get_rgp(self.h, _d(mu), _d(C))
        return mu, C

    def get_kx(self):
        Kx = np.zeros((3, self.nb, self.nb))
        Kxi = np.zeros((3, self.nb, self.nb))
        self.lib.orc_get_kx(self.h, _d(Kx), _d(Kxi))
        return Kx, Kxi

    def step(self, x_meas):
        x = np.ascontiguousarray(x_meas, dtype=np.float64).reshape(self.B, NX)
        w = np.zeros((self.B, NU))
        xp = np.zeros((self.B, NX))
        self.lib.orc_step(self.h, _d(x), _d(w), _d(xp))
        return w, xp

    def get_finished(self):
        out = np.zeros(self.B, np.int32)
        self.lib.orc_get_finished(self.h, _i(out))
        return out

    def get_command(self):
        rotor, coll, rates = np.zeros((self.B, NU)), np.zeros(self.B), np.zeros((self.B, 3))
        self.lib.orc_get_command(self.h, _d(rotor), _d(coll), _d(rates))
        return rotor, coll, rates

    def set_static_gp(self, on=True):
        """use_gp = 1: the GP in the model is fixed (mu = training responses via set_params), no regress in the loop."""
        self.lib.orc_set_static_gp(self.h, int(bool(on)))

    def set_threads(self, n):
        return self.lib.orc_set_threads(int(n))

    def get_tracking_stats(self):
        out = np.zeros(4)
        self.lib.orc_get_tracking_stats(self.h, _d(out))
        return out

    # ---- plant harness
    def plant_update(self, x, u, dt):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.B, NX).copy()
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.B, NU)
        self.lib.orc_plant_update(self.h, _d(x), _d(u), ctypes.c_double(dt))
        return x

    def plant_control_period(self, x, u, control_dt, sim_dt=5e-3):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.B, NX).copy()
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.B, NU)
        n = self.lib.orc_plant_control_period(self.h, _d(x), _d(u), ctypes.c_double(control_dt), ctypes.c_double(sim_dt))
        return x, n

    # ---- single-instance pieces (use instance 0's model)
    def model_f(self, x, u, mu=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        mu = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
        f = np.zeros(NX)
        J = np.zeros((NX, NY))
        self.lib.orc_model_f(self.h, _d(x), _d(u), _d(mu), _d(f), _d(J))
        return f, J

    def rk4_sens(self, x, u, mu, h):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        mu = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
        phi = np.zeros(NX)
        AB = np.zeros((NX, NY))
        self.lib.orc_rk4_sens(self.h, _d(x), _d(u), _d(mu), ctypes.c_double(h), _d(phi), _d(AB))
        return phi, AB


class OracleLearner:
    """B x 3 recursive GPs with hyper-parameter learning (RGP.learn, src/gp/RGP.py:332-505), fp64 CPU restatement."""

    def __init__(self, batch, basis, theta):
        self.lib = load()
        self.basis = np.ascontiguousarray(basis, dtype=np.float64).reshape(3, -1)
        th = np.asarray(theta, dtype=np.float64)
        self.theta = np.ascontiguousarray(np.tile(th, (3, 1)) if th.shape == (3,) else th.reshape(3, 3))
        self.B, self.nb = batch, self.basis.shape[1]
        self.h = ctypes.c_void_p(self.lib.orc_learn_create(batch, self.nb, _d(self.basis), _d(self.theta)))

    def step(self, s, y):
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(self.B, 3)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(self.B, 3)
        self.lib.orc_learn_step(self.h, _d(s), _d(y))

    def get(self):
        B, n = self.B, self.nb
        out = dict(mu_g=np.zeros((B, 3, n)), C_g=np.zeros((B, 3, n, n)), mu_eta=np.zeros((B, 3, 3)), C_eta=np.zeros((B, 3, 3, 3)),
                   K_x_inv=np.zeros((B, 3, n, n)))
        self.lib.orc_learn_get(self.h, _d(out["mu_g"]), _d(out["C_g"]), _d(out["mu_eta"]), _d(out["C_eta"]), _d(out["K_x_inv"]))
        return out

    def close(self):
        if self.h:
            self.lib.orc_learn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def compute_a_drag(x, x_pred_minus_1, dt):
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    xp = np.ascontiguousarray(x_pred_minus_1, dtype=np.float64)
    vb, ad = np.zeros(3), np.zeros(3)
    lib.orc_compute_a_drag(_d(x), _d(xp), ctypes.c_double(dt), _d(vb), _d(ad))
    return vb, ad


def reference_chunk(traj, idx, N, skip=1):
    lib = load()
    traj = np.ascontiguousarray(traj, dtype=np.float64)
    out = np.zeros((N, NX))
    lib.orc_reference_chunk(_d(traj), ctypes.c_int(traj.shape[0]), ctypes.c_int(idx), ctypes.c_int(N), ctypes.c_int(skip), _d(out))
    return out
