"""Quadrotor constants, OCP weights and engine configuration.

Plain data shared by the host facade, the tests and the bench.  Values and their
provenance in the reference (paths relative to the reference checkout):

* hummingbird(): ``Quadrotor3D.set_parameters_from_file`` src/quad.py:385-417 applied to
  config/hummingbird.xacro:29-50 ('+' layout, flipped z_l_tau sign, src/quad.py:414-417).
* legacy_sim(): the constants the shipped python-simulation logs were produced with
  (mass 1.0, max_thrust 20, J=[.03,.03,.06], arm .235, c .013) — the *current*
  defaults at src/quad.py:41-67 (mass 0.03, arm 0.04) match no log.
* weights: src/quad_opt.py:122-130; input bounds src/quad_opt.py:142-144; hover reference
  u_ref = 0.16 src/quad_opt.py:304.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

NX, NU, NY = 13, 4, 17


@dataclass
class QuadParams:
    mass: float
    J: Sequence[float]
    max_thrust: float
    x_f: Sequence[float]
    y_f: Sequence[float]
    z_l_tau: Sequence[float]
    g: float = 9.81                      # src/quad.py:73
    rotor_drag: Sequence[float] = (0.3, 0.3, 0.0)   # src/quad.py:79-84 (plant only)
    aero_drag: float = 0.008             # src/quad.py:89 (plant only)


def hummingbird() -> QuadParams:
    L, c = 0.17, 0.016
    return QuadParams(
        mass=0.68 + 4 * 0.009,
        J=(0.007, 0.007, 0.012),
        max_thrust=838.0 ** 2 * 8.54858e-06,
        x_f=(L, 0.0, -L, 0.0),
        y_f=(0.0, L, 0.0, -L),
        z_l_tau=(c, -c, c, -c),          # -[-c, c, -c, c], src/quad.py:417
    )


def legacy_sim() -> QuadParams:
    L, c = 0.47 / 2, 0.013
    return QuadParams(
        mass=1.0,
        J=(0.03, 0.03, 0.06),
        max_thrust=20.0,
        x_f=(L, 0.0, -L, 0.0),
        y_f=(0.0, L, 0.0, -L),
        z_l_tau=(-c, c, -c, c),
    )


def default_W() -> np.ndarray:
    # q_cost with the mean attitude weight inserted for the 4th quaternion component, then r_cost
    q_cost = np.array([10, 10, 10, 0.1, 0.1, 0.1, 0.05, 0.05, 0.05, 0.05, 0.05, 0.05])
    q_diag = np.concatenate((q_cost[:3], np.mean(q_cost[3:6])[np.newaxis], q_cost[3:]))
    return np.concatenate((q_diag, np.full(4, 0.1)))


class CTuning(ctypes.Structure):
    """Binary layout of ``mpcq_tuning`` (include/mpcq.h); every field 0 = default."""
    _fields_ = [
        ("warm_max", ctypes.c_int32), ("warm_retry", ctypes.c_int32), ("flip_max", ctypes.c_int32),
        ("abort_pins", ctypes.c_int32), ("abort_wrong", ctypes.c_int32), ("polish_max", ctypes.c_int32),
        ("stage_mem", ctypes.c_int32), ("generic_kernel", ctypes.c_int32),
        ("pin_ratio", ctypes.c_double), ("ipm_mu0", ctypes.c_double), ("ipm_margin", ctypes.c_double), ("ipm_tol", ctypes.c_double),
        ("block_order", ctypes.c_int32),     # since 0.4
        ("groups", ctypes.c_int32),          # since 0.6 (0.4 / 0.5: reserved, 0)
    ]


STAGE_MEM = {"auto": 0, "lds": 1, "global": 2, "compact": 3}


class CConfig(ctypes.Structure):
    """Binary layout of ``mpcq_config`` (include/mpcq.h); the oracle's ``orc_config`` has the
    same leading fields (it ignores device / precision / qp options)."""
    _fields_ = [
        ("batch", ctypes.c_int32), ("N", ctypes.c_int32), ("nb", ctypes.c_int32), ("skip", ctypes.c_int32),
        ("T", ctypes.c_double), ("dt_pred", ctypes.c_double),
        ("mass", ctypes.c_double), ("J", ctypes.c_double * 3), ("max_thrust", ctypes.c_double),
        ("x_f", ctypes.c_double * 4), ("y_f", ctypes.c_double * 4), ("z_l_tau", ctypes.c_double * 4),
        ("g", ctypes.c_double),
        ("rotor_drag", ctypes.c_double * 3), ("aero_drag", ctypes.c_double),
        ("W", ctypes.c_double * 17), ("W_e", ctypes.c_double * 13),
        ("u_lb", ctypes.c_double * 4), ("u_ub", ctypes.c_double * 4), ("u_ref", ctypes.c_double * 4),
        ("qp_tol", ctypes.c_double),
        ("basis", ctypes.POINTER(ctypes.c_double)), ("theta", ctypes.POINTER(ctypes.c_double)),
        # --- product-only tail (the oracle's struct ends above)
        ("device", ctypes.c_int32), ("precision", ctypes.c_int32),
        ("qp_max_iter", ctypes.c_int32), ("flags", ctypes.c_int32),
        ("finish_radius", ctypes.c_double),
        ("tune", CTuning),
    ]


PRECISION_F64 = 0
PRECISION_F32 = 1


@dataclass
class EngineConfig:
    """One engine = B independent quadrotors sharing (N, T, nb, quad constants, theta)."""
    batch: int
    N: int = 20                       # n_nodes
    T: float = 1.0                    # t_lookahead
    quad: QuadParams = field(default_factory=hummingbird)
    nb: int = 0                       # basis points per axis (0: no GP in the model)
    basis: Optional[np.ndarray] = None    # [3, nb]
    theta: Optional[np.ndarray] = None    # [3, 3] rows = (L, sigma_f, sigma_n) per axis
    dt_pred: float = 0.01             # ODOMETRY_DT (node) / optimization_dt (python sim)
    skip: Optional[int] = None        # control_freq_factor; default int((T/N)/0.01) as the node
    W: np.ndarray = field(default_factory=default_W)
    W_e: Optional[np.ndarray] = None
    u_lb: Sequence[float] = (0.0, 0.0, 0.0, 0.0)
    u_ub: Sequence[float] = (1.0, 1.0, 1.0, 1.0)
    u_ref: Sequence[float] = (0.16, 0.16, 0.16, 0.16)
    qp_tol: float = 0.0               # 0 -> implementation default
    device: int = 0
    precision: int = PRECISION_F64
    qp_max_iter: int = 0              # 0 -> implementation default
    static_gp: bool = False           # MPCQ_FLAG_STATIC_GP: fixed GP in the model (use_gp = 1), no recursive update in the step
    finish_radius: float = 0.0        # EPSILON_TRAJECTORY_FINISHED [m]; 0 -> 1.0 (src/mpc_controller_node.py:118)
    tune: Optional[dict] = None       # mpcq_tuning fields by name (include/mpcq.h); stage_mem also as "lds" / "global"

    def __post_init__(self):
        if self.skip is None:
            # src/mpc_controller_node.py:222
            self.skip = int((self.T / self.N) / 0.01)
        if self.W_e is None:
            self.W_e = np.asarray(self.W)[:NX].copy()
        if self.nb:
            if self.basis is None:
                raise ValueError("nb > 0 needs basis[3, nb]")
            self.basis = np.ascontiguousarray(np.asarray(self.basis, dtype=np.float64).reshape(3, self.nb))
            if self.theta is None:
                self.theta = np.tile(np.array([1.0, 0.1, 0.1]), (3, 1))   # RGP default, src/gp/RGP.py:106
            th = np.asarray(self.theta, dtype=np.float64)
            if th.shape == (3,):
                th = np.tile(th, (3, 1))
            self.theta = np.ascontiguousarray(th.reshape(3, 3))
        else:
            self.basis = np.zeros((3, 0))
            self.theta = np.zeros((3, 3))

    @property
    def optimization_dt(self) -> float:
        return self.T / self.N

    def to_c(self) -> CConfig:
        c = CConfig()
        c.batch, c.N, c.nb, c.skip = self.batch, self.N, self.nb, int(self.skip)
        c.T, c.dt_pred = float(self.T), float(self.dt_pred)
        q = self.quad
        c.mass, c.max_thrust, c.g, c.aero_drag = q.mass, q.max_thrust, q.g, q.aero_drag
        c.J[:] = list(q.J); c.x_f[:] = list(q.x_f); c.y_f[:] = list(q.y_f); c.z_l_tau[:] = list(q.z_l_tau)
        c.rotor_drag[:] = list(q.rotor_drag)
        c.W[:] = [float(v) for v in self.W]; c.W_e[:] = [float(v) for v in self.W_e]
        c.u_lb[:] = list(self.u_lb); c.u_ub[:] = list(self.u_ub); c.u_ref[:] = list(self.u_ref)
        c.qp_tol = float(self.qp_tol)
        self._basis_buf = np.ascontiguousarray(self.basis.ravel(), dtype=np.float64)
        self._theta_buf = np.ascontiguousarray(self.theta.ravel(), dtype=np.float64)
        c.basis = self._basis_buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        c.theta = self._theta_buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        c.device, c.precision, c.qp_max_iter, c.flags = self.device, self.precision, self.qp_max_iter, (1 if self.static_gp else 0)
        c.finish_radius = float(self.finish_radius)
        for k, v in (self.tune or {}).items():
            if k not in dict(CTuning._fields_):
                raise ValueError(f"unknown tuning field {k!r}")
            if k == "stage_mem" and isinstance(v, str):
                v = STAGE_MEM[v]
            setattr(c.tune, k, v)
        return c


def static_gp_theta(theta):
    """Hyper-parameters of a static GP (src/gp/GP.py:113-130: K + (noise + 1e-7) I with noise = theta[-1], NOT squared) in
    the engine's convention K_x = K + sigma_n^2 I: [L, sigma_f, sqrt(noise + 1e-7)]."""
    th = np.asarray(theta, dtype=np.float64)
    if th.ndim == 1:
        return np.array([th[0], th[-2], np.sqrt(th[-1] + 1e-7)])
    return np.stack([static_gp_theta(t) for t in th])


def rgp_basis_linspace(v_max: float, nb: int) -> np.ndarray:
    """Node basis: np.linspace(-v_max, v_max, nb) on each axis, src/mpc_controller_node.py:212."""
    return np.tile(np.linspace(-v_max, v_max, nb), (3, 1))
