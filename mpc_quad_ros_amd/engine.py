"""Batched engine handle over the C ABI (include/mpcq.h): B quadrotors advanced in lockstep on
one MI355X, all state resident in HBM.  Thin: every method is one C call plus numpy marshalling."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .params import NU, NX, NY, EngineConfig


# decimal fields of mpcq_get_qp_iter (include/mpcq.h)
def qp_passes(it):
    """Riccati factorisations of the solve (active-set passes + interior-point iterations)."""
    return np.asarray(it) % 1000


def qp_fallback(it):
    """True where the warm active-set attempt was given up or skipped and the solve went through the interior point."""
    return (np.asarray(it) // 1000) % 10 != 0


def qp_flip(it):
    """True where the solve carries the flip mark (the next one skips the warm attempt)."""
    return (np.asarray(it) // 10000) % 10 != 0


def qp_warm_exit(it):
    """MPCQ_WARM_* code: why the warm attempt ended without a solution (0: it succeeded / there was none)."""
    return np.asarray(it) // 100000


def order_bin(it):
    """Cost bin of mpcq::order_kernel (0 = predicted most expensive) for a qp_iter value of the previous period."""
    it = np.maximum(np.asarray(it), 0)      # (a negative value is not one the solver writes: binned like a cold start, as on the device)
    total = it % 1000
    cost = np.where(it == 0, 15, np.minimum(total, 15))
    cost = np.where(((it // 1000) % 100 != 0) & (cost < 8), 8, cost)
    return 15 - cost


WARM_BUDGET, WARM_PINS, WARM_WRONG, WARM_BOUNCE, WARM_NUMERIC, WARM_SKIPPED = 1, 2, 3, 4, 5, 6
SOLVE_LOW_ACCURACY = 8


class Engine:
    def __init__(self, cfg: EngineConfig, lib_path: str | None = None):
        self.cfg = cfg
        self.lib = _lib.load(lib_path)
        self._c = cfg.to_c()
        h = ctypes.c_void_p()
        self._check(self.lib.mpcq_create_sized(ctypes.byref(self._c), ctypes.sizeof(self._c), ctypes.byref(h)))
        self.h = h
        self.B, self.N, self.nb = cfg.batch, cfg.N, cfg.nb

    def _check(self, rc):
        if rc != 0:
            raise _lib.MpcqError(f"mpcq error {rc}: {self.lib.mpcq_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.mpcq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _f(a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a if shape is None else a.reshape(shape)

    # ---- state
    def reset(self):
        self._check(self.lib.mpcq_reset(self.h))

    def set_trajectories(self, traj, lengths=None):
        traj = self._f(traj)
        if traj.ndim != 3 or traj.shape[0] != self.B or traj.shape[2] != NX:
            raise ValueError(f"traj must be [B={self.B}, T, 13]")
        if lengths is None:
            lengths = np.full(self.B, traj.shape[1])
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        self._check(self.lib.mpcq_set_trajectories(self.h, _lib.d(traj), _lib.i(lengths), traj.shape[1]))

    def set_reference(self, yref, yrefN):
        yref = self._f(yref, (self.B, self.N, NY))
        yrefN = self._f(yrefN, (self.B, NX))
        self._check(self.lib.mpcq_set_reference(self.h, _lib.d(yref), _lib.d(yrefN)))

    def set_params(self, mu):
        mu = self._f(mu, (self.B, 3 * self.nb))
        self._check(self.lib.mpcq_set_params(self.h, _lib.d(mu)))

    def get_state(self):
        B, N, nb = self.B, self.N, self.nb
        s = dict(X=np.zeros((B, N + 1, NX)), U=np.zeros((B, N, NU)), mu=np.zeros((B, 3, nb)),
                 C=np.zeros((B, 3, nb, nb)), x_pred_prev=np.zeros((B, NX)),
                 has_prev=np.zeros(B, np.int32), idx=np.zeros(B, np.int32))
        self._check(self.lib.mpcq_get_state(self.h, _lib.d(s["X"]), _lib.d(s["U"]), _lib.d(s["mu"]), _lib.d(s["C"]),
                                            _lib.d(s["x_pred_prev"]), _lib.i(s["has_prev"]), _lib.i(s["idx"])))
        return s

    def set_state(self, X=None, U=None, mu=None, C=None, x_pred_prev=None, has_prev=None, idx=None):
        g = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        X, U, mu, C, xp = (self._f(a) for a in (X, U, mu, C, x_pred_prev))
        hp, ix = g(has_prev), g(idx)
        self._check(self.lib.mpcq_set_state(self.h, _lib.d(X), _lib.d(U), _lib.d(mu), _lib.d(C), _lib.d(xp),
                                            _lib.i(hp), _lib.i(ix)))

    # ---- explicit path (acados-style set / solve / get)
    def solve(self, x0):
        x0 = self._f(x0, (self.B, NX))
        self._check(self.lib.mpcq_solve(self.h, _lib.d(x0)))

    def get_x(self, stage):
        out = np.zeros((self.B, NX))
        self._check(self.lib.mpcq_get_x(self.h, stage, _lib.d(out)))
        return out

    def get_u(self, stage):
        out = np.zeros((self.B, NU))
        self._check(self.lib.mpcq_get_u(self.h, stage, _lib.d(out)))
        return out

    def get_cost(self):
        out = np.zeros(self.B)
        self._check(self.lib.mpcq_get_cost(self.h, _lib.d(out)))
        return out

    def get_status(self):
        out = np.zeros(self.B, np.int32)
        self._check(self.lib.mpcq_get_status(self.h, _lib.i(out)))
        return out

    def get_qp_iter(self):
        out = np.zeros(self.B, np.int32)
        self._check(self.lib.mpcq_get_qp_iter(self.h, _lib.i(out)))
        return out

    def get_qp_work(self):
        """(factorisations, vector sweeps) the last solve of every quadrotor executed."""
        out = np.zeros(self.B, np.int32)
        self._check(self.lib.mpcq_get_qp_work(self.h, _lib.i(out)))
        return out & 0x7FFF, (out >> 16) & 0x7FF

    def get_qp_float_iterations(self):
        """fp64 instances: how many interior-point iterations of the last solve ran in float (bits 27..31 of mpcq_get_qp_work)."""
        out = np.zeros(self.B, np.int32)
        self._check(self.lib.mpcq_get_qp_work(self.h, _lib.i(out)))
        return (out.view(np.uint32) >> 27).astype(np.int32)

    def get_qp_float_breakdown(self):
        """fp64 instances: True where the float interior point of the last (fallback) solve broke down and the double one ran instead."""
        out = np.zeros(self.B, np.int32)
        self._check(self.lib.mpcq_get_qp_work(self.h, _lib.i(out)))
        return (out & 0x8000) != 0

    def get_block_order(self):
        """Launch order of the last lockstep period: workgroup p ran quadrotor out[p] (identity when unused)."""
        out = np.zeros(self.B, np.int32)
        self._check(self.lib.mpcq_get_block_order(self.h, _lib.i(out)))
        return out

    def get_groups(self):
        """Groups mpcq_sim_steps runs the batch in (mpcq_tuning.groups resolved; 1 = one launch per period over the whole batch)."""
        out = ctypes.c_int32()
        self._check(self.lib.mpcq_get_groups(self.h, ctypes.byref(out)))
        return out.value

    def get_time(self):
        t = ctypes.c_double()
        self._check(self.lib.mpcq_get_stats(self.h, ctypes.byref(t)))
        return t.value

    def predict_nominal(self, x, u, dt):
        x, u = self._f(x, (self.B, NX)), self._f(u, (self.B, NU))
        out = np.zeros((self.B, NX))
        self._check(self.lib.mpcq_predict_nominal(self.h, _lib.d(x), _lib.d(u), float(dt), _lib.d(out)))
        return out

    def rgp_regress(self, v_body, a_drag):
        vb, ad = self._f(v_body, (self.B, 3)), self._f(a_drag, (self.B, 3))
        self._check(self.lib.mpcq_rgp_regress(self.h, _lib.d(vb), _lib.d(ad)))

    def get_rgp(self):
        mu = np.zeros((self.B, 3, self.nb))
        C = np.zeros((self.B, 3, self.nb, self.nb))
        self._check(self.lib.mpcq_get_rgp(self.h, _lib.d(mu), _lib.d(C)))
        return mu, C

    # ---- fused path
    def step(self, x_meas):
        x = self._f(x_meas, (self.B, NX))
        w = np.zeros((self.B, NU))
        xp = np.zeros((self.B, NX))
        self._check(self.lib.mpcq_step(self.h, _lib.d(x), _lib.d(w), _lib.d(xp)))
        return w, xp

    def sim_reset(self, x0):
        x0 = self._f(x0, (self.B, NX))
        self._check(self.lib.mpcq_sim_reset(self.h, _lib.d(x0)))

    def sim_steps(self, K, n_sub, sim_dt=5e-3):
        self._check(self.lib.mpcq_sim_steps(self.h, int(K), int(n_sub), float(sim_dt)))

    def sim_run(self, K, n_sub, sim_dt=5e-3):
        """K closed-loop periods in one launch, every instance advancing on its own (same results as sim_steps)."""
        self._check(self.lib.mpcq_sim_run(self.h, int(K), int(n_sub), float(sim_dt)))

    def sim_get_state(self):
        x = np.zeros((self.B, NX))
        w = np.zeros((self.B, NU))
        self._check(self.lib.mpcq_sim_get_state(self.h, _lib.d(x), _lib.d(w)))
        return x, w

    # ---- outputs of the loop body beyond w (a4, a9)
    def get_command(self):
        """(rotor_thrusts [B,4], collective_thrust [B], bodyrates [B,3]) of publish_control_gazebo."""
        rotor, coll, rates = np.zeros((self.B, NU)), np.zeros(self.B), np.zeros((self.B, 3))
        self._check(self.lib.mpcq_get_command(self.h, _lib.d(rotor), _lib.d(coll), _lib.d(rates)))
        return rotor, coll, rates

    def get_finished(self):
        out = np.zeros(self.B, np.int32)
        self._check(self.lib.mpcq_get_finished(self.h, _lib.i(out)))
        return out

    def get_reference_chunk(self):
        out = np.zeros((self.B, self.N, NX))
        self._check(self.lib.mpcq_get_reference_chunk(self.h, _lib.d(out)))
        return out

    def plant_substeps(self, control_dt, sim_dt=5e-3):
        return int(self.lib.mpcq_plant_substeps(float(control_dt), float(sim_dt)))

    def sim_plant_period(self, w, control_dt, sim_dt=5e-3):
        """Advance the plant state by the reference's float-accumulated substep loop; returns the substep count."""
        w = self._f(w, (self.B, NU))
        n = ctypes.c_int32()
        self._check(self.lib.mpcq_sim_plant_period(self.h, _lib.d(w), float(control_dt), float(sim_dt), ctypes.byref(n)))
        return n.value

    def sim_control_periods(self, K, control_dt, sim_dt=5e-3):
        n = ctypes.c_int32()
        self._check(self.lib.mpcq_sim_control_periods(self.h, int(K), float(control_dt), float(sim_dt), ctypes.byref(n)))
        return n.value

    def step_device_async(self, d_x_meas: int, d_w_out: int = 0):
        """Fused step on float64 device buffers (raw device addresses); asynchronous on the engine's stream."""
        self._check(self.lib.mpcq_step_device_async(self.h, ctypes.c_void_p(d_x_meas), ctypes.c_void_p(d_w_out or None)))

    def synchronize(self):
        self._check(self.lib.mpcq_synchronize(self.h))

    def get_solver_state(self):
        s = dict(qp_iter=np.zeros(self.B, np.int32), stats=np.zeros((self.B, 4)), finished=np.zeros(self.B, np.int32))
        self._check(self.lib.mpcq_get_solver_state(self.h, _lib.i(s["qp_iter"]), _lib.d(s["stats"]), _lib.i(s["finished"])))
        return s

    def set_solver_state(self, qp_iter=None, stats=None, finished=None):
        g = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        q, f, st = g(qp_iter), g(finished), self._f(stats)
        self._check(self.lib.mpcq_set_solver_state(self.h, _lib.i(q), _lib.d(st), _lib.i(f)))

    def get_kernel_time_minmax(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        self._check(self.lib.mpcq_get_kernel_time_minmax(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def get_kernel_time(self):
        t = ctypes.c_double()
        n = ctypes.c_int32()
        self._check(self.lib.mpcq_get_kernel_time(self.h, ctypes.byref(t), ctypes.byref(n)))
        return t.value, n.value

    def get_tracking_stats(self):
        out = np.zeros(5)
        self._check(self.lib.mpcq_get_tracking_stats(self.h, _lib.d(out)))
        return out

    # ---- multi-GPU statistics
    def comm_unique_id(self) -> bytes:
        buf = ctypes.create_string_buffer(128)
        self._check(self.lib.mpcq_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, nranks, uid: bytes):
        buf = ctypes.create_string_buffer(uid, 128)
        self._check(self.lib.mpcq_comm_init(self.h, rank, nranks, buf))

    def comm_share(self, owner: "Engine"):
        """Reduce over the communicator another engine of this process initialised (one communicator per rank)."""
        self._check(self.lib.mpcq_comm_share(self.h, owner.h))

    def allreduce_tracking_stats(self):
        out = np.zeros(5)
        self._check(self.lib.mpcq_allreduce_tracking_stats(self.h, _lib.d(out)))
        return out


class Learner:
    """batch x 3 recursive GPs with hyper-parameter learning on the device (RGP.learn, src/gp/RGP.py:332-505);
    the offline estimator of the reference, not part of the control step."""

    def __init__(self, batch, basis, theta, device=0, lib_path=None):
        self.lib = _lib.load(lib_path)
        self.basis = np.ascontiguousarray(basis, dtype=np.float64).reshape(3, -1)
        th = np.asarray(theta, dtype=np.float64)
        self.theta = np.ascontiguousarray(np.tile(th, (3, 1)) if th.shape == (3,) else th.reshape(3, 3))
        self.B, self.nb = int(batch), self.basis.shape[1]
        h = ctypes.c_void_p()
        self._check(self.lib.mpcq_learn_create(self.B, self.nb, _lib.d(self.basis), _lib.d(self.theta), int(device), ctypes.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise _lib.MpcqError(f"mpcq error {rc}: {self.lib.mpcq_learn_last_error().decode()}")

    def step(self, v_body, a_drag):
        s = np.ascontiguousarray(v_body, dtype=np.float64).reshape(self.B, 3)
        y = np.ascontiguousarray(a_drag, dtype=np.float64).reshape(self.B, 3)
        self._check(self.lib.mpcq_learn_step(self.h, _lib.d(s), _lib.d(y)))

    def get(self):
        B, n = self.B, self.nb
        out = dict(mu_g=np.zeros((B, 3, n)), C_g=np.zeros((B, 3, n, n)), mu_eta=np.zeros((B, 3, 3)), C_eta=np.zeros((B, 3, 3, 3)),
                   K_x_inv=np.zeros((B, 3, n, n)))
        self._check(self.lib.mpcq_learn_get(self.h, _lib.d(out["mu_g"]), _lib.d(out["C_g"]), _lib.d(out["mu_eta"]), _lib.d(out["C_eta"]),
                                            _lib.d(out["K_x_inv"])))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.mpcq_learn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
