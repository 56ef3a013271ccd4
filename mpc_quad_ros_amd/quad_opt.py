"""Batched drop-in for the reference's controller class ``quad_optimizer`` (src/quad_opt.py:35).

Same method names, argument meaning and return shapes, with a leading batch axis B; the arithmetic
runs in libmpcq.so on the MI355X.  Error behaviour mirrors the reference: ``run_optimization(None)``
raises ValueError (src/quad_opt.py:325-326); shape mismatches raise AssertionError like the
reference's asserts (src/quad_opt.py:360-361,388-391)."""
from __future__ import annotations

import numpy as np

from .engine import Engine
from .params import NU, NX, EngineConfig, QuadParams, hummingbird


class _RGPView:
    """What callers read from ``quad_opt.gpe.gp[d]`` (src/mpc_controller_node.py:304-318)."""

    def __init__(self, X, theta):
        self.X = X
        self._theta = list(theta)

    def get_theta(self):
        return list(self._theta)


class _GPEView:
    type = "RGP"

    def __init__(self, basis, theta):
        self.gp = [_RGPView(basis[d], theta[d]) for d in range(3)]

    def get_theta(self):
        return [g.get_theta() for g in self.gp]


class quad_optimizer:
    def __init__(self, quad: QuadParams | None = None, t_horizon=1, n_nodes=100, gpe=None, batch=1,
                 dt_pred=0.01, device=0, precision=0, lib_path=None):
        """gpe: None (nominal model) or a dict(basis=[3,nb], theta=[3,3] or [3]) describing the RGP
        ensemble (GPEnsemble.fromrange / fromemptybasisvectors, src/gp/GPE.py:110-150)."""
        self.quad = quad or hummingbird()
        self.n_nodes = n_nodes
        self.t_horizon = t_horizon
        self.optimization_dt = self.t_horizon / self.n_nodes
        self.nx, self.nu, self.ny = NX, NU, NX + NU
        nb, basis, theta = 0, None, None
        if gpe is not None:
            basis = np.asarray(gpe["basis"], dtype=np.float64)
            nb = basis.shape[1]
            theta = gpe.get("theta")
        self.cfg = EngineConfig(batch=batch, N=n_nodes, T=float(t_horizon), quad=self.quad, nb=nb, basis=basis,
                                theta=theta, dt_pred=dt_pred, device=device, precision=precision)
        self.gpe = _GPEView(self.cfg.basis, self.cfg.theta) if nb else None
        self.np = 3 * nb
        self.batch = batch
        self.engine = Engine(self.cfg, lib_path=lib_path)
        self.yref = None
        self.yref_N = None

    # -- src/quad_opt.py:271-292
    def set_reference_state(self, x_target=None, u_target=None):
        if u_target is None:
            u_target = np.ones((self.nu,)) * 0.16
        if x_target is None:
            x_target = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0], dtype=float)
        x_target = np.broadcast_to(np.asarray(x_target, dtype=float), (self.batch, NX))
        x_traj = np.repeat(x_target[:, None, :], self.n_nodes, axis=1)
        u_traj = np.broadcast_to(np.asarray(u_target, dtype=float), (self.batch, self.n_nodes, NU))
        return self.set_reference_trajectory(x_traj, u_traj)

    # -- src/quad_opt.py:295-317
    def set_reference_trajectory(self, x_trajectory, u_trajectory=None):
        x_trajectory = np.asarray(x_trajectory, dtype=np.float64).reshape(self.batch, self.n_nodes, NX)
        if u_trajectory is None:
            u_trajectory = np.ones((self.batch, self.n_nodes, NU)) * 0.16   # hover
        u_trajectory = np.asarray(u_trajectory, dtype=np.float64).reshape(self.batch, self.n_nodes, NU)
        self.yref = np.concatenate((x_trajectory, u_trajectory), axis=2)
        self.yref_N = x_trajectory[:, -1, :].copy()                        # the LAST chunk row, not a new one
        self.engine.set_reference(self.yref, self.yref_N)
        return self.yref, self.yref_N

    # -- src/quad_opt.py:321-350
    def run_optimization(self, x_init):
        if x_init is None:
            raise ValueError("x_init has to be set before running the optimization")
        x_init = np.asarray(x_init, dtype=np.float64).reshape(self.batch, NX)
        self.engine.solve(x_init)
        st = self.engine.get_state()
        return st["X"], st["U"], self.engine.get_time(), self.engine.get_cost()

    # -- src/quad_opt.py:353-377 (nominal model only: the node calls it on quad_nominal)
    def discrete_dynamics(self, x, u, dt, body_frame=False):
        x = np.asarray(x, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        assert x.shape == (self.batch, self.nx), f"x has to be of shape ({self.batch}, {self.nx})"
        assert u.shape == (self.batch, self.nu), f"u has to be of shape ({self.batch}, {self.nu})"
        x_out = self.engine.predict_nominal(x, u, dt)
        if body_frame:
            from .host_math import v_dot_q_inv
            x_out[:, 7:10] = v_dot_q_inv(x_out[:, 7:10], x_out[:, 3:7])
        return x_out

    # -- src/quad_opt.py:380-406
    def regress_and_update_RGP_model(self, v_body, a_drag):
        assert len(v_body) == 3, "v_body has to be a list of length 3"
        assert len(a_drag) == 3, "a_drag has to be a list of length 3"
        assert self.gpe is not None, "RGP model has to be initialized before calling this method"
        vb = np.stack([np.asarray(v, dtype=np.float64).reshape(self.batch) for v in v_body], axis=1)
        ad = np.stack([np.asarray(a, dtype=np.float64).reshape(self.batch) for a in a_drag], axis=1)
        self.engine.rgp_regress(vb, ad)      # new means become the stage parameters on the device
        mu, C = self.engine.get_rgp()
        return [mu[:, d] for d in range(3)], [C[:, d] for d in range(3)]
