"""Reference-format run log for a swarm.

The reference appends one dict per control step with the keys of
src/mpc_controller_node.py:354-357 / src/execute_trajectory.py:270-273 and pickles a dict of stacked
arrays (src/Logger.py:37-62).  `SwarmLogger` collects the same keys with a leading batch axis and can
emit, per quadrotor, a dict with exactly the reference's layout, so the reference's own analysis code
(src/Visualiser.py:791-987: RMS and CPU-time summaries) can read runs produced by this engine."""
from __future__ import annotations

import pickle

import numpy as np

REFERENCE_KEYS = ("x_odom", "x_pred_odom", "x_ref", "t_odom", "w_odom", "t_cpu", "cost_solution",
                  "rgp_basis_vectors", "rgp_mu_g_t", "rgp_C_g_t", "rgp_theta", "v_body", "a_drag")


class SwarmLogger:
    def __init__(self, engine):
        self.engine = engine
        self.rows = {k: [] for k in REFERENCE_KEYS}
        self._x_pred_prev = None

    def log_step(self, t, x_meas, w, x_pred, x_ref0, with_rgp=True):
        """Call once per control step with the arrays of that step ([B, ...])."""
        from .host_math import compute_a_drag
        e = self.engine
        B = e.B
        r = self.rows
        r["x_odom"].append(np.array(x_meas)); r["x_pred_odom"].append(np.array(x_pred)); r["x_ref"].append(np.array(x_ref0))
        r["t_odom"].append(np.full(B, t)); r["w_odom"].append(np.array(w))
        r["t_cpu"].append(np.full((B, 1), e.get_time())); r["cost_solution"].append(e.get_cost())
        if e.nb and with_rgp:
            mu, C = e.get_rgp()
            xpm1 = self._x_pred_prev if self._x_pred_prev is not None else np.asarray(x_meas)
            vb, ad = compute_a_drag(np.asarray(x_meas), xpm1, e.cfg.dt_pred)
            r["rgp_basis_vectors"].append(np.broadcast_to(e.cfg.basis, (B, 3, e.nb)).copy())
            r["rgp_mu_g_t"].append(mu); r["rgp_C_g_t"].append(C)
            r["rgp_theta"].append(np.broadcast_to(e.cfg.theta, (B, 3, 3)).copy())
            r["v_body"].append(vb[:, :, None]); r["a_drag"].append(ad[:, :, None])
        else:
            for k in ("rgp_basis_vectors", "rgp_mu_g_t", "rgp_C_g_t", "rgp_theta", "v_body", "a_drag"):
                r[k].append(None)
        self._x_pred_prev = np.array(x_pred)

    def quad_log(self, b):
        """Dict for quadrotor b in the reference's pickle layout (arrays stacked over steps)."""
        out = {}
        for k, v in self.rows.items():
            if not v:
                out[k] = np.array([])
            elif v[0] is None:
                out[k] = np.array([None] * len(v), dtype=object)
            else:
                out[k] = np.stack([s[b] for s in v])
        return out

    def save(self, path, b=0):
        with open(path, "wb") as f:
            pickle.dump(self.quad_log(b), f)

    def rms_position_error(self):
        """Visualiser definition (src/Visualiser.py:787-789,809-811,918): per step sqrt(mean_xyz e^2), total RMS over steps."""
        x = np.stack(self.rows["x_odom"])[:, :, :3]
        xr = np.stack(self.rows["x_ref"])[:, :, :3]
        rms_k = np.sqrt(np.mean((x - xr) ** 2, axis=2))
        return np.sqrt(np.mean(rms_k ** 2, axis=0))

    def cpu_time_summary(self):
        """The CPU-time line of Visualiser.plot_data (src/Visualiser.py:981-987): avg_cpu = np.mean(t_cpu), std_cpu = np.std(t_cpu)
        over the logged steps, in seconds.  Here t_cpu is the device time of the whole-batch launch of each step (the engine's
        analogue of acados' 'time_tot'), identical for every quadrotor of the batch; returns (avg, std, per-quadrotor-share avg),
        the last being avg / B -- the time one quadrotor's solve costs the device."""
        t = np.stack(self.rows["t_cpu"])[:, 0, 0]
        avg, std = float(np.mean(t)), float(np.std(t))
        return avg, std, avg / self.engine.B

    def summary_title(self):
        """'MPC CPU Time, Avg: ..ms, STD: ..' as the reference's plot title prints it (src/Visualiser.py:987)."""
        avg, std, _ = self.cpu_time_summary()
        return f"MPC CPU Time, Avg: {avg * 1e3:.2f}ms, STD: {std * 1e3:.2f}"
