"""Shared test helpers: fixture loading and engine configs (tests only)."""
import os

import numpy as np

from mpc_quad_ros_amd.params import EngineConfig, hummingbird, legacy_sim

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def config_for_log(g, batch=1, **kw):
    """Engine configuration that produced a logged run (see tests/golden/make_golden.py)."""
    quad = legacy_sim() if str(g["quad"]) == "legacy" else hummingbird()
    N, nb = int(g["N"]), int(g["nb"])
    if str(g["quad"]) == "legacy":      # python sim: src/execute_trajectory.py:123,202,214
        extra = dict(dt_pred=1.0 / N, skip=1)
    else:                               # gazebo node: src/mpc_controller_node.py:116,222
        extra = dict(dt_pred=0.01)
    if nb:
        extra.update(basis=g["basis"], theta=g["theta"])
    extra.update(kw)
    return EngineConfig(batch=batch, N=N, T=1.0, quad=quad, nb=nb, **extra)


def random_states(rng, B, scale=1.0):
    """Plausible quadrotor states: position around hover, near-unit quaternion, moderate v, r."""
    x = np.zeros((B, 13))
    x[:, 0:3] = rng.normal(0, 2.0 * scale, (B, 3)) + np.array([0, 0, 3.0])
    q = rng.normal(0, 0.25 * scale, (B, 4)) + np.array([1.0, 0, 0, 0])
    x[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    x[:, 7:10] = rng.normal(0, 2.0 * scale, (B, 3))
    x[:, 10:13] = rng.normal(0, 0.5 * scale, (B, 3))
    return x


def visualiser_summaries(log):
    """TEST HELPER: the numbers Visualiser.plot_data prints from a run log, restated (src/Visualiser.py:787-789 `rms`, :795-811 the
    stacked arrays and the per-step position RMS, :918 the total in the title of the position panel, :981-987 the CPU-time panel).
    `log` is a pickle written in the reference's layout (loaded the way src/utils/save_dataset.py:6-9 does) or the dictionary
    itself.  Returns the per-step RMS, the total RMS [m], avg and std of t_cpu [s] and the two title strings."""
    import pickle
    if isinstance(log, (str, os.PathLike)):
        with open(log, "rb") as f:
            log = pickle.load(f)
    rms = lambda x, axis=0: np.sqrt(np.mean(x ** 2, axis=axis))                      # Visualiser.rms
    x_world = np.stack(log["x_odom"], axis=0)
    x_world_ref = np.stack(log["x_ref"], axis=0)
    t_cpu = np.stack(log["t_cpu"], axis=0)
    e_pos_ref = x_world[:, 0:3] - x_world_ref[:, 0:3]
    rms_pos_ref = rms(e_pos_ref, 1)
    total = rms(rms_pos_ref, 0)
    avg_cpu, std_cpu = np.mean(t_cpu), np.std(t_cpu)
    return {"rms_pos_ref": rms_pos_ref, "rms_total": float(total), "avg_cpu": float(avg_cpu), "std_cpu": float(std_cpu),
            "title_rms": f"RMS Position Error, Total: {total * 1e3:.2f}mm",
            "title_cpu": f"MPC CPU Time, Avg: {avg_cpu * 1e3:.2f}ms, STD: {std_cpu * 1e3:.2f}"}
