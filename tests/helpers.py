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
