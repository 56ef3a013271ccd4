"""Small host-side helpers of the loop body that callers of the facade use between engine calls
(batched numpy restatements of src/utils/utils.py:317-340,434-440,897-950)."""
from __future__ import annotations

import numpy as np


def q_to_rot_mat(q):
    qw, qx, qy, qz = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (qy ** 2 + qz ** 2); R[..., 0, 1] = 2 * (qx * qy - qw * qz); R[..., 0, 2] = 2 * (qx * qz + qw * qy)
    R[..., 1, 0] = 2 * (qx * qy + qw * qz); R[..., 1, 1] = 1 - 2 * (qx ** 2 + qz ** 2); R[..., 1, 2] = 2 * (qy * qz - qw * qx)
    R[..., 2, 0] = 2 * (qx * qz - qw * qy); R[..., 2, 1] = 2 * (qy * qz + qw * qx); R[..., 2, 2] = 1 - 2 * (qx ** 2 + qy ** 2)
    return R


def v_dot_q(v, q):
    return np.einsum("...ij,...j->...i", q_to_rot_mat(q), v)


def v_dot_q_inv(v, q):
    qc = q * np.array([1.0, -1.0, -1.0, -1.0])
    return v_dot_q(v, qc)


def compute_a_drag(x_now, x_pred_minus_1, dt):
    """Batched compute_a_drag (src/utils/utils.py:934-950): returns (v_body [B,3], a_drag [B,3])."""
    vb = v_dot_q_inv(x_now[..., 7:10], x_now[..., 3:7])
    vp = v_dot_q_inv(x_pred_minus_1[..., 7:10], x_pred_minus_1[..., 3:7])
    return vb, (vb - vp) / dt


def get_reference_chunk(reference_trajectory, current_idx, control_nodes, skip=1):
    """Row indices semantics of src/utils/utils.py:897-931 for one trajectory [T,13]."""
    T = reference_trajectory.shape[0]
    left = T - current_idx
    if left > control_nodes * skip:
        have = control_nodes
    elif left > skip - 1:
        have = min(control_nodes, -(-left // skip))
    else:
        have = 0
    rows = [current_idx + j * skip if j < have else T - 1 for j in range(control_nodes)]
    return reference_trajectory[rows]
