#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference checkout.

Runs ONLY in the dev container (needs /root/reference).  Nothing here travels to the GPU
box except the .npz files it writes, which hold data only:

* log_*.npz        — windows of the reference's own logged runs (outputs/*/data/*.pkl, schema
                     src/mpc_controller_node.py:354-357 / src/execute_trajectory.py:270-273):
                     x_odom, x_ref, w_odom, cost_solution, x_pred_odom (+ rgp_* / v_body / a_drag).
                     These are outputs of the real acados+HPIPM+numpy reference.
* rgp_vectors.npz  — input/output vectors produced by importing the reference's src/gp/RGP.py
                     (casadi stubbed: only its numpy path is executed) on seeded random streams.
* learn_vectors.npz— RGP.learn (hyper-parameter UKF) streams, gp_vectors.npz — static GP posterior means (src/gp/GP.py), both
                     produced the same way.
* utils_vectors.npz— outputs of the reference's get_reference_chunk / compute_a_drag
                     (src/utils/utils.py:897-950; the module itself cannot be imported because of
                     dead imports at :22,:29,:30, so the needed function definitions are
                     extracted with ast and executed at generation time).

Usage: python tests/golden/make_golden.py [/root/reference]
"""
import ast
import os
import pickle
import sys
import types
import warnings

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")


def load_log(rel):
    with open(os.path.join(REF, rel), "rb") as f:
        return pickle.load(f)


def theta_of(d):
    return np.array([[float(np.ravel(t)[0]) for t in ax] for ax in d["rgp_theta"][0]])


def save_log(name, rel, K, meta, c_every=1):
    d = load_log(rel)
    out = dict(meta)
    for k in ("x_odom", "x_ref", "w_odom", "cost_solution", "x_pred_odom"):
        out[k] = np.asarray(d[k], dtype=np.float64)[:K + 1] if k != "x_ref" else np.asarray(d[k], dtype=np.float64)
    if meta.get("nb", 0):
        out["basis"] = np.asarray(d["rgp_basis_vectors"][0], dtype=np.float64)
        out["theta"] = theta_of(d)
        out["rgp_mu"] = np.asarray(d["rgp_mu_g_t"], dtype=np.float64)[:K]
        C = np.asarray(d["rgp_C_g_t"], dtype=np.float64)[:K]
        out["rgp_C_steps"] = np.arange(0, K, c_every)
        out["rgp_C"] = C[::c_every]
        out["v_body"] = np.asarray(d["v_body"], dtype=np.float64)[:K, :, 0]
        out["a_drag"] = np.asarray(d["a_drag"], dtype=np.float64)[:K, :, 0]
    out["source"] = rel
    np.savez_compressed(os.path.join(OUT, name), **out)
    print("wrote", name, {k: np.shape(v) for k, v in out.items() if hasattr(v, "shape")})


def make_logs():
    P = "outputs/python_simulation/data/"
    # python-sim: legacy constants, N=10, T=1, skip=1, dt_pred=optimization_dt=0.1
    save_log("log_traj1_v10_a10_gp0.npz", P + "traj1_v10_a10_gp0.pkl", 780, dict(N=10, nb=0, K=780, quad="legacy"))
    save_log("log_traj0_v10_a10_gp2.npz", P + "traj0_v10_a10_gp2.pkl", 110, dict(N=10, nb=10, K=110, quad="legacy"), c_every=5)
    save_log("log_traj0_v15_a5_gp2.npz", P + "traj0_v15_a5_gp2.pkl", 150, dict(N=10, nb=10, K=150, quad="legacy"), c_every=10)
    save_log("log_traj1_v15_a5_gp2.npz", P + "traj1_v15_a5_gp2.pkl", 45, dict(N=10, nb=10, K=45, quad="legacy"), c_every=5)
    save_log("log_trajectory_v15_a5_gp2.npz", P + "trajectory_v15_a5_gp2.pkl", 80, dict(N=10, nb=20, K=80, quad="legacy"), c_every=10)
    save_log("log_traj2_v10_a10_gp2.npz", P + "traj2_v10_a10_gp2.pkl", 100, dict(N=10, nb=10, K=100, quad="legacy"), c_every=10)
    # gazebo: hummingbird, N=5, T=1 -> skip 20, dt_pred=0.01; stale-trajectory junction at step 110
    d = load_log("outputs/gazebo_simulation/data/traj0_v12_a12_gp0.pkl")
    K = 400
    out = dict(N=5, nb=0, K=K, quad="hummingbird", junction=110, source="outputs/gazebo_simulation/data/traj0_v12_a12_gp0.pkl")
    for k in ("x_odom", "w_odom", "cost_solution", "x_pred_odom"):
        out[k] = np.asarray(d[k], dtype=np.float64)[:K + 1]
    out["x_ref"] = np.asarray(d["x_ref"], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "log_gazebo_traj0_v12_a12_gp0.npz"), **out)
    print("wrote log_gazebo_traj0_v12_a12_gp0.npz")


def import_reference_rgp():
    cs = types.ModuleType("casadi")

    class MX:  # only isinstance() checks reach it on the numpy path
        pass

    cs.MX = MX
    cs.Function = type("Function", (), {})   # only named in annotations of src/gp/GP.py
    sys.modules["casadi"] = cs
    if not hasattr(np, "NaN"):
        np.NaN = np.nan  # src/gp/RGP.py:92 predates numpy 2
    sys.path.insert(0, os.path.join(REF, "src", "gp"))
    import RGP as ref_rgp  # noqa
    return ref_rgp


def make_rgp_vectors():
    ref = import_reference_rgp()
    rng = np.random.default_rng(20261002)
    out = {}
    cases = [(10, [1.0, 0.1, 0.1], 12.0), (10, [3.0, 0.1, 0.01], 10.0), (20, [1.0, 1.0, 0.1], 10.0),
             (50, [1.0, 0.1, 0.1], 12.0), (20, [1.0, 0.1, 0.1], 12.0), (10, [3.0, 0.5, 0.01], 15.0)]
    for ci, (nb, theta, vmax) in enumerate(cases):
        X = np.linspace(-vmax, vmax, nb)
        g = ref.RGP(X, np.zeros(nb), theta=list(theta))
        K = 25
        s = rng.uniform(-1.2 * vmax, 1.2 * vmax, K)
        y = rng.normal(0, 2.0, K)
        mus, Cs = [], []
        for k in range(K):
            mu, C = g.regress(np.array([s[k]]), np.array([y[k]]))
            mus.append(np.array(mu, dtype=np.float64).copy())
            Cs.append(np.array(C, dtype=np.float64).copy())
        # predict_using_y at a few points (the numpy twin of what the OCP model evaluates)
        sp = rng.uniform(-vmax, vmax, 8)
        pred = np.array([g.predict_using_y(np.array([v]), mus[-1])[0] for v in sp])
        p = f"c{ci}_"
        out[p + "nb"] = nb; out[p + "theta"] = np.array(theta); out[p + "X"] = X
        out[p + "K_x"] = np.array(g.K_x); out[p + "K_x_inv"] = np.array(g.K_x_inv)
        out[p + "s"] = s; out[p + "y"] = y
        out[p + "mu"] = np.array(mus); out[p + "C_last"] = Cs[-1]; out[p + "C_first"] = Cs[0]
        out[p + "pred_s"] = sp; out[p + "pred_m"] = pred
    out["ncases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "rgp_vectors.npz"), **out)
    print("wrote rgp_vectors.npz")


def make_learn_vectors():
    """RGP.learn (src/gp/RGP.py:332-505) on seeded streams: per step the joint mean / covariance it returns, the
    hyper-parameter estimate and the rebuilt K_x^-1."""
    ref = import_reference_rgp()
    rng = np.random.default_rng(20261003)
    out = {}
    cases = [(10, [1.0, 0.1, 0.1], 12.0, 30), (20, [1.0, 1.0, 0.1], 10.0, 20), (10, [3.0, 0.5, 0.05], 15.0, 30)]
    for ci, (nb, theta, vmax, K) in enumerate(cases):
        X = np.linspace(-vmax, vmax, nb)
        g = ref.RGP(X, np.zeros(nb), theta=list(theta))
        s = rng.uniform(-vmax, vmax, K)
        y = 0.3 * s + 0.02 * s * np.abs(s) + rng.normal(0, 0.1, K)
        mu_z, C_z, kxi = [], [], []
        for k in range(K):
            m, C = g.learn(np.array([s[k]]), np.array([y[k]]))
            mu_z.append(np.real(np.array(m, dtype=np.complex128)).copy())
            C_z.append(np.real(np.array(C, dtype=np.complex128)).copy())
            kxi.append(np.array(g.K_x_inv, dtype=np.float64).copy())
        p = f"c{ci}_"
        out[p + "nb"] = nb; out[p + "theta"] = np.array(theta); out[p + "X"] = X
        out[p + "s"] = s; out[p + "y"] = y
        out[p + "mu_z"] = np.array(mu_z); out[p + "C_z"] = np.array(C_z)[[0, K // 2, K - 1]]; out[p + "C_z_steps"] = np.array([0, K // 2, K - 1])
        out[p + "K_x_inv_last"] = kxi[-1]
    out["ncases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "learn_vectors.npz"), **out)
    print("wrote learn_vectors.npz")


def make_gp_vectors():
    """Static GP of the use_gp = 1 model path (src/gp/GP.py:76-175): posterior mean at a few points for fixed training data."""
    import_reference_rgp()          # installs the casadi stub
    sys.path.insert(0, os.path.join(REF, "src", "gp"))
    import GP as ref_gp  # noqa
    rng = np.random.default_rng(5)
    out = {}
    cases = [(12, [1.0, 1.0, 0.1]), (20, [2.0, 0.5, 0.01]), (8, [0.7, 2.0, 0.3])]
    for ci, (n, theta) in enumerate(cases):
        X = np.sort(rng.uniform(-10, 10, n))
        y = 0.3 * X + 0.02 * X * np.abs(X) + rng.normal(0, 0.05, n)
        g = ref_gp.GP(X, y, theta=list(theta))
        xs = rng.uniform(-11, 11, 16)
        mean = np.array([np.ravel(g.predict(np.array([v])))[0] for v in xs], dtype=np.float64)
        p = f"c{ci}_"
        out[p + "X"] = X; out[p + "y"] = y; out[p + "theta"] = np.array(theta); out[p + "xs"] = xs; out[p + "mean"] = mean
    out["ncases"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "gp_vectors.npz"), **out)
    print("wrote gp_vectors.npz")


def extract_utils_functions(names):
    src = open(os.path.join(REF, "src", "utils", "utils.py")).read()
    tree = ast.parse(src)
    ns = {"np": np, "cs": types.SimpleNamespace(MX=type("MX", (), {}))}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), "utils_extract", "exec"), ns)
    return ns


def make_utils_vectors():
    ns = extract_utils_functions({"get_reference_chunk", "compute_a_drag", "v_dot_q", "q_to_rot_mat", "quaternion_inverse"})
    rng = np.random.default_rng(7)
    out = {}
    # reference chunks: every (len, idx, N, skip) corner
    cases = []
    for T in (1, 2, 7, 23, 100, 101, 250):
        traj = rng.normal(size=(T, 13))
        for N in (5, 10, 20, 50):
            for skip in (1, 2, 3, 5, 10, 20):
                for idx in sorted(set([0, 1, max(0, T - N * skip - 1), max(0, T - N * skip), max(0, T - N * skip + 1),
                                       max(0, T - skip - 1), max(0, T - skip), T - 1, T, T + 3, T // 2])):
                    cases.append((T, N, skip, idx))
    trajs = {}
    chunk_cases, chunk_out = [], []
    for (T, N, skip, idx) in cases:
        if T not in trajs:
            trajs[T] = rng.normal(size=(T, 13))
        ch = ns["get_reference_chunk"](trajs[T], idx, N, skip)
        assert ch.shape == (N, 13), (T, N, skip, idx, ch.shape)
        # store as row indices into the trajectory (bit-exact bookkeeping)
        rows = np.array([int(np.where((trajs[T] == r).all(axis=1))[0][0]) for r in ch])
        chunk_cases.append((T, N, skip, idx))
        chunk_out.append(np.pad(rows, (0, 50 - N), constant_values=-1))
    out["chunk_cases"] = np.array(chunk_cases)
    out["chunk_rows"] = np.array(chunk_out)
    # compute_a_drag
    xs = rng.normal(size=(40, 13)); xp = xs + 0.05 * rng.normal(size=(40, 13))
    xs[:, 3:7] /= np.linalg.norm(xs[:, 3:7], axis=1, keepdims=True) * rng.uniform(0.95, 1.05, (40, 1))
    vb, ad = [], []
    for i in range(40):
        dt = (0.01, 0.05, 0.1)[i % 3]
        v, a = ns["compute_a_drag"](xs[i], xp[i], dt)
        vb.append(np.concatenate(v)); ad.append(np.concatenate(a))
    out["drag_x"] = xs; out["drag_xp"] = xp; out["drag_vb"] = np.array(vb); out["drag_ad"] = np.array(ad)
    np.savez_compressed(os.path.join(OUT, "utils_vectors.npz"), **out)
    print("wrote utils_vectors.npz", len(chunk_cases), "chunk cases")


def make_circle_vectors():
    """Outputs of the reference's closed-form circle generators (TrajectoryGenerator.py:41-131) read back through
    its own load_trajectory (:223-244)."""
    import tempfile
    sys.path.insert(0, os.path.join(REF, "src", "trajectory_generation"))
    import TrajectoryGenerator as TG
    g = TG.TrajectoryGenerator.__new__(TG.TrajectoryGenerator)
    g.sampled_trajectory_filename = os.path.join(tempfile.mkdtemp(), "t.csv")
    out = {}
    g.sample_circle_trajectory_accelerating(10, 12, 30, 0.1, start_point=np.array([0.0, 0.0, 3.0]))
    out["acc_x"], out["acc_t"] = g.load_trajectory()
    g.sample_circle_trajectory(5.0, 8.0, 0.05, start_point=np.array([1.0, -2.0, 3.0]))
    out["const_x"], out["const_t"] = g.load_trajectory()
    g.sample_circle_trajectory_acc_dec(10, 10, 0.01)
    out["ad_x"], out["ad_t"] = g.load_trajectory()
    np.savez_compressed(os.path.join(OUT, "circle_vectors.npz"), **out)
    print("wrote circle_vectors.npz", {k: v.shape for k, v in out.items()})


def make_poly_vectors():
    """Sampled references of random piecewise polynomials through the reference's own evaluator
    (uav_trajectory.Trajectory.loadcsv / eval) and TrajectoryGenerator.save_evals_csv / load_trajectory."""
    import tempfile
    sys.path.insert(0, os.path.join(REF, "src", "trajectory_generation"))
    import TrajectoryGenerator as TG
    import uav_trajectory
    rng = np.random.default_rng(11)
    d = tempfile.mkdtemp()
    g = TG.TrajectoryGenerator.__new__(TG.TrajectoryGenerator)
    g.sampled_trajectory_filename = os.path.join(d, "s.csv")
    out = {}
    for case, (npieces, dt) in enumerate([(3, 0.01), (5, 0.05), (1, 0.02)]):
        pieces = np.zeros((npieces, 33))
        pieces[:, 0] = np.round(rng.uniform(0.4, 1.7, npieces), 6)
        pieces[:, 1:] = np.round(rng.normal(0, 1.0, (npieces, 32)) * (0.5 ** np.tile(np.arange(8), 4)), 6)
        f = os.path.join(d, f"p{case}.csv")
        np.savetxt(f, pieces, fmt="%.6f", delimiter=",", header="duration," + ",".join(f"c{i}" for i in range(32)))
        tr = uav_trajectory.Trajectory()
        tr.loadcsv(f)
        g.save_evals_csv(tr, g.sampled_trajectory_filename, dt=dt)
        x, t = g.load_trajectory()
        out[f"p{case}_pieces"] = pieces; out[f"p{case}_dt"] = dt; out[f"p{case}_x"] = x; out[f"p{case}_t"] = t
    out["ncases"] = 3
    np.savez_compressed(os.path.join(OUT, "poly_vectors.npz"), **out)
    print("wrote poly_vectors.npz", {k: np.shape(v) for k, v in out.items()})


def make_tumbling_log():
    """Round 6: the WHOLE flight of traj2_v10_a10_gp2 (299 control periods).  The reference's own loop loses the quadrotor behind step ~100
    (cost_solution 0.7 -> 1.5e5, |q| far from 1 along the predictions): the window the mixed-precision mode's validity limit is tested on
    (tests/test_gpu_parity.py).  Measurements, controls and costs only; the RGP posterior of this run is in log_traj2_v10_a10_gp2.npz."""
    d = load_log("outputs/python_simulation/data/traj2_v10_a10_gp2.pkl")
    K = len(d["w_odom"])
    out = dict(N=10, nb=10, K=K, quad="legacy", source="outputs/python_simulation/data/traj2_v10_a10_gp2.pkl")
    for k in ("x_odom", "x_ref", "w_odom", "cost_solution"):
        out[k] = np.asarray(d[k], dtype=np.float64)
    out["basis"] = np.asarray(d["rgp_basis_vectors"][0], dtype=np.float64)
    out["theta"] = theta_of(d)
    np.savez_compressed(os.path.join(OUT, "log_traj2_v10_a10_gp2_whole.npz"), **out)
    print("wrote log_traj2_v10_a10_gp2_whole.npz", {k: np.shape(v) for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "tumbling":      # (only this fixture; the others are unchanged since round 1)
        make_tumbling_log()
        sys.exit(0)
    make_poly_vectors()
    make_circle_vectors()
    make_logs()
    make_rgp_vectors()
    make_learn_vectors()
    make_gp_vectors()
    make_utils_vectors()
    make_tumbling_log()
