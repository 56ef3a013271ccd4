"""Pins the CPU oracle (oracle/mpcq_oracle.cpp) against the reference's own outputs:
logged acados+HPIPM+RGP runs, vectors produced by the imported src/gp/RGP.py, and vectors
produced by the reference's get_reference_chunk / compute_a_drag.  CPU only.

Tolerances: the logged w_odom come out of HPIPM stopped at its own tolerance and are replayed
free-running (iterate persisted), so agreement is ~1e-12 on the first steps and stays at the
1e-6 level on contractive windows; x_pred / RGP / a_drag / plant are
teacher-forced and agree to rounding."""
import numpy as np
import pytest

from helpers import config_for_log, load_golden
from oracle.oracle import OracleEngine, compute_a_drag, reference_chunk

LOGS_GP = [  # file, steps, w tolerance
    ("log_traj0_v10_a10_gp2.npz", 110, 5e-6),
    ("log_traj0_v15_a5_gp2.npz", 150, 5e-6),
    ("log_traj1_v15_a5_gp2.npz", 45, 5e-6),
    ("log_trajectory_v15_a5_gp2.npz", 80, 2e-4),   # aggressive flight: amplification before step 85
    ("log_traj2_v10_a10_gp2.npz", 100, 5e-6),
]


def test_replay_no_gp_cold_start():
    g = load_golden("log_traj1_v10_a10_gp0.npz")
    e = OracleEngine(config_for_log(g))
    e.set_trajectories(g["x_ref"][None])
    K = int(g["K"])
    err = np.zeros(K)
    cerr = np.zeros(K)
    for k in range(K):
        w, _ = e.step(g["x_odom"][k][None])
        err[k] = np.abs(w[0] - g["w_odom"][k]).max()
        cerr[k] = abs(e.get_cost()[0] - g["cost_solution"][k]) / max(1.0, abs(g["cost_solution"][k]))
        assert e.get_status()[0] == 0
    assert err[:10].max() < 1e-9          # cold start, before HPIPM's tolerance accumulates
    assert err[:60].max() < 1e-6
    assert err.max() < 1e-4
    assert cerr[:60].max() < 1e-7 and cerr.max() < 1e-5


@pytest.mark.parametrize("name,K,tol", LOGS_GP)
def test_replay_rgp_logs(name, K, tol):
    g = load_golden(name)
    e = OracleEngine(config_for_log(g))
    e.set_trajectories(g["x_ref"][None])
    werr, muerr, aerr, verr = [], [], [], []
    csteps = {int(s): i for i, s in enumerate(g["rgp_C_steps"])}
    for k in range(K):
        if k > 0:   # the node reads x_pred_{k-1} from the log tail: teacher-force it
            e.set_state(x_pred_prev=g["x_pred_odom"][k - 1][None])
        w, _ = e.step(g["x_odom"][k][None])
        mu, C = e.get_rgp()
        werr.append(np.abs(w[0] - g["w_odom"][k]).max())
        scale = max(1.0, np.abs(g["rgp_mu"][k]).max())
        muerr.append(np.abs(mu[0] - g["rgp_mu"][k]).max() / scale)
        if k in csteps:
            Cg = g["rgp_C"][csteps[k]]
            assert np.abs(C[0] - Cg).max() <= 1e-11 * max(1.0, np.abs(Cg).max())
        xpm1 = g["x_pred_odom"][k - 1] if k > 0 else g["x_odom"][k]
        vb, ad = compute_a_drag(g["x_odom"][k], xpm1, 0.1)
        verr.append(np.abs(vb - g["v_body"][k]).max())
        aerr.append(np.abs(ad - g["a_drag"][k]).max())
    assert max(werr) < tol, (max(werr), int(np.argmax(werr)))
    assert max(muerr) < 1e-10
    assert max(verr) < 1e-14 and max(aerr) < 1e-12    # numpy's BLAS dot order differs by an ulp


def test_replay_gazebo_hummingbird_strided_chunk_and_junction():
    g = load_golden("log_gazebo_traj0_v12_a12_gp0.npz")
    e = OracleEngine(config_for_log(g))
    assert e.cfg.skip == 20
    J = int(g["junction"])
    # steps < J run on the stale previous trajectory (chunk = its last row), then idx restarts at 0
    e.set_trajectories(np.repeat(g["x_ref"][0][None], 3, axis=0)[None])
    e.set_state(idx=np.array([5]))
    err = []
    for k in range(int(g["K"])):
        if k == J:
            e.set_trajectories(g["x_ref"][J:][None])
        w, _ = e.step(g["x_odom"][k][None])
        err.append(np.abs(w[0] - g["w_odom"][k]).max())
    err = np.array(err)
    assert err[6:].max() < 1e-7      # the first steps absorb the unlogged warm start
    assert err[20:100].max() < 1e-9


@pytest.mark.parametrize("name", ["log_traj1_v10_a10_gp0.npz", "log_traj0_v10_a10_gp2.npz", "log_gazebo_traj0_v12_a12_gp0.npz"])
def test_nominal_prediction_is_logged_x_pred(name):
    g = load_golden(name)
    cfg = config_for_log(g, batch=64)
    e = OracleEngine(cfg)
    xp = e.predict_nominal(g["x_odom"][:64], g["w_odom"][:64], cfg.dt_pred)
    assert np.abs(xp - g["x_pred_odom"][:64]).max() < 1e-14


def test_plant_reproduces_logged_odometry():
    # python sim: x_odom[k+1] = 20 RK4 substeps of 5 ms with drag from (x_odom[k], w[k])
    g = load_golden("log_traj0_v10_a10_gp2.npz")
    cfg = config_for_log(g, batch=100)
    e = OracleEngine(cfg)
    x1, n = e.plant_control_period(g["x_odom"][:100], g["w_odom"][:100], 0.1, 5e-3)
    assert n == 20
    assert np.abs(x1 - g["x_odom"][1:101]).max() < 1e-13
    for dt, nsub in ((0.05, 11), (0.02, 4), (0.1, 20)):   # float-accumulation quirk, App. B
        _, n = e.plant_control_period(g["x_odom"][:100], g["w_odom"][:100], dt, 5e-3)
        assert n == nsub


def test_rgp_against_imported_reference():
    v = load_golden("rgp_vectors.npz")
    from mpc_quad_ros_amd.params import EngineConfig
    for ci in range(int(v["ncases"])):
        p = f"c{ci}_"
        nb = int(v[p + "nb"])
        cfg = EngineConfig(batch=1, N=5, nb=nb, basis=np.tile(v[p + "X"], (3, 1)), theta=v[p + "theta"])
        e = OracleEngine(cfg)
        Kx, Kxi = e.get_kx()
        assert np.abs(Kx[0] - v[p + "K_x"]).max() < 1e-15
        assert np.abs(Kxi[0] - v[p + "K_x_inv"]).max() < 1e-9 * np.abs(v[p + "K_x_inv"]).max()
        mu0, C0 = e.get_rgp()
        assert np.all(mu0 == 0) and np.abs(C0[0, 0] - v[p + "K_x"]).max() < 1e-15   # C0 = K + sn^2 I
        s, y = v[p + "s"], v[p + "y"]
        for k in range(len(s)):
            e.rgp_regress(np.array([[s[k], 0.0, 0.0]]), np.array([[y[k], 0.0, 0.0]]))
            mu, C = e.get_rgp()
            assert np.abs(mu[0, 0] - v[p + "mu"][k]).max() < 1e-11 * max(1.0, np.abs(v[p + "mu"][k]).max())
            if k == 0:
                assert np.abs(C[0, 0] - v[p + "C_first"]).max() < 1e-13
        assert np.abs(C[0, 0] - v[p + "C_last"]).max() < 1e-12
        # the model-side mean k*(s) K_x^-1 mu (predict_using_y) through the OCP model's GP term:
        # at q = identity, v = [s,0,0], u = 0: vdot_x = m_x(s)
        e.set_params(np.concatenate([mu[0, 0], np.zeros(2 * nb)])[None])
        for sp, mp in zip(v[p + "pred_s"], v[p + "pred_m"]):
            x = np.zeros(13); x[3] = 1.0; x[7] = sp
            f, _ = e.model_f(x, np.zeros(4), np.concatenate([mu[0, 0], np.zeros(2 * nb)]))
            assert abs(f[7] - mp) < 1e-10 * max(1.0, abs(mp))


def test_reference_chunk_bit_exact():
    v = load_golden("utils_vectors.npz")
    rng = np.random.default_rng(0)
    trajs = {}
    for (T, N, skip, idx), rows in zip(v["chunk_cases"], v["chunk_rows"]):
        T, N, skip, idx = int(T), int(N), int(skip), int(idx)
        if T not in trajs:
            trajs[T] = rng.normal(size=(T, 13))
        ch = reference_chunk(trajs[T], idx, N, skip)
        assert np.array_equal(ch, trajs[T][rows[:N]]), (T, N, skip, idx)


def test_compute_a_drag():
    v = load_golden("utils_vectors.npz")
    for i in range(len(v["drag_x"])):
        dt = (0.01, 0.05, 0.1)[i % 3]
        vb, ad = compute_a_drag(v["drag_x"][i], v["drag_xp"][i], dt)
        assert np.abs(vb - v["drag_vb"][i]).max() < 1e-14     # ulp-level: BLAS dot order
        assert np.abs(ad - v["drag_ad"][i]).max() < 1e-12


def test_analytic_jacobian_and_sensitivities_vs_finite_differences():
    g = load_golden("log_traj0_v10_a10_gp2.npz")
    e = OracleEngine(config_for_log(g))
    rng = np.random.default_rng(3)
    mu = rng.normal(0, 1.0, 30)
    for _ in range(10):
        x = rng.normal(0, 1.0, 13); x[3:7] += np.array([1.0, 0, 0, 0]); u = rng.uniform(0, 1, 4)
        f, J = e.model_f(x, u, mu)
        phi, AB = e.rk4_sens(x, u, mu, 0.1)
        Jfd = np.zeros((13, 17)); ABfd = np.zeros((13, 17))
        z = np.concatenate([x, u]); h = 1e-6
        for j in range(17):
            zp, zm = z.copy(), z.copy(); zp[j] += h; zm[j] -= h
            Jfd[:, j] = (e.model_f(zp[:13], zp[13:], mu)[0] - e.model_f(zm[:13], zm[13:], mu)[0]) / (2 * h)
            ABfd[:, j] = (e.rk4_sens(zp[:13], zp[13:], mu, 0.1)[0] - e.rk4_sens(zm[:13], zm[13:], mu, 0.1)[0]) / (2 * h)
        assert np.abs(J - Jfd).max() < 1e-7 * max(1.0, np.abs(J).max())
        assert np.abs(AB - ABfd).max() < 1e-7 * max(1.0, np.abs(AB).max())


def test_qp_solution_satisfies_kkt():
    g = load_golden("log_trajectory_v15_a5_gp2.npz")   # aggressive: inputs on bounds
    e = OracleEngine(config_for_log(g))
    e.set_trajectories(g["x_ref"][None])
    nact = 0
    for k in range(60):
        w, _ = e.step(g["x_odom"][k][None])
        assert 0 <= e.get_kkt()[0] < 1e-9
        U = np.stack([e.get_u(i)[0] for i in range(10)])
        assert U.min() >= 0.0 and U.max() <= 1.0
        nact += int(((U == 0.0) | (U == 1.0)).sum())
    assert nact > 0     # the window really exercises active bounds


def test_rgp_learn_against_imported_reference():
    """RGP.learn (hyper-parameter UKF, src/gp/RGP.py:332-505): the restatement against streams produced by importing the
    reference (tests/golden/make_golden.py: make_learn_vectors): joint mean every step, joint covariance at three steps,
    the rebuilt K_x^-1 at the end."""
    from oracle.oracle import OracleLearner
    v = load_golden("learn_vectors.npz")
    for c in range(int(v["ncases"])):
        p = f"c{c}_"
        nb, X, theta = int(v[p + "nb"]), v[p + "X"], v[p + "theta"]
        lr = OracleLearner(1, np.tile(X, (3, 1)), theta)
        s, y = v[p + "s"], v[p + "y"]
        csteps = {int(k): i for i, k in enumerate(v[p + "C_z_steps"])}
        for k in range(len(s)):
            lr.step(np.full((1, 3), s[k]), np.full((1, 3), y[k]))
            g = lr.get()
            mu_z = np.concatenate([g["mu_g"][0, 1], g["mu_eta"][0, 1]])
            assert np.abs(mu_z - v[p + "mu_z"][k]).max() < 1e-9 * max(1.0, np.abs(v[p + "mu_z"][k]).max()), (c, k)
            if k in csteps:
                Cz = v[p + "C_z"][csteps[k]]
                assert np.abs(g["C_g"][0, 1] - Cz[:nb, :nb]).max() < 1e-9 * max(1.0, np.abs(Cz).max())
                assert np.abs(g["C_eta"][0, 1] - Cz[nb:, nb:]).max() < 1e-9 * max(1.0, np.abs(Cz).max())
        Ki = v[p + "K_x_inv_last"]
        assert np.abs(g["K_x_inv"][0, 2] - Ki).max() < 1e-8 * np.abs(Ki).max()
        assert np.array_equal(g["mu_g"][0, 0], g["mu_g"][0, 2])        # the three axes received the same stream


def test_static_gp_model_matches_reference_gp_predict():
    """use_gp = 1: the static GP of the model (src/gp/GP.py:76-175) is the RGP-shaped model term with basis = training
    inputs, mu = training responses and K_x = K + (noise + 1e-7) I, i.e. sigma_n = sqrt(noise + 1e-7)
    (mpc_quad_ros_amd.params.static_gp_theta).  Posterior means against vectors from the imported reference GP."""
    from mpc_quad_ros_amd.params import EngineConfig, hummingbird, static_gp_theta
    v = load_golden("gp_vectors.npz")
    for c in range(int(v["ncases"])):
        p = f"c{c}_"
        X, y, theta, xs, mean = v[p + "X"], v[p + "y"], v[p + "theta"], v[p + "xs"], v[p + "mean"]
        n = len(X)
        cfg = EngineConfig(batch=1, N=5, quad=hummingbird(), nb=n, basis=np.tile(X, (3, 1)), theta=static_gp_theta(theta))
        o = OracleEngine(cfg)
        Kx, Kxi = o.get_kx()
        alpha = Kxi[0] @ y
        L, sf = theta[0], theta[1]
        pred = np.array([np.sum(alpha * sf ** 2 * np.exp(-0.5 * (s - X) ** 2 / L ** 2)) for s in xs])
        assert np.abs(pred - mean).max() < 1e-9 * max(1.0, np.abs(mean).max())
        # and through the model itself: vdot gains R m(v_body) with identity attitude
        x = np.zeros(13); x[3] = 1.0; x[7:10] = xs[:3]
        f1, _ = o.model_f(x, np.full(4, 0.3), np.tile(y, 3))
        f0, _ = o.model_f(x, np.full(4, 0.3), np.zeros(3 * n))
        assert np.abs((f1 - f0)[7:10] - mean[:3]).max() < 1e-9 * max(1.0, np.abs(mean).max())
