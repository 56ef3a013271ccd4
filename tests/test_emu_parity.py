"""CPU-only regression of the *product kernel sources* through the test-only lane emulator
(tests/wave_emu: the unchanged mpc_quad_ros_amd/csrc files compiled for the host, one fiber per
lane).  It checks lane logic / LDS layout / barriers of the kernels against the oracle where no
GPU exists; the real parity gate is tests/test_gpu_parity.py on the MI355X."""
import os
import subprocess

import numpy as np
import pytest

import parity_cases as pc
from mpc_quad_ros_amd.engine import Engine

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wave_emu")
EMU = os.path.join(EMU_DIR, "libmpcq_emu.so")


@pytest.fixture(scope="module", autouse=True)
def build_emu():
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)


def make(cfg):
    return Engine(cfg, lib_path=EMU)


def test_emu_teacher_forced_nominal():
    assert pc.case_teacher_forced_log(make, "log_traj1_v10_a10_gp0.npz", 6) < 1e-9


def test_emu_teacher_forced_rgp_active_bounds():
    assert pc.case_teacher_forced_log(make, "log_trajectory_v15_a5_gp2.npz", 10) < 1e-8


def test_emu_free_running_cold_start():
    pc.case_free_running_log(make, "log_traj0_v10_a10_gp2.npz", 8)


def test_emu_explicit_api():
    pc.case_explicit_api(make, B=2, N=5, nb=10)
    pc.case_explicit_api(make, B=2, N=5, nb=0)


def test_emu_swarm_closed_loop_hummingbird():
    assert pc.case_swarm_closed_loop(make, B=3, N=10, nb=10, K=6) < 1e-7


def test_emu_lane_order_independent(monkeypatch):
    # a missing barrier would make results depend on the order lanes run within a phase
    import ctypes, shutil, tempfile
    w_fwd = pc.case_teacher_forced_log(make, "log_traj0_v15_a5_gp2.npz", 3)
    env = dict(os.environ, MPCQ_EMU_REVERSE="1")
    code = ("import sys; sys.path[:0]=[%r,%r]; import parity_cases as pc, test_emu_parity as t; "
            "print(pc.case_teacher_forced_log(t.make, 'log_traj0_v15_a5_gp2.npz', 3))") % (
        os.path.dirname(EMU_DIR), os.path.dirname(os.path.dirname(EMU_DIR)))
    out = subprocess.check_output(["python", "-c", code], env=env).decode().strip().splitlines()[-1]
    assert float(out) == w_fwd


def test_emu_two_waves_per_quad(monkeypatch):
    monkeypatch.setenv("MPCQ_THREADS", "128")
    assert pc.case_teacher_forced_log(make, "log_traj0_v10_a10_gp2.npz", 3) < 1e-9
