"""Shader clock and board power while the lockstep launches run (sysfs of the amdgpu driver, polled from a thread).
usage: python tools/clock_under_load.py [B] [repeats]     -- is the large-batch rate bound by the power-managed clock?"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mpc_quad_ros_amd.engine import Engine  # noqa: E402
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 6
K, pre = 300, 300
files = {}
for pat, key in (("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "sclk_hz"), ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "power_uw"),
                 ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", "power_in_uw"), ("/sys/class/drm/card*/device/pp_dpm_sclk", "dpm_sclk")):
    g = sorted(glob.glob(pat))
    if g:
        files[key] = g[0]
samples, stop, phase = [], threading.Event(), ["idle"]


def rd(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None


def poll():
    while not stop.is_set():
        s = {"t": time.perf_counter(), "phase": phase[0]}
        for k, p in files.items():
            v = rd(p)
            if v is None:
                continue
            if k == "dpm_sclk":
                cur = [ln for ln in v.splitlines() if ln.rstrip().endswith("*")]
                s[k] = cur[0] if cur else v.replace("\n", " | ")
            else:
                s[k] = float(v)
        samples.append(s)
        time.sleep(0.004)


refs = bench.workload(2026, 0, B, pre + K + 30)
e = Engine(EngineConfig(batch=B, N=20, quad=hummingbird(), nb=10, basis=rgp_basis_linspace(12.0, 10)))
th = threading.Thread(target=poll); th.start()
time.sleep(0.3)
rates, smi = [], None
for r in range(REP):
    e.set_trajectories(*refs); e.sim_reset(np.tile(bench.X0, (B, 1)))
    e.sim_run(pre, 2, 5e-3); e.synchronize()
    phase[0] = "load"
    t0 = time.perf_counter()
    e.sim_steps(K, 2, 5e-3)
    if r == REP - 1:
        try:
            smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        except Exception as ex:   # noqa: BLE001
            smi = repr(ex)
    e.synchronize()
    rates.append(B * K / (time.perf_counter() - t0))
    phase[0] = "between"
phase[0] = "idle_after"; time.sleep(0.3)
stop.set(); th.join()
e.close()


def summ(ph, key):
    v = [s[key] for s in samples if s["phase"] == ph and isinstance(s.get(key), float)]
    return None if not v else {"n": len(v), "min": min(v), "median": float(np.median(v)), "max": max(v)}


out = {"B": B, "periods_per_repeat": K, "steps_per_s": rates, "files": files,
       "sclk_hz": {ph: summ(ph, "sclk_hz") for ph in ("idle", "load", "between", "idle_after")},
       "power_uw": {ph: summ(ph, "power_uw") or summ(ph, "power_in_uw") for ph in ("idle", "load", "between", "idle_after")},
       "dpm_sclk_seen_under_load": sorted({s.get("dpm_sclk") for s in samples if s["phase"] == "load" and s.get("dpm_sclk")}),
       "rocm_smi_during_last_repeat": smi}
print(json.dumps(out, indent=1))
