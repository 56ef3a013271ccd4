"""Diagnostic (GPU box): find the solves of a long f32 lockstep run that do not report status 0 and save the state in front of each for a
single-quadrotor replay (lane emulator: LIB=tests/wave_emu/libmpcq_emu_dbg.so).  usage: [SOAK_B= SOAK_N= SOAK_NB=] f32_hunt.py periods seed"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
K, seed = int(sys.argv[1]), int(sys.argv[2])
B, N, nb = int(os.environ.get('SOAK_B', 1024)), int(os.environ.get('SOAK_N', 20)), int(os.environ.get('SOAK_NB', 10))
refs = bench.workload(seed, 0, B, K + 10)
e, _ = bench.make_engine(B, N, nb, 1, 0, 0, seed, periods=K + 10, refs=refs)
hits = []
for k in range(K):
    e.sim_steps(1, 2, 5e-3)
    st = e.get_status()
    for b in np.flatnonzero(st != 0):
        hits.append((k, int(b), int(st[b]), int(e.get_qp_iter()[b])))
e.close()
print("solves with status != 0:", hits)
os.makedirs("gpurun_out", exist_ok=True)
for n, (k, b, st, it) in enumerate(hits[:4]):
    e, _ = bench.make_engine(B, N, nb, 1, 0, 0, seed, periods=K + 10, refs=refs)
    if k:
        e.sim_steps(k, 2, 5e-3)
    s, sol, x = e.get_state(), e.get_solver_state(), e.sim_get_state()[0]
    e.sim_steps(1, 2, 5e-3)
    print("  replayed: status", int(e.get_status()[b]), "qp_iter", int(e.get_qp_iter()[b]))
    np.savez(f"gpurun_out/f32_hunt_{n}.npz", N=N, nb=nb, k=k, b=b, x=x[b], traj=refs[0][b], len=refs[1][b], prev=sol["qp_iter"][b], w=e.sim_get_state()[1][b], wo=np.zeros(4),
             **{f"st_{name}": v[b] for name, v in s.items()})
    e.close()
