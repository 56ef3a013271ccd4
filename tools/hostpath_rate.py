import sys, time, numpy as np
sys.path.insert(0,'.')
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
from mpc_quad_ros_amd.trajectories import swarm_trajectories
B=1024
e=Engine(EngineConfig(batch=B,N=20,quad=hummingbird(),nb=10,basis=rgp_basis_linspace(12.0,10)))
traj,lens=swarm_trajectories(2026,0,B); e.set_trajectories(traj,lens)
x=np.tile(np.array([0,0,3.0,1,0,0,0,0,0,0,0,0,0]),(B,1))
for _ in range(20): w,xp=e.step(x); x=xp
t=time.perf_counter()
for _ in range(100): w,xp=e.step(x); x=xp
dt=(time.perf_counter()-t)/100
print("host-buffer mpcq_step: %.3f ms per call, %.0f steps/s (PCIe + sync inclusive)"%(dt*1e3, B/dt), "kernel", e.get_time()*1e3)
