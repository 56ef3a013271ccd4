import sys, time, json, numpy as np
sys.path.insert(0,'.')
import bench
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
B,N,nb,pre,K=8192,20,10,600,120
refs=bench.workload(2026,0,B,pre+2*K+30)
out={}
for name,tune in (("lockstep_g2",dict(groups=2)),("free_running",dict())):
    e=Engine(EngineConfig(batch=B,N=N,quad=hummingbird(),nb=nb,basis=rgp_basis_linspace(12.0,nb),tune=tune))
    e.set_trajectories(*refs); e.sim_reset(np.tile(bench.X0,(B,1)))
    e.sim_run(pre,2,5e-3); e.synchronize()
    t0=time.perf_counter()
    if name=="free_running": e.sim_run(K,2,5e-3)
    else: e.sim_steps(K,2,5e-3)
    e.synchronize(); dt=time.perf_counter()-t0
    x,w=e.sim_get_state()
    out[name]={"steps_per_s":B*K/dt,"digest":float(np.sum(x)+np.sum(w))}
    e.close()
print(json.dumps(out))
