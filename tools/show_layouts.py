"""Which working-set layout the engine picks per (precision, shape, batch), with LDS bytes and residency (stderr of MPCQ_VERBOSE)."""
import os, sys
os.environ["MPCQ_TUNING"] = "1"; os.environ["MPCQ_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_quad_ros_amd.engine import Engine
from mpc_quad_ros_amd.params import EngineConfig, hummingbird, rgp_basis_linspace
for prec in (0, 1):
    for B, N, nb in ((1024, 20, 10), (8192, 20, 10), (8192, 20, 20), (4096, 50, 50)):
        print(f"precision {prec} B {B} N {N} nb {nb}:", file=sys.stderr, flush=True)
        Engine(EngineConfig(batch=B, N=N, quad=hummingbird(), nb=nb, basis=rgp_basis_linspace(12.0, nb), precision=prec)).close()
